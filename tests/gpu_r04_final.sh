#!/bin/bash
# round 4 final session: smoke, the full GPU suite (parity record), the bench line as the driver runs it, evidence
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r04}
timeout 300 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke=$?"; tail -2 gpurun_out/${TAG}_smoke.log
LINA_PARITY_TAG=${TAG} timeout 1500 python -m pytest tests -m gpu -q --timeout=600 --durations=8 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest=$?"; tail -16 gpurun_out/${TAG}_pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench=$?"; tail -3 gpurun_out/${TAG}_bench.err
python tools/bench_summary.py gpurun_out/${TAG}_bench.json
bash tests/gpu_evidence.sh ${TAG}
