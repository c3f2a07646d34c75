#!/bin/bash
# quick validation of the tree on the GPU box: smoke, the full GPU suite (parity record under the tag r04c), the bench line
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke > gpurun_out/r04c_smoke.log 2>&1; echo "smoke=$?"; tail -1 gpurun_out/r04c_smoke.log
LINA_PARITY_TAG=r04c timeout 1500 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/r04c_pytest_gpu.log 2>&1; echo "pytest=$?"; tail -3 gpurun_out/r04c_pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04c_bench.json 2> gpurun_out/r04c_bench.err; echo "bench=$?"
python tools/bench_summary.py gpurun_out/r04c_bench.json
