#!/bin/bash
# round 6, final tree: the full GPU suite (parity record), the driver's bench command, the train step's wall time + kernel table.
# (The counter passes of tests/gpu_r06_evidence.sh are not repeated: the decode / chunk kernels did not change after them.)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r06}
timeout 300 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke=$?"; tail -1 gpurun_out/${TAG}_smoke.log
LINA_PARITY_TAG=$TAG timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest=$?"; tail -3 gpurun_out/${TAG}_pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench=$?"; tail -2 gpurun_out/${TAG}_bench.err
python tools/bench_summary.py gpurun_out/${TAG}_bench.json
timeout 300 python tools/perf_train_step.py 10 > gpurun_out/${TAG}_train_step.json 2>/dev/null; echo "train_step=$?"; cat gpurun_out/${TAG}_train_step.json
rm -rf /tmp/tp; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/tp -o tp -- python tools/perf_train_step.py 5 > /dev/null 2>&1; echo "train_prof=$?"
db=$(find /tmp/tp -name "*results.db" | head -1)
python tools/prof_summary.py $db gpurun_out/${TAG}_train_step_kernel_stats.csv
