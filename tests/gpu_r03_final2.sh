#!/bin/bash
# round 3, second final session (after the train-path work): smoke, full GPU suite (parity record), the bench line,
# the train step's kernel table (rocprofv3) and op table (torch profiler)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke > gpurun_out/r03_smoke.log 2>&1; echo "smoke=$?"; tail -2 gpurun_out/r03_smoke.log
LINA_PARITY_TAG=r03 timeout 1200 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/r03_pytest_gpu.log 2>&1; echo "pytest=$?"; tail -4 gpurun_out/r03_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "bench=$?"; tail -2 gpurun_out/r03_bench.err; cut -c1-300 gpurun_out/r03_bench.json
timeout 400 bash tests/gpu_prof_train.sh r03_train 8 2>&1 | grep -v "^W2026\|simple_timer" | grep "^{" | tail -2
timeout 300 python tools/prof_train_ops.py > gpurun_out/r03_prof_ops.log 2>&1; tail -1 gpurun_out/r03_prof_ops.log
