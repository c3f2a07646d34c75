#!/bin/bash
# round 3, session u: ops.linear (token-split weight gradients) -- parity on the device, the train step, the op table
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export LINA_PARITY_TAG=r03u
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "linear_train or split_slab or train or mixer" > gpurun_out/u_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/u_pytest.log
timeout 600 python tools/perf_train.py > gpurun_out/u_perf_train.log 2>&1; tail -1 gpurun_out/u_perf_train.log
timeout 600 python tools/prof_train_ops.py > gpurun_out/u_prof.log 2>&1; tail -1 gpurun_out/u_prof.log
