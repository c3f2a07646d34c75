#!/bin/bash
# round 3, session z: pipelined K3b, sequence-form SwiGLU forward, K5b grid -- parity on the device, the train step, kernel table
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export LINA_PARITY_TAG=r03z
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "conv or swiglu or rmsnorm or split_slab or train or golden" > gpurun_out/z_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/z_pytest.log
timeout 300 python tools/perf_train.py > gpurun_out/z_perf_train.log 2>&1; tail -1 gpurun_out/z_perf_train.log
timeout 300 python tools/prof_train_ops.py > gpurun_out/z_prof.log 2>&1; tail -1 gpurun_out/z_prof.log
