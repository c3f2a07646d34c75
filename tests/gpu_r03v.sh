#!/bin/bash
# round 3, session v: K12b on the device, the train step with it, padded-hidden GEMM shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export LINA_PARITY_TAG=r03v
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "gate_lowrank or train or mixer or golden or forward" > gpurun_out/v_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/v_pytest.log
timeout 600 python tools/perf_train.py > gpurun_out/v_perf_train.log 2>&1; tail -1 gpurun_out/v_perf_train.log
timeout 300 python tools/perf_pad_gemm.py > gpurun_out/v_pad_gemm.txt 2>&1; tail -1 gpurun_out/v_pad_gemm.txt
timeout 600 python tools/prof_train_ops.py > gpurun_out/v_prof.log 2>&1; tail -1 gpurun_out/v_prof.log
