#!/bin/bash
# round 3, session x: K13, block chain, weight casts inside the nodes -- parity on the device, the train step, the op table
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export LINA_PARITY_TAG=r03x
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "block_chain or conv or rmsnorm or layer_norm or swiglu or gate_lowrank or split_slab or train or mixer or golden or forward" > gpurun_out/x_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/x_pytest.log
timeout 600 python tools/perf_train.py > gpurun_out/x_perf_train.log 2>&1; tail -1 gpurun_out/x_perf_train.log
timeout 600 python tools/prof_train_ops.py > gpurun_out/x_prof.log 2>&1; tail -1 gpurun_out/x_prof.log
