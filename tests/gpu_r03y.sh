#!/bin/bash
# round 3, session y: K14 + the captured train step -- parity on the device, timing, the op table
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export LINA_PARITY_TAG=r03y
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "cross_entropy or train or golden or forward or conv" > gpurun_out/y_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/y_pytest.log
timeout 600 python tools/probe_train_graph.py > gpurun_out/y_train_graph.txt 2>&1; tail -2 gpurun_out/y_train_graph.txt | cut -c1-300
timeout 600 python tools/prof_train_ops.py > gpurun_out/y_prof.log 2>&1; tail -1 gpurun_out/y_prof.log
