#!/bin/bash
# round 3, session t: the train-path fusions (K12 gate, gradient slab) + where the remaining torch kernels come from
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export LINA_PARITY_TAG=r03t
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "gate_logsigmoid or split_slab or rmsnorm or conv or train or mixer or layer" > gpurun_out/t_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_pytest.log
timeout 600 python tools/perf_train.py > gpurun_out/t_perf_train.log 2>&1; tail -2 gpurun_out/t_perf_train.log
timeout 600 python tools/prof_train_ops.py > gpurun_out/t_prof.log 2>&1; tail -2 gpurun_out/t_prof.log
