#!/bin/bash
# round 3, call j: K10 / K11 on the GPU (parity + the train step with them), live-RCCL launch test again
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_bench_launch.py -m gpu -q -x -k "layer_norm or swiglu or train or golden or reference or rccl or config5 or mixer" 2>&1 | tail -6
timeout 300 python tools/perf_train.py 2>&1 | grep -v amdgpu.ids | tail -4
bash tests/gpu_prof_train.sh r03j_train 8
