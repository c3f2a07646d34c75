#!/bin/bash
# round 3, call o: training streaming kernels with their loads in flight (norms, conv), in-projection tile classes
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "layer_norm or swiglu or train or conv or rmsnorm or norm or config5 or golden or reference" 2>&1 | tail -4
bash tests/gpu_prof_train.sh r03o_train 8 2>&1 | grep -v "^W2026\|simple_timer" | head -12
timeout 300 python tools/probe_skinny_prof.py 2>&1 | grep -v amdgpu.ids | head -10 | tee gpurun_out/r03o_skprof.log
