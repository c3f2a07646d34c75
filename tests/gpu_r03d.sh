#!/bin/bash
# round 3, call d: time stamps inside the projection launches, K6e tests + sampled-decode timing
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/probe_skinny_prof.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03d_skprof.log; cat gpurun_out/r03d_skprof.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "sample_pick_embed or topk or linear_skinny_packed or inproj or greedy_pick" 2>&1 | tail -3
PROBE=sampled timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee gpurun_out/r03d_sampled.log
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "sampling or hipgraph" 2>&1 | tail -3
