#!/bin/bash
# round 3, call r: row-split finalisation (MT x RS waves finish a tile)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do
  PROBE=base timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee -a gpurun_out/r03r_base.log
done
timeout 300 python tools/probe_skinny_prof.py 2>&1 | grep -v amdgpu.ids | head -24 | tee gpurun_out/r03r_skprof.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "skinny or inproj or engine or l169 or fused or greedy or golden or reference" 2>&1 | tail -3
