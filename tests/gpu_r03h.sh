#!/bin/bash
# round 3, call h: 16 waves in the SwiGLU up-projection (VALU LayerNorm statistics), step time + stamps + engine tests
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  PROBE=base timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee -a gpurun_out/r03h_base.log
  PROBE=base LINA_SKINNY_WAVES=8 timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee -a gpurun_out/r03h_base.log
done
timeout 300 python tools/probe_skinny_prof.py 2>&1 | grep -v amdgpu.ids | head -16 | tee gpurun_out/r03h_skprof.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_bench_launch.py -m gpu -q -x -k "skinny or inproj or engine or l169 or fused or rccl or greedy" 2>&1 | tail -4
