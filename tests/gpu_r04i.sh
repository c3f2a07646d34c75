#!/bin/bash
# round 4: K2 on twelve specialised waves -- parity on the GPU, A/B against the sixteen-wave kernel, per-phase clocks
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "chunk" --timeout=600 2>&1 | tail -2
{
for rnd in 0 1; do
  for W in 16 12; do echo -n "waves=$W: "; LINA_K2_WAVES=$W K2_HT=0 K2_REPS=1500 timeout 200 python tools/perf_k2.py 2>&1 | tail -1; done
done
LINA_GLA_LIB=tools/abl/liblina_w12prof.so K2_W12_PROF=1 K2_HT=0 K2_REPS=300 timeout 200 python tools/perf_k2.py 2>&1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04i_k2_w12.txt
