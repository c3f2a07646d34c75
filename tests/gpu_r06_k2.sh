#!/bin/bash
# round 6: K2 forward with the loader waves out of phase A (LINA_K2_SKIPA=1, the build) against the sixteen-wave phase A (=0)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "chunk" 2>&1 | grep -v amdgpu.ids | tail -3
for i in 1 2; do
  K2_HT=0 K2_REPS=300 timeout 120 python tools/perf_k2.py 2>&1 | tail -1 | sed "s/^/skipa=1 /" >> gpurun_out/r06_k2_skipa.txt
  LINA_GLA_LIB=tools/abl/liblina_k2_skipa0.so K2_HT=0 K2_REPS=300 timeout 120 python tools/perf_k2.py 2>&1 | tail -1 | sed "s/^/skipa=0 /" >> gpurun_out/r06_k2_skipa.txt
done
K2_B=8 K2_HT=0 K2_REPS=300 timeout 120 python tools/perf_k2.py 2>&1 | tail -1 | sed "s/^/skipa=1 b8 /" >> gpurun_out/r06_k2_skipa.txt
LINA_GLA_LIB=tools/abl/liblina_k2_skipa0.so K2_B=8 K2_HT=0 K2_REPS=300 timeout 120 python tools/perf_k2.py 2>&1 | tail -1 | sed "s/^/skipa=0 b8 /" >> gpurun_out/r06_k2_skipa.txt
K2_HT=1 K2_REPS=300 timeout 120 python tools/perf_k2.py 2>&1 | tail -1 | sed "s/^/skipa=1 ht /" >> gpurun_out/r06_k2_skipa.txt
LINA_GLA_LIB=tools/abl/liblina_k2prof.so K2_PROF=1 K2_HT=0 K2_REPS=20 timeout 120 python tools/perf_k2.py 2>&1 | tail -18 >> gpurun_out/r06_k2_skipa.txt
cat gpurun_out/r06_k2_skipa.txt
