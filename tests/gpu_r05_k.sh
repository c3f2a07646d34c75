#!/bin/bash
# round 5, session K: register-ring depth of the tall kernels' variant 1 (D = 4 product, 6, 8)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for V in product d6 d8; do
  LIB=""; [ $V != product ] && LIB="tools/abl/liblina_tall_$V.so"
  LINA_GLA_LIB=$LIB LINA_TALL_V=1 timeout 60 python tools/perf_tall.py 512 40 2>/dev/null | sed "s/^/$V /"
done | tee gpurun_out/r05k_depth.txt
