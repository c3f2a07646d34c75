#!/bin/bash
# round 5, session R: stage size of the tall kernels' LDS ring with 64-row workgroups (3 x 2 product vs 3 x 4)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for V in product s31; do
  LIB=""; [ $V != product ] && LIB="tools/abl/liblina_tall_$V.so"
  for MM in 512 256; do LINA_GLA_LIB=$LIB timeout 60 python tools/perf_tall.py $MM 40 2>/dev/null | sed "s/^/$V /"; done
done | tee gpurun_out/r05r_stage.txt
