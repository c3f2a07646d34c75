#!/bin/bash
# round 3, call e: split-K width of the packed projection kernels (4 / 8 / 16 waves per workgroup), narrow in-projection tiles
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r03e.log; : > $L
for rep in 1 2; do
  for nw in 4 8 16; do
    for nar in "" 1; do
      if [ -n "$nar" ]; then export LINA_INPROJ_NARROW=1; else unset LINA_INPROJ_NARROW; fi
      echo -n "NW=$nw narrow=[$nar] " >> $L
      PROBE=base LINA_SKINNY_WAVES=$nw timeout 300 python tools/probe_decode.py 2>&1 | tail -1 >> $L
    done
  done
done
unset LINA_INPROJ_NARROW
cat $L
LINA_SKINNY_WAVES=8 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "linear_skinny_packed or inproj" 2>&1 | tail -3
LINA_SKINNY_WAVES=16 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "linear_skinny_packed or inproj" 2>&1 | tail -3
for nw in 8 16; do
  LINA_SKINNY_WAVES=$nw LINA_GLA_LIB= timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -12
import os, subprocess, sys
PY
done
for nw in 8 16; do echo "== time stamps NW=$nw"; LINA_SKINNY_WAVES=$nw timeout 300 python tools/probe_skinny_prof.py 2>&1 | grep -v amdgpu.ids | head -16 | tee -a gpurun_out/r03e_skprof.log; done
