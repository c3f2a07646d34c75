#!/bin/bash
# round 3, call m: time stamps inside cross_scores; epilogue operands behind the fragment loads (A/B)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/probe_skinny_prof.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r03m_skprof.log
for rep in 1 2; do
  PROBE=base timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee -a gpurun_out/r03m_base.log
  PROBE=base LINA_GLA_LIB=tools/abl/liblina_preafter.so timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee -a gpurun_out/r03m_base.log
done
