#!/bin/bash
# round 3, call s: gate workgroups of the in-projection with cacheable loads of the shared low-rank rows (A/B)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  PROBE=base timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee -a gpurun_out/r03s_base.log
  PROBE=base LINA_GLA_LIB=tools/abl/liblina_gateplain.so timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee -a gpurun_out/r03s_base.log
done
