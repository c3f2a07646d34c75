#!/bin/bash
# round 3, call b: decode-step probes + per-phase K2 profiles (default / W32)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/probe_decode.py > gpurun_out/r03b_probe.log 2>&1; echo "probe=$?"; tail -14 gpurun_out/r03b_probe.log
for t in ""; do   # (the _w32 profile build of this call is recorded in profiles/r03_k2_variants.txt; the variant was removed)
  K2_PROF=1 K2_HT=0 K2_REPS=300 LINA_GLA_LIB=tools/abl/liblina_k2prof$t.so timeout 200 python tools/perf_k2.py > gpurun_out/r03b_k2prof$t.log 2>&1
  echo "== k2prof$t"; tail -22 gpurun_out/r03b_k2prof$t.log
done
