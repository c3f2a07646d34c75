#!/bin/bash
# timing probes of K2r (WRONG results by construction): line-friendly load addresses / no output stores
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for L in "" lines nostore linesns noreg; do
  if [ -z "$L" ]; then P=""; N=product; else P=tools/abl/liblina_k2$L.so; N=$L; fi
  echo -n "$N: "; LINA_GLA_LIB=$P K2_HT=0 K2_REPS=1500 timeout 200 python tools/perf_k2.py 2>&1 | tail -1
done
} > gpurun_out/r04_k2probe.txt 2>&1
cat gpurun_out/r04_k2probe.txt
LINA_GLA_LIB=tools/abl/liblina_k2linesprof.so K2_PROF=reg K2_HT=0 K2_REPS=300 timeout 200 python tools/perf_k2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_k2probe_prof.txt
tail -13 gpurun_out/r04_k2probe_prof.txt
