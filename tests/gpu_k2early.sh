#!/bin/bash
# K2 (C = 32 full-head kernel) with the prefetch issued during phase A ("early"): parity on the device, ms per launch at
# B=64,H=4,T=4096 for the product and the experiment builds of tools/k2_tune.sh, K2b, the per-phase clock profiles.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r04_k2early}
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "chunk" --timeout=600 > gpurun_out/${TAG}_pytest.txt 2>&1; echo "pytest=$?"; tail -3 gpurun_out/${TAG}_pytest.txt
{
for L in "" ${K2_LIBS:-noearly ld8 ld2 prio0 prio1}; do
  if [ -z "$L" ]; then P=""; N=product; else P=tools/abl/liblina_k2$L.so; N=$L; fi
  echo -n "$N: "; LINA_GLA_LIB=$P K2_HT=0 K2_REPS=${K2_REPS:-1500} timeout 200 python tools/perf_k2.py 2>&1 | tail -1
done
echo -n "product again: "; K2_HT=0 K2_REPS=${K2_REPS:-1500} timeout 200 python tools/perf_k2.py 2>&1 | tail -1
echo -n "noearly again: "; LINA_GLA_LIB=tools/abl/liblina_k2noearly.so K2_HT=0 K2_REPS=${K2_REPS:-1500} timeout 200 python tools/perf_k2.py 2>&1 | tail -1
echo -n "product + final state: "; K2_HT=1 K2_REPS=${K2_REPS:-1500} timeout 200 python tools/perf_k2.py 2>&1 | tail -1
for L in "" noearly; do
  if [ -z "$L" ]; then P=""; N=product; else P=tools/abl/liblina_k2$L.so; N=$L; fi
  echo -n "$N K2b B=64: "; LINA_GLA_LIB=$P K2_B=64 K2_REPS=300 timeout 200 python tools/perf_k2b.py 2>&1 | tail -1
  echo -n "$N K2b b=8: "; LINA_GLA_LIB=$P K2_B=8 K2_REPS=600 timeout 200 python tools/perf_k2b.py 2>&1 | tail -1
  echo -n "$N K2 b=8 (segments): "; LINA_GLA_LIB=$P K2_B=8 K2_HT=0 K2_REPS=1000 timeout 200 python tools/perf_k2.py 2>&1 | tail -1
  echo -n "$N K2 H=16: "; LINA_GLA_LIB=$P K2_H=16 K2_HT=0 K2_REPS=1000 timeout 200 python tools/perf_k2.py 2>&1 | tail -1
done
} > gpurun_out/${TAG}_variants.txt 2>&1
cat gpurun_out/${TAG}_variants.txt
for L in prof prof_noearly; do
  echo "== $L"; LINA_GLA_LIB=tools/abl/liblina_k2$L.so K2_PROF=1 K2_HT=0 K2_REPS=300 timeout 200 python tools/perf_k2.py 2>&1 | grep -v amdgpu.ids | tail -18
done > gpurun_out/${TAG}_prof.txt 2>&1
cat gpurun_out/${TAG}_prof.txt
