#!/bin/bash
# K2r (gla_chunk_reg.hip: register-prefetched forward): parity on the device, ms per launch at B=64,H=4,T=4096 against the DMA
# kernel and K2r's variants (tools/k2_tune.sh), the per-phase clock profile.  Output: gpurun_out/${TAG}_*.txt
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r04_k2reg}
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "chunk and not bwd" --timeout=600 > gpurun_out/${TAG}_pytest.txt 2>&1; echo "pytest=$?"; tail -3 gpurun_out/${TAG}_pytest.txt
{
for L in "" ${K2_LIBS:-noreg ta2 ord1}; do
  if [ -z "$L" ]; then P=""; N=product; else P=tools/abl/liblina_k2$L.so; N=$L; fi
  echo -n "$N: "; LINA_GLA_LIB=$P K2_HT=0 K2_REPS=${K2_REPS:-1500} timeout 200 python tools/perf_k2.py 2>&1 | tail -1
done
echo -n "product again: "; K2_HT=0 K2_REPS=${K2_REPS:-1500} timeout 200 python tools/perf_k2.py 2>&1 | tail -1
echo -n "noreg again: "; LINA_GLA_LIB=tools/abl/liblina_k2noreg.so K2_HT=0 K2_REPS=${K2_REPS:-1500} timeout 200 python tools/perf_k2.py 2>&1 | tail -1
echo -n "product + final state: "; K2_HT=1 K2_REPS=${K2_REPS:-1500} timeout 200 python tools/perf_k2.py 2>&1 | tail -1
} > gpurun_out/${TAG}_variants.txt 2>&1
cat gpurun_out/${TAG}_variants.txt
LINA_GLA_LIB=tools/abl/liblina_k2prof.so K2_PROF=reg K2_HT=0 K2_REPS=300 timeout 200 python tools/perf_k2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_prof.txt
tail -14 gpurun_out/${TAG}_prof.txt
