#!/bin/bash
# K2 pipelined forward (gla_chunk_pipe.hip, tools-only build): parity on the device through that build, then ms per launch at
# B=64,H=4,T=4096 against the product (C = 32 kernel), then the per-phase clock profile.  Output: gpurun_out/${TAG}_*.txt
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r04_k2pipe2}
LINA_GLA_LIB=tools/abl/liblina_k2pipe.so timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "chunk and not bwd" --timeout=600 > gpurun_out/${TAG}_pytest.txt 2>&1; echo "pytest=$?"; tail -3 gpurun_out/${TAG}_pytest.txt
{
for L in "" ${K2_LIBS:-pipe}; do
  if [ -z "$L" ]; then P=""; N=product; else P=tools/abl/liblina_k2$L.so; N=$L; fi
  echo -n "$N: "; LINA_GLA_LIB=$P K2_HT=0 K2_REPS=${K2_REPS:-1500} timeout 200 python tools/perf_k2.py 2>&1 | tail -1
done
echo -n "product again: "; K2_HT=0 K2_REPS=${K2_REPS:-1500} timeout 200 python tools/perf_k2.py 2>&1 | tail -1
echo -n "pipe again: "; LINA_GLA_LIB=tools/abl/liblina_k2pipe.so K2_HT=0 K2_REPS=${K2_REPS:-1500} timeout 200 python tools/perf_k2.py 2>&1 | tail -1
echo -n "pipe + final state: "; LINA_GLA_LIB=tools/abl/liblina_k2pipe.so K2_HT=1 K2_REPS=${K2_REPS:-1500} timeout 200 python tools/perf_k2.py 2>&1 | tail -1
} > gpurun_out/${TAG}_variants.txt 2>&1
cat gpurun_out/${TAG}_variants.txt
LINA_GLA_LIB=tools/abl/liblina_k2pipeprof.so K2_PROF=pipe K2_HT=0 K2_REPS=300 timeout 200 python tools/perf_k2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_prof.txt
tail -14 gpurun_out/${TAG}_prof.txt
