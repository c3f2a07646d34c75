#!/bin/bash
# round 4: K1w with staged state loads (NPRE of 16 vectors up front) -- parity on the variant build, A/B of the decode loop
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for n in 8 12; do LINA_GLA_LIB=tools/abl/liblina_k1wnpre$n.so timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "decode_window" --timeout=300 2>&1 | tail -1; done
{
for rnd in 0 1 2; do
  for L in "" tools/abl/liblina_k1wnpre12.so tools/abl/liblina_k1wnpre8.so; do
    echo -n "round $rnd lib=${L:-product}: "; LINA_GLA_LIB=$L PROBE=base timeout 200 python tools/probe_decode.py 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
} | tee gpurun_out/r04j_k1w_npre.txt
