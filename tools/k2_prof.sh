#!/bin/bash
# Per-phase shader-clock profile of the K2 full-head kernel (workgroup 0, every wave): builds tools/abl/liblina_k2prof.so
# with -DLINA_K2_PROF (clock64() reads around every barrier / step; each read also waits for the wave's LDS traffic, so the
# build is ~5 % slower than the product).  Run on the GPU box:
#   K2_PROF=1 LINA_GLA_LIB=tools/abl/liblina_k2prof.so python tools/perf_k2.py
# (K2_EXTRA / K2_TAG: extra -D flags and a library suffix for experiment builds)
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I $CS -DLINA_K2_PROF=1 ${K2_EXTRA} -c $CS/gla_chunk_full.hip -o tools/abl/full_prof.o 2>/dev/null || exit 1
g++ -shared -fPIC $(ls $CS/*.o | grep -v gla_chunk_full.o) tools/abl/full_prof.o -o tools/abl/liblina_k2prof${K2_TAG}.so
ls -la tools/abl/liblina_k2prof${K2_TAG}.so
