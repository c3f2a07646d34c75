#!/bin/bash
# Experiment builds of the K2 full-head kernel (-DLINA_K2_VAR=n), for A/B timing on the GPU box:
#   for n in 0 1; do LINA_GLA_LIB=tools/abl/liblina_k2var$n.so K2_REPS=3000 python tools/perf_k2.py | tail -1; done
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
for n in ${VARS:-0 1}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I $CS -DLINA_K2_VAR=$n -c $CS/gla_chunk_full.hip -o tools/abl/full_var$n.o &
done
wait
for n in ${VARS:-0 1}; do
  g++ -shared -fPIC $(ls $CS/*.o | grep -v gla_chunk_full.o) tools/abl/full_var$n.o -o tools/abl/liblina_k2var$n.so
done
ls -la tools/abl/liblina_k2var*.so
