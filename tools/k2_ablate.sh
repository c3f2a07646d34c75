#!/bin/bash
# Timing-only ablations of the K2 full-head kernel: each variant skips one phase (its results are wrong) to show what
# that phase costs (3: step (1) q~.S, 4: state-update MFMAs, 7: o stores, 8: o pack + stores).  The per-phase clock
# profile of tools/k2_prof.sh has replaced most of these.  Builds tools/abl/liblina_k2abl<N>.so here (hipcc, no GPU needed); run on the GPU box with
#   for n in 0 3 4 7 8; do LINA_GLA_LIB=tools/abl/liblina_k2abl$n.so python tools/perf_k2.py | tail -1; done
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
for n in 0 3 4 7 8; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I $CS -DLINA_K2_ABL=$n -c $CS/gla_chunk_full.hip -o tools/abl/full$n.o &
done
wait
for n in 0 3 4 7 8; do
  g++ -shared -fPIC $(ls $CS/*.o | grep -v gla_chunk_full.o) tools/abl/full$n.o -o tools/abl/liblina_k2abl$n.so
done
ls -la tools/abl/*.so | wc -l
