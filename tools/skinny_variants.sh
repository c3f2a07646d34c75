#!/bin/bash
# Experiment build of the decode-step projection kernels with non-temporal weight loads (-DLINA_SKINNY_W_NT=1):
#   LINA_GLA_LIB=tools/abl/liblina_wnt.so python bench.py --no-train --no-cpu-baseline --no-chunk
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
for f in linear_skinny gla_inproj; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I $CS -DLINA_SKINNY_W_NT=1 -c $CS/$f.hip -o tools/abl/${f}_wnt.o &
done
wait
g++ -shared -fPIC $(ls $CS/*.o | grep -v "linear_skinny.o\|gla_inproj.o") tools/abl/linear_skinny_wnt.o tools/abl/gla_inproj_wnt.o -o tools/abl/liblina_wnt.so
ls -la tools/abl/liblina_wnt.so
