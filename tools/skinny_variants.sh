#!/bin/bash
# Experiment builds of the decode-step projection kernels with non-temporal weight loads (-DLINA_SKINNY_W_NT=1):
#   liblina_wnt.so      all projections;   liblina_wnt_in.so   the in-projection only (its 110 MB of weights streamed past
#   the 256 MB Infinity Cache so that the other 165 MB might stay resident)
#   LINA_GLA_LIB=tools/abl/liblina_wnt_in.so python bench.py --no-train --no-cpu-baseline --no-chunk
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
for f in linear_skinny gla_inproj; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I $CS -DLINA_SKINNY_W_NT=1 -c $CS/$f.hip -o tools/abl/${f}_wnt.o &
done
wait
g++ -shared -fPIC $(ls $CS/*.o | grep -v "linear_skinny.o\|gla_inproj.o") tools/abl/linear_skinny_wnt.o tools/abl/gla_inproj_wnt.o -o tools/abl/liblina_wnt.so
g++ -shared -fPIC $(ls $CS/*.o | grep -v "gla_inproj.o") tools/abl/gla_inproj_wnt.o -o tools/abl/liblina_wnt_in.so
ls -la tools/abl/liblina_wnt*.so
