#!/bin/bash
# A/B builds of the tall projection kernels (linear_tall.h) with other LDS ring shapes: tools/abl/liblina_tall_<NS>x<KB>.so.
#   bash tools/tall_variants.sh "6 2" "6 1" ...      then   LINA_GLA_LIB=tools/abl/liblina_tall_6x2.so python tools/perf_loop.py 512
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I $CS"
for v in "$@"; do
  set -- $v; NS=$1; KB=$2
  for f in linear_skinny gla_inproj; do
    /opt/rocm/bin/hipcc $FL -DLINA_TALL_NS=$NS -DLINA_TALL_KB=$KB -c $CS/$f.hip -o tools/abl/${f}_t${NS}x${KB}.o 2>/dev/null || exit 1 &
  done
  wait
  g++ -shared -fPIC $(ls $CS/*.o | grep -v -e linear_skinny.o -e gla_inproj.o) tools/abl/linear_skinny_t${NS}x${KB}.o tools/abl/gla_inproj_t${NS}x${KB}.o -o tools/abl/liblina_tall_${NS}x${KB}.so
  ls -la tools/abl/liblina_tall_${NS}x${KB}.so
done
