#!/bin/bash
# A/B builds of the tall projection kernels (linear_tall.h) with other ring shapes: tools/abl/liblina_tall_<tag>.so.
#   bash tools/tall_variants.sh "6x2 -DLINA_TALL_NS=6 -DLINA_TALL_KB=2" "d8 -DLINA_TALL_D=8" ...
#   then   LINA_GLA_LIB=tools/abl/liblina_tall_d8.so LINA_TALL_V=1 python tools/perf_tall.py 512
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I $CS"
for v in "$@"; do
  set -- $v; TAG=$1; shift
  for f in linear_skinny gla_inproj; do
    /opt/rocm/bin/hipcc $FL "$@" -Rpass-analysis=kernel-resource-usage -c $CS/$f.hip -o tools/abl/${f}_t$TAG.o 2> tools/abl/${f}_t$TAG.log || { tail -5 tools/abl/${f}_t$TAG.log; exit 1; } &
  done
  wait
  grep -A4 "linear_tall_kernelItLb1ELb1ELi2ELi1E\|linear_tall_kernelItLb0ELb0ELi4ELi1E" tools/abl/linear_skinny_t$TAG.log | grep -E "VGPRs:|Scratch" | tr '\n' ' '; echo
  g++ -shared -fPIC $(ls $CS/*.o | grep -v -e linear_skinny.o -e gla_inproj.o) tools/abl/linear_skinny_t$TAG.o tools/abl/gla_inproj_t$TAG.o -o tools/abl/liblina_tall_$TAG.so
  ls -la tools/abl/liblina_tall_$TAG.so
done
