#!/bin/bash
# Builds tools/abl/liblina_k2tr.so = the product library with gla_chunk_full.hip compiled -DLINA_K2_TR=1 (K2 / K2b without the
# transposed operand tiles), for A/B timing on the GPU box with tests/gpu_k2tr.sh.  Run after the normal build.
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -Wno-inline-asm -I $CS -I include \
  -DLINA_K2_TR=1 -c $CS/gla_chunk_full.hip -o tools/abl/full_tr.o || exit 1
g++ -shared -fPIC $(ls $CS/*.o | grep -v gla_chunk_full.o) tools/abl/full_tr.o -o tools/abl/liblina_k2tr.so
ls -la tools/abl/liblina_k2tr.so
