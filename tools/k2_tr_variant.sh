#!/bin/bash
# Builds the opt-in K2 / K2b variants beside the product library, for A/B timing on the GPU box with tests/gpu_k2tr.sh
# (run after the normal build):
#   tools/abl/liblina_k2tr.so   gla_chunk_full.hip with -DLINA_K2_TR=1 (no transposed operand tiles)
#   tools/abl/liblina_k2w32.so  ... and -DLINA_K2_W32=1 (128 x 32 state block per wave in the forward / sweep V)
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -Wno-inline-asm -I $CS -I include"
/opt/rocm/bin/hipcc $FL -DLINA_K2_TR=1 -c $CS/gla_chunk_full.hip -o tools/abl/full_tr.o || exit 1
/opt/rocm/bin/hipcc $FL -DLINA_K2_TR=1 -DLINA_K2_W32=1 -c $CS/gla_chunk_full.hip -o tools/abl/full_w32.o || exit 1
OTHERS=$(ls $CS/*.o | grep -v gla_chunk_full.o)
g++ -shared -fPIC $OTHERS tools/abl/full_tr.o -o tools/abl/liblina_k2tr.so
g++ -shared -fPIC $OTHERS tools/abl/full_w32.o -o tools/abl/liblina_k2w32.so
ls -la tools/abl/liblina_k2tr.so tools/abl/liblina_k2w32.so
