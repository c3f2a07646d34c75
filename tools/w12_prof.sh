#!/bin/bash
# Per-phase clock build of the twelve-wave K2 (-DLINA_W12_PROF): tools/abl/liblina_w12prof.so, read by tools/perf_k2.py (K2_W12_PROF=1).
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -Wno-inline-asm -I $CS -I include -DLINA_W12_PROF=1"
/opt/rocm/bin/hipcc $FL -c $CS/gla_chunk_w12.hip -o tools/abl/w12_prof.o || exit 1
g++ -shared -fPIC $(ls $CS/*.o | grep -v "gla_chunk_w12.o") tools/abl/w12_prof.o -o tools/abl/liblina_w12prof.so
ls -la tools/abl/liblina_w12prof.so
