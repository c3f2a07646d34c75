#!/bin/bash
# Time-stamp builds of the decode-step projection kernels (-DLINA_SKINNY_PROF): tools/abl/liblina_skprof.so, read by
# tools/probe_skinny_prof.py on the GPU box.
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -Wno-inline-asm -I $CS -I include -DLINA_SKINNY_PROF=1"
/opt/rocm/bin/hipcc $FL -c $CS/gla_inproj.hip -o tools/abl/inproj_prof.o || exit 1
/opt/rocm/bin/hipcc $FL -c $CS/linear_skinny.hip -o tools/abl/skinny_prof.o || exit 1
/opt/rocm/bin/hipcc $FL -c $CS/cross_att.hip -o tools/abl/cross_prof.o || exit 1
g++ -shared -fPIC $(ls $CS/*.o | grep -v "gla_inproj.o\|linear_skinny.o\|cross_att.o") tools/abl/inproj_prof.o tools/abl/skinny_prof.o tools/abl/cross_prof.o -o tools/abl/liblina_skprof.so
ls -la tools/abl/liblina_skprof.so
