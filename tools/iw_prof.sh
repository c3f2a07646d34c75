#!/bin/bash
# Time-stamp build of the one-launch mixer kernel (-DLINA_IW_PROF): tools/abl/liblina_iwprof.so, read by tools/probe_iw_prof.py.
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -Wno-inline-asm -I $CS -I include -DLINA_IW_PROF=1"
/opt/rocm/bin/hipcc $FL -c $CS/gla_inproj_window.hip -o tools/abl/iw_prof.o || exit 1
g++ -shared -fPIC $(ls $CS/*.o | grep -v "gla_inproj_window.o") tools/abl/iw_prof.o -o tools/abl/liblina_iwprof.so
ls -la tools/abl/liblina_iwprof.so
