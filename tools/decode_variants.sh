#!/bin/bash
# Experiment builds of single decode-step kernels beside the product library (A/B on the GPU box with tests/gpu_r03c.sh):
#   tools/abl/liblina_k1w_plain.so   K1w state loads without the non-temporal hint
#   tools/abl/liblina_k1w_notail.so  K1w's K5 tail without its norm-weight / gate loads (WRONG results; latency probe)
#   tools/abl/liblina_inproj_r02.so  the round-2 in-projection (gate tiles load their rank-16 rows after the reduction)
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -Wno-inline-asm -I $CS -I include"
link() { g++ -shared -fPIC $(ls $CS/*.o | grep -v "$1.o") "$2" -o "$3"; }
/opt/rocm/bin/hipcc $FL -DLINA_K1W_STATE_PLAIN=1 -c $CS/gla_decode_window.hip -o tools/abl/k1w_plain.o || exit 1
link gla_decode_window tools/abl/k1w_plain.o tools/abl/liblina_k1w_plain.so
/opt/rocm/bin/hipcc $FL -DLINA_K1W_NO_TAIL_LOADS=1 -c $CS/gla_decode_window.hip -o tools/abl/k1w_notail.o || exit 1
link gla_decode_window tools/abl/k1w_notail.o tools/abl/liblina_k1w_notail.so
git show d856931:lina-speech_amd/csrc/gla_inproj.hip > tools/abl/inproj_r02.hip
/opt/rocm/bin/hipcc $FL -c tools/abl/inproj_r02.hip -o tools/abl/inproj_r02.o || exit 1
link gla_inproj tools/abl/inproj_r02.o tools/abl/liblina_inproj_r02.so
ls -la tools/abl/liblina_k1w_*.so tools/abl/liblina_inproj_r02.so
