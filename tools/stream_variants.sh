#!/bin/bash
# Which decode-time operands should be STREAMED past the Infinity Cache?  Builds liblina_histnt.so (K1w window history with
# non-temporal loads / stores); on the GPU box run tests/gpu_stream.sh (LINA_DECODE_STREAM = list of streamed weights).
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I $CS -DLINA_K1W_HIST_NT=1 -c $CS/gla_decode_window.hip -o tools/abl/window_histnt.o
g++ -shared -fPIC $(ls $CS/*.o | grep -v "gla_decode_window.o") tools/abl/window_histnt.o -o tools/abl/liblina_histnt.so
ls -la tools/abl/liblina_histnt.so
