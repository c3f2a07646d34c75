#!/bin/bash
# Experiment builds of the K2 kernels (tools only, never the product library): tools/abl/liblina_k2<tag>.so
#   old        the C = 32 kernel of rounds 1-3 for the plain forward too (-DLINA_K2_NOPIPE)
#   old_nont   ... with the default-policy prefetch DMA
#   ord0/1/3/4 the pipelined kernel with another phase order per wave (LINA_PIPE_ORDER)
#   nont       the pipelined kernel with the default-policy prefetch DMA
#   prof       the pipelined kernel with per-phase clocks (tools/perf_k2.py K2_PROF=1)
# Run on the GPU box:  bash tests/gpu_k2pipe.sh
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I $CS"
python -c "import sys; sys.path.insert(0,'.'); import lina_speech_amd.build as b; b.build(verbose=False)" || exit 1
build() {  # tag, file, extra flags
  /opt/rocm/bin/hipcc $FL $3 -c $CS/$2.hip -o tools/abl/k2_$1.o 2>/dev/null || { echo "build $1 failed"; exit 1; }
  g++ -shared -fPIC $(ls $CS/*.o | grep -v "$2.o") tools/abl/k2_$1.o -o tools/abl/liblina_k2$1.so
  echo "built tools/abl/liblina_k2$1.so"
}
build old gla_chunk_full "-DLINA_K2_NOPIPE=1" &
build old_nont gla_chunk_full "-DLINA_K2_NOPIPE=1 -DLINA_DMA_NT=0" &
for o in 0 1 3 4; do build ord$o gla_chunk_pipe "-DLINA_PIPE_ORDER=$o" & done
build nont gla_chunk_pipe "-DLINA_DMA_NT=0" &
build prof gla_chunk_pipe "-DLINA_K2_PROF=1" &
wait
