#!/bin/bash
# Experiment builds of the K2 full-head kernel (tools only, never the product library): tools/abl/liblina_k2<tag>.so
#   noearly     the prefetch issued after barrier (2) as in rounds 1-3 (-DLINA_K2_EARLY=0)
#   ld8 / ld2   early prefetch with 8 / 2 loader waves;  prio0: loader waves without raised priority in phase A
#   prof / prof_noearly   per-phase clocks (tools/perf_k2.py K2_PROF=1)
#   pipe        the software-pipelined 16-token-chunk forward (gla_chunk_pipe.hip, -DLINA_K2_PIPE=1)
# Run on the GPU box:  bash tests/gpu_k2early.sh
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I $CS"
python -c "import sys; sys.path.insert(0,'.'); import lina_speech_amd.build as b; b.build(verbose=False)" || exit 1
build() {  # tag, file, extra flags
  /opt/rocm/bin/hipcc $FL $3 -c $CS/$2.hip -o tools/abl/k2_$1.o -Rpass-analysis=kernel-resource-usage 2> tools/abl/k2_$1.log || { echo "build $1 failed"; tail -5 tools/abl/k2_$1.log; exit 1; }
  if grep -q "ScratchSize \[bytes/lane\]: [1-9]" tools/abl/k2_$1.log; then echo "WARNING: $1 uses scratch"; fi
  g++ -shared -fPIC $(ls $CS/*.o | grep -v "$2.o") tools/abl/k2_$1.o -o tools/abl/liblina_k2$1.so
  echo "built tools/abl/liblina_k2$1.so"
}
build noearly gla_chunk_full "-DLINA_K2_EARLY=0" &
build ld8 gla_chunk_full "-DLINA_K2_LOADERS=8" &
build ld2 gla_chunk_full "-DLINA_K2_LOADERS=2" &
build prio0 gla_chunk_full "-DLINA_K2_LPRIO=0" &
build prio1 gla_chunk_full "-DLINA_K2_LPRIO=1" &
build prof gla_chunk_full "-DLINA_K2_PROF=1" &
build prof_noearly gla_chunk_full "-DLINA_K2_PROF=1 -DLINA_K2_EARLY=0" &
wait
