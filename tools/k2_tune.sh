#!/bin/bash
# Experiment builds of the K2 kernels (tools only, never the product library): tools/abl/liblina_k2<tag>.so
#   pipe        the software-pipelined 16-token-chunk forward (gla_chunk_pipe.hip) takes the plain forward (-DLINA_K2_PIPE=1)
#   pipeprof    ... with per-phase clocks (tools/perf_k2.py K2_PROF=pipe)
#   early       the C = 32 kernel with the prefetch issued during phase A (-DLINA_K2_EARLY=1; measured slower, round 4)
#   prof        the C = 32 kernel with per-phase clocks (tools/perf_k2.py K2_PROF=1)
# Run on the GPU box:  bash tests/gpu_k2pipe.sh
cd "$(dirname "$0")/.."
mkdir -p tools/abl
CS=lina-speech_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I $CS"
python -c "import sys; sys.path.insert(0,'.'); import lina_speech_amd.build as b; b.build(verbose=False)" || exit 1
cc() {  # object tag, file, flags
  /opt/rocm/bin/hipcc $FL $3 -c $CS/$2.hip -o tools/abl/k2_$1.o -Rpass-analysis=kernel-resource-usage 2> tools/abl/k2_$1.log || { echo "build $1 failed"; tail -5 tools/abl/k2_$1.log; exit 1; }
  if grep -q "ScratchSize \[bytes/lane\]: [1-9]" tools/abl/k2_$1.log; then echo "WARNING: $1 uses scratch"; fi
}
link() {  # library tag, replaced files (regex), objects
  g++ -shared -fPIC $(ls $CS/*.o | grep -Ev "$2") $3 -o tools/abl/liblina_k2$1.so && echo "built tools/abl/liblina_k2$1.so"
}
cc full_pipe gla_chunk_full "-DLINA_K2_PIPE=1" &
cc pipe_prof gla_chunk_pipe "-DLINA_K2_PROF=1" &
cc early gla_chunk_full "-DLINA_K2_EARLY=1" &
cc prof gla_chunk_full "-DLINA_K2_PROF=1" &
wait
link pipe "gla_chunk_full.o" tools/abl/k2_full_pipe.o
link pipeprof "gla_chunk_full.o|gla_chunk_pipe.o" "tools/abl/k2_full_pipe.o tools/abl/k2_pipe_prof.o"
link early "gla_chunk_full.o" tools/abl/k2_early.o
link prof "gla_chunk_full.o" tools/abl/k2_prof.o
