#!/bin/bash
# Experiment builds of the K2 forward kernels (tools only, never the product library): tools/abl/liblina_k2<tag>.so
#   noreg       the C = 32 DMA kernel of rounds 1-4 for the plain forward (-DLINA_K2_NOREG=1)
#   ta2 / ord1  K2r with a shallower k^^T ring / with waves 8..15 running phase A before phase B
#   prof        K2r with per-phase clocks (tools/perf_k2.py K2_PROF=reg)
# Run on the GPU box:  bash tests/gpu_k2reg.sh
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
cc noreg gla_chunk_full "-DLINA_K2_NOREG=1" &
cc ta2 gla_chunk_reg "-DLINA_K2R_TA=2" &
cc ord1 gla_chunk_reg "-DLINA_K2R_ORDER=1" &
cc prof gla_chunk_reg "-DLINA_K2_PROF=1" &
cc lines gla_chunk_reg "-DLINA_K2R_PROBE_LINES=1" &
cc linesns gla_chunk_reg "-DLINA_K2R_PROBE_LINES=1 -DLINA_K2R_PROBE_NOSTORE=1" &
cc nostore gla_chunk_reg "-DLINA_K2R_PROBE_NOSTORE=1" &
cc linesprof gla_chunk_reg "-DLINA_K2R_PROBE_LINES=1 -DLINA_K2_PROF=1" &
wait
link noreg "gla_chunk_full.o" tools/abl/k2_noreg.o
link ta2 "gla_chunk_reg.o" tools/abl/k2_ta2.o
link ord1 "gla_chunk_reg.o" tools/abl/k2_ord1.o
link prof "gla_chunk_reg.o" tools/abl/k2_prof.o
for t in lines linesns nostore linesprof; do link $t "gla_chunk_reg.o" tools/abl/k2_$t.o; done
