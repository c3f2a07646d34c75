#!/bin/bash
# A/B of the LINA_K2_TR and LINA_K2_TR + LINA_K2_W32 variants (tools/k2_tr_variant.sh) against the default build: parity
# tests, then K2 and K2b timings.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for lib in "" tools/abl/liblina_k2tr.so tools/abl/liblina_k2w32.so; do
  echo "== lib=[$lib]"
  LINA_GLA_LIB=$lib timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "chunk_full_head_kernel or test_chunk_equals_recurrent or segment_parallel or chunk_bwd_full_head_sweeps or segments_agree" 2>&1 | tail -2
  for r in 1 2; do LINA_GLA_LIB=$lib K2_HT=0 K2_REPS=${K2_REPS:-3000} python tools/perf_k2.py | tail -1; done
  LINA_GLA_LIB=$lib K2_B=64 K2_REPS=60 python tools/perf_k2b.py | tail -1
  LINA_GLA_LIB=$lib K2_B=8 K2_REPS=300 python tools/perf_k2b.py | tail -1
done
