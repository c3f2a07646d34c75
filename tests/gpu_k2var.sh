cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for n in ${VARS:-0 1}; do
  echo "== variant $n"
  LINA_GLA_LIB=tools/abl/liblina_k2var$n.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "chunk_full_head_kernel or test_chunk_equals_recurrent or segment_parallel" 2>&1 | tail -2
  for r in 1 2; do LINA_GLA_LIB=tools/abl/liblina_k2var$n.so K2_REPS=${K2_REPS:-3000} python tools/perf_k2.py | tail -1; done
done
