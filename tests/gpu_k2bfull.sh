cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "chunk_bwd_full or segments_agree or value_column" 2>&1 | tail -3
for cfg in "8 300" "64 60"; do set -- $cfg
  K2_B=$1 K2_REPS=$2 rocprofv3 --kernel-trace --stats -d /tmp/k2bf_$1 -o k2bf -- python tools/perf_k2b.py 2>&1 | grep K2b
  db=$(find /tmp/k2bf_$1 -name "*results.db" | head -1); python tools/prof_summary.py "$db" gpurun_out/r02_k2bfull_b$1_kernel_stats.csv > /dev/null
done
