cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for ht in 0 1; do
  K2_HT=$ht K2_REPS=2500 rocprofv3 --kernel-trace --stats -d /tmp/k2ht_$ht -o k2 -- python tools/perf_k2.py 2>&1 | grep "K2\["
  db=$(find /tmp/k2ht_$ht -name "*results.db" | head -1); python tools/prof_summary.py "$db" gpurun_out/r02_k2_ht${ht}_kernel_stats.csv > /dev/null; head -2 gpurun_out/r02_k2_ht${ht}_kernel_stats.csv | tail -1 | cut -d, -f2-6 | sed 's/^.*)",//'
done
