#!/bin/bash
# K2 alone at the bench shape (B=64, H=4, T=4096): rocprofv3 kernel stats under settled clocks + HBM traffic counters
# (one counter per pass; MI355X_MICROARCH.md, HBM).  Outputs -> gpurun_out/<tag>_k2_*
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r01}
K2_REPS=2500 timeout 120 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_k2prof -o ${TAG} -- python tools/perf_k2.py > gpurun_out/${TAG}_k2_prof.log 2>&1; echo "stats=$?"; tail -1 gpurun_out/${TAG}_k2_prof.log
db=$(find gpurun_out/${TAG}_k2prof -name "*results.db" | head -1); [ -n "$db" ] && python tools/prof_summary.py "$db" gpurun_out/${TAG}_k2_kernel_stats.csv && head -4 gpurun_out/${TAG}_k2_kernel_stats.csv | cut -c1-80,300-420; rm -rf gpurun_out/${TAG}_k2prof
for C in FETCH_SIZE WRITE_SIZE; do
  K2_REPS=4 timeout 100 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/${TAG}_k2_$C -o ${TAG} --output-format csv -- python tools/perf_k2.py > gpurun_out/${TAG}_k2_$C.log 2>&1; echo "$C=$?"
  f=$(find gpurun_out/${TAG}_k2_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $C <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gla_chunk_bf16_h256' in r.get('Kernel_Name', '') and r.get('Counter_Name') == sys.argv[2]]
v = [float(r['Counter_Value']) for r in rows]
print(sys.argv[2], "dispatches", len(v), "mean", sum(v) / max(len(v), 1), "min", min(v) if v else None, "max", max(v) if v else None)
PY
  rm -rf gpurun_out/${TAG}_k2_$C
done
