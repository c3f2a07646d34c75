cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "8 300" "64 60"; do set -- $cfg
  K2_B=$1 K2_REPS=$2 rocprofv3 --kernel-trace --stats -d gpurun_out/k2bf_$1 -o k2bf -- python tools/perf_k2b.py 2>&1 | grep K2b
  f=$(find gpurun_out/k2bf_$1 -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r02_k2bfull_b$1_kernel_stats.csv
  python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:9]:
    n=r["Name"]; import re
    m=re.search(r"kernel<([^>]*)>", n); tag=m.group(1) if m else n[:50]
    print(f'{tag:40s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.1f} pct={r["Percentage"]}')
PY
done
