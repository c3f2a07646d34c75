#!/bin/bash
# round 5, session I: the train step -- wall / host-issue time, and the rocprofv3 kernel table of exactly 7 steps
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/perf_train_step.py 5 2>/dev/null | tee gpurun_out/r05_train_step.json
rm -rf /tmp/tp; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/tp -o tp -- python tools/perf_train_step.py 5 > gpurun_out/r05i_prof.log 2>&1; echo "prof=$?"; grep ms_per_step gpurun_out/r05i_prof.log
db=$(find /tmp/tp -name "*results.db" | head -1)
python tools/prof_summary.py $db gpurun_out/r05_train_step_kernel_stats.csv
python - gpurun_out/r05_train_step_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = sum(int(r["Calls"]) for r in rows); t = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"{n} dispatches / 7 steps = {n / 7:.0f} per step; kernel time {t / 7 / 1e6:.2f} ms per step")
for r in rows[:25]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']) / 7:7.1f}/step {int(r['TotalDurationNs']) / 7 / 1e6:8.3f} ms/step")
PY
