#!/bin/bash
# round 5, session B: where the B=512 step spends its time -- rocprofv3 kernel table + the launches of one step in order
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for BB in 512 256 128; do
rm -rf /tmp/kp; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python bench.py --batch-per-gpu $BB --steps 200 --warmup 16 --no-train --no-cpu-baseline --no-chunk > gpurun_out/r05b_b${BB}_prof.log 2>&1; echo "prof $BB=$?"
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_summary.py $db gpurun_out/r05b_b${BB}_kernel_stats.csv
python tools/prof_step_timeline.py $db gpurun_out/r05b_b${BB}_step_timeline.csv > gpurun_out/r05b_b${BB}_step_timeline.txt; head -3 gpurun_out/r05b_b${BB}_step_timeline.txt
done
