#!/bin/bash
# kernel stats of the training-path kernels (K2 / K2b alone, then one L169 train step); keeps only the CSV summary
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01t}
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_prof -o ${TAG} -- python tools/perf_train.py --b ${2:-8} ${3:-} > gpurun_out/${TAG}_prof.log 2>&1; echo "prof=$?"
cat gpurun_out/${TAG}_prof.log | grep -v amdgpu.ids | tail -5
db=$(find /tmp/${TAG}_prof -name '*results.db' | head -1)
[ -n "$db" ] && python tools/prof_summary.py "$db" gpurun_out/${TAG}_kernel_stats.csv && head -16 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150,400-
