#!/bin/bash
# round 6, train step: operator origin of the torch kernels (torch.profiler), kernel table under rocprofv3, wall time
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r06_train}
timeout 300 python tools/prof_train_ops.py > gpurun_out/${TAG}_ops.txt 2> gpurun_out/${TAG}_ops.err; echo "ops=$?"; head -3 gpurun_out/${TAG}_ops.txt
timeout 300 python tools/perf_train_step.py 10 > gpurun_out/${TAG}_step.json 2>/dev/null; echo "step=$?"; cat gpurun_out/${TAG}_step.json
rm -rf /tmp/tp; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/tp -o tp -- python tools/perf_train_step.py 5 > gpurun_out/${TAG}_prof.log 2>&1; echo "prof=$?"
db=$(find /tmp/tp -name "*results.db" | head -1)
python tools/prof_summary.py $db gpurun_out/${TAG}_kernel_stats.csv
