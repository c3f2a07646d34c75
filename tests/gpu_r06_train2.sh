#!/bin/bash
# round 6, train step after the one-pass weight operands (K15 / K16): train-path tests, wall time, kernel table, operator origins
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r06_train2}
timeout 900 python -m pytest tests -m gpu -q -x -k "train or mlp or stacked or weight_operands or mixer or golden or lina_forward" --timeout=600 > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest=$?"; tail -2 gpurun_out/${TAG}_pytest.log
timeout 300 python tools/perf_train_step.py 10 > gpurun_out/${TAG}_step.json 2>/dev/null; echo "step=$?"; cat gpurun_out/${TAG}_step.json
timeout 300 python tools/perf_train_step.py 10 2>/dev/null | tee -a gpurun_out/${TAG}_step.json
rm -rf /tmp/tp; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/tp -o tp -- python tools/perf_train_step.py 5 > gpurun_out/${TAG}_prof.log 2>&1; echo "prof=$?"
db=$(find /tmp/tp -name "*results.db" | head -1)
python tools/prof_summary.py $db gpurun_out/${TAG}_kernel_stats.csv
timeout 300 python tools/prof_train_ops.py > gpurun_out/${TAG}_ops.txt 2> gpurun_out/${TAG}_ops.err; echo "ops=$?"; head -3 gpurun_out/${TAG}_ops.txt
