#!/bin/bash
# GPU-box session for the training path: backward parity tests, K2/K2b timing, L169 train step, kernel stats.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01t}
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -k "bwd or train" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest=$?"; tail -15 gpurun_out/${TAG}_pytest.log
timeout 600 python tools/perf_train.py > gpurun_out/${TAG}_perf.jsonl 2> gpurun_out/${TAG}_perf.err; echo "perf=$?"; cat gpurun_out/${TAG}_perf.jsonl; tail -5 gpurun_out/${TAG}_perf.err
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} -- python tools/perf_train.py --b 4 > gpurun_out/${TAG}_prof.log 2>&1; echo "prof=$?"
db=$(find gpurun_out/${TAG}_prof -name '*results.db' | head -1)
[ -n "$db" ] && python tools/prof_summary.py "$db" gpurun_out/${TAG}_kernel_stats.csv && head -25 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
