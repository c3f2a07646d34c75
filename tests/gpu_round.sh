#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench, rocprofv3 kernel stats.  Outputs -> gpurun_out/
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01}
python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke=$?"; tail -3 gpurun_out/${TAG}_smoke.log
timeout ${PYTEST_TO:-1200} python -m pytest tests -m gpu -q --timeout=600 --durations=25 ${PYTEST_ARGS} > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest=$?"; tail -25 gpurun_out/${TAG}_pytest_gpu.log
timeout ${BENCH_TO:-900} python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench=$?"; cat gpurun_out/${TAG}_bench.json; tail -5 gpurun_out/${TAG}_bench.err
timeout ${PROF_TO:-900} rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_prof -o ${TAG} -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train > gpurun_out/${TAG}_bench_prof.json 2> gpurun_out/${TAG}_bench_prof.err; echo "prof=$?"
db=$(find /tmp/${TAG}_prof -name "*results.db" | head -1); [ -n "$db" ] && python tools/prof_summary.py "$db" gpurun_out/${TAG}_kernel_stats.csv && head -25 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
