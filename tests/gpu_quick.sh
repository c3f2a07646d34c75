#!/bin/bash
# A short GPU session: selected tests (-k "$1"), optional micro-benchmarks, a short bench.  Outputs -> gpurun_out/<tag>_*
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${TAG:-q}
if [ -n "$1" ]; then
  timeout ${PYTEST_TO:-900} python -m pytest tests -m gpu -q --timeout=600 --durations=15 -k "$1" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest=$?"; tail -30 gpurun_out/${TAG}_pytest.log
fi
for m in ${MICRO}; do
  [ -x tools/micro/$m ] && timeout 120 tools/micro/$m > gpurun_out/${TAG}_$m.log 2>&1; echo "$m=$?"; cat gpurun_out/${TAG}_$m.log
done
if [ -n "${BENCH_ARGS+x}" ]; then
  timeout ${BENCH_TO:-600} python bench.py ${BENCH_ARGS} > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench=$?"; cat gpurun_out/${TAG}_bench.json; tail -5 gpurun_out/${TAG}_bench.err
fi
