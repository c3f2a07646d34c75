#!/bin/bash
# round 5, session C: the tall projection kernels -- parity, then the decode loop at B = 512 / 256 / 128 with and without them
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r05c}
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x -k "tall or linear_skinny or inproj" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest=$?"; tail -5 gpurun_out/${TAG}_pytest.log
rm -f gpurun_out/${TAG}_loop.txt
for BB in 512 256 128; do
  LINA_TALL=0 timeout 300 python tools/perf_loop.py $BB 2>> gpurun_out/${TAG}_loop.err | tee -a gpurun_out/${TAG}_loop.txt
  timeout 300 python tools/perf_loop.py $BB 2>> gpurun_out/${TAG}_loop.err | tee -a gpurun_out/${TAG}_loop.txt
done
rm -rf /tmp/kp; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python tools/perf_loop.py 512 > gpurun_out/${TAG}_b512_prof.log 2>&1; echo "prof=$?"
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_summary.py $db gpurun_out/${TAG}_b512_kernel_stats.csv
python tools/prof_step_timeline.py $db gpurun_out/${TAG}_b512_step_timeline.csv > gpurun_out/${TAG}_b512_step_timeline.txt; head -8 gpurun_out/${TAG}_b512_step_timeline.txt; tail -3 gpurun_out/${TAG}_b512_step_timeline.txt
