#!/bin/bash
# round 5, session E: the register-ring variant of the tall projection kernels (LINA_TALL_V=1) vs the LDS ring (0) vs the 64-row kernels
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x -k "tall" > gpurun_out/r05e_pytest.log 2>&1; echo "pytest=$?"; tail -3 gpurun_out/r05e_pytest.log
rm -f gpurun_out/r05e_loop.txt
for BB in 512 256 128; do
  for V in 0 1; do
    echo -n "V=$V " >> gpurun_out/r05e_loop.txt
    LINA_TALL_V=$V timeout 300 python tools/perf_loop.py $BB 2>> gpurun_out/r05e.err >> gpurun_out/r05e_loop.txt
  done
done
cat gpurun_out/r05e_loop.txt
rm -rf /tmp/kp; LINA_TALL_V=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python tools/perf_loop.py 512 > gpurun_out/r05e_prof.log 2>&1; echo "prof=$?"
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_step_timeline.py $db gpurun_out/r05e_v1_step_timeline.csv > gpurun_out/r05e_v1_step_timeline.txt; head -6 gpurun_out/r05e_v1_step_timeline.txt; tail -2 gpurun_out/r05e_v1_step_timeline.txt
