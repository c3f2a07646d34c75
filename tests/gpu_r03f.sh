#!/bin/bash
# round 3, call f: new defaults (16 / 8 waves in the packed projections, fast gate logsigmoid, one-round K tails, prefetched
# text rows, batched arg-max loads): full GPU test suite, step time, time stamps, timeline
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
PROBE=base timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee gpurun_out/r03f_base.log
PROBE=base LINA_SKINNY_WAVES=4 timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee -a gpurun_out/r03f_base.log
timeout 300 python tools/probe_skinny_prof.py 2>&1 | grep -v amdgpu.ids | head -16 | tee gpurun_out/r03f_skprof.log
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -x > gpurun_out/r03f_pytest_gpu.log 2>&1; echo "pytest=$?"; tail -8 gpurun_out/r03f_pytest_gpu.log
rm -rf /tmp/kp; timeout 300 rocprofv3 --kernel-trace -d /tmp/kp -o kp -- python bench.py --steps 300 --warmup 16 --no-train --no-cpu-baseline --no-chunk > gpurun_out/r03f_bench_prof.log 2>&1
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_step_timeline.py $db gpurun_out/r03f_step_timeline.csv > gpurun_out/r03f_step_timeline.txt; head -8 gpurun_out/r03f_step_timeline.txt; sed -n '30,45p;70,76p' gpurun_out/r03f_step_timeline.txt
