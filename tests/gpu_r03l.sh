#!/bin/bash
# round 3, call l: cross_scores with the query / LayerNorm parameters requested in front of the text rows
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  PROBE=base timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee -a gpurun_out/r03l_base.log
  PROBE=base LINA_DECODE_CROSS=fused timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee -a gpurun_out/r03l_base.log
done
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "cross or engine or fused or greedy" 2>&1 | tail -3
rm -rf /tmp/kp; timeout 300 rocprofv3 --kernel-trace -d /tmp/kp -o kp -- python bench.py --steps 300 --warmup 16 --no-train --no-cpu-baseline --no-chunk > gpurun_out/r03l_bench_prof.log 2>&1
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_step_timeline.py $db gpurun_out/r03l_step_timeline.csv > gpurun_out/r03l_step_timeline.txt; head -2 gpurun_out/r03l_step_timeline.txt; sed -n '31,35p;70,74p' gpurun_out/r03l_step_timeline.txt
