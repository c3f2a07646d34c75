#!/bin/bash
# round 3, call n: branch-free / raw epilogue preloads behind the fragment loads, clamped loads in arg-max + cross scores
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  PROBE=base timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee -a gpurun_out/r03n_base.log
done
PROBE=base LINA_DECODE_CROSS=fused timeout 300 python tools/probe_decode.py 2>&1 | tail -1 | tee -a gpurun_out/r03n_base.log
timeout 300 python tools/probe_skinny_prof.py 2>&1 | grep -v amdgpu.ids | grep -v "other (head" -A0 | head -24 | tee gpurun_out/r03n_skprof.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "cross or skinny or inproj or engine or l169 or fused or greedy or golden or reference or pick or argmax or topk" 2>&1 | tail -4
rm -rf /tmp/kp; timeout 300 rocprofv3 --kernel-trace -d /tmp/kp -o kp -- python bench.py --steps 300 --warmup 16 --no-train --no-cpu-baseline --no-chunk > gpurun_out/r03n_bench_prof.log 2>&1
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_step_timeline.py $db gpurun_out/r03n_step_timeline.csv > gpurun_out/r03n_step_timeline.txt; head -7 gpurun_out/r03n_step_timeline.txt; sed -n '31,42p;70,74p' gpurun_out/r03n_step_timeline.txt
