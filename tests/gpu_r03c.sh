#!/bin/bash
# round 3, call c: A/B of decode-step kernel variants (tools/decode_variants.sh) + the step's launch timeline
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r03c.log; : > $L
for rep in 1 2; do
  for lib in "" tools/abl/liblina_inproj_r02.so tools/abl/liblina_k1w_notail.so; do
    PROBE=base LINA_GLA_LIB=$lib timeout 300 python tools/probe_decode.py 2>&1 | tail -1 >> $L
  done
  PROBE=base LINA_INPROJ_NARROW=1 timeout 300 python tools/probe_decode.py 2>&1 | tail -1 >> $L
done
PROBE=state LINA_GLA_LIB=tools/abl/liblina_k1w_plain.so timeout 300 python tools/probe_decode.py 2>&1 | tail -3 >> $L
cat $L
rm -rf /tmp/kp; timeout 300 rocprofv3 --kernel-trace -d /tmp/kp -o kp -- python bench.py --steps 300 --warmup 16 --no-train --no-cpu-baseline --no-chunk > gpurun_out/r03c_bench_prof.log 2>&1
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_step_timeline.py $db gpurun_out/r03c_step_timeline.csv | tee gpurun_out/r03c_step_timeline.txt | head -90
python tools/prof_summary.py $db gpurun_out/r03c_bench_kernel_stats.csv
