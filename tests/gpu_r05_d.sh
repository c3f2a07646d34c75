#!/bin/bash
# round 5, session D: LDS ring shapes of the tall projection kernels (tools/tall_variants.sh) in the decode loop at B = 512 / 256
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r05d_variants.txt
for V in product 4x2 6x1 6x2 10x1 12x1; do
  LIB=""; [ $V != product ] && LIB="tools/abl/liblina_tall_$V.so"
  for BB in 512 256; do
    echo -n "$V: " >> gpurun_out/r05d_variants.txt
    LINA_GLA_LIB=$LIB timeout 300 python tools/perf_loop.py $BB 2>> gpurun_out/r05d.err >> gpurun_out/r05d_variants.txt
  done
done
cat gpurun_out/r05d_variants.txt
for V in 6x2 12x1; do
rm -rf /tmp/kp; LINA_GLA_LIB=tools/abl/liblina_tall_$V.so timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python tools/perf_loop.py 512 > gpurun_out/r05d_prof.log 2>&1; echo "prof=$?"
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_step_timeline.py $db gpurun_out/r05d_${V}_step_timeline.csv > gpurun_out/r05d_${V}_step_timeline.txt; head -6 gpurun_out/r05d_${V}_step_timeline.txt; tail -2 gpurun_out/r05d_${V}_step_timeline.txt
done
