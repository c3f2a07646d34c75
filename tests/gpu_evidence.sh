#!/bin/bash
# Evidence session: everything under profiles/<tag>_* that bench.py and DESIGN.md cite, produced on ONE box:
#   <tag>_bench_kernel_stats.csv, <tag>_step_timeline.{txt,csv}   rocprofv3 --kernel-trace --stats of the decode bench
#   <tag>_k1w_traffic.json                                         K1w + K5: FETCH_SIZE / WRITE_SIZE, one counter per pass
#   <tag>_k2_h{4,8,16}_kernel_stats.csv, <tag>_k2_h{4,8,16}_traffic.json   K2 forward (training call) at B=64, T=4096
#   <tag>_k2b_kernel_stats.csv, <tag>_k2b_traffic.json                    K2b (three sweeps) at B=64, H=4, T=4096
# Counters are collected in their own runs with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r04}
rm -rf /tmp/kp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python bench.py --steps 300 --warmup 16 --no-train --no-cpu-baseline --no-chunk > gpurun_out/${TAG}_bench_prof.log 2>&1; echo "bench_prof=$?"
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_summary.py $db gpurun_out/${TAG}_bench_kernel_stats.csv > /dev/null
python tools/prof_step_timeline.py $db gpurun_out/${TAG}_step_timeline.csv > gpurun_out/${TAG}_step_timeline.txt; head -2 gpurun_out/${TAG}_step_timeline.txt
for C in FETCH_SIZE WRITE_SIZE; do
  K1_REPS=16 timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/k1_$C -o k1 --output-format csv -- python tools/perf_k1w.py > gpurun_out/${TAG}_k1w_$C.log 2>&1; echo "k1w $C=$?"
done
python tools/pmc_traffic.py k1w /tmp/k1_FETCH_SIZE /tmp/k1_WRITE_SIZE gpurun_out/${TAG}_k1w_traffic.json
for HH in 4 8 16; do
  rm -rf /tmp/k2p; K2_H=$HH K2_HT=0 K2_REPS=2500 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/k2p -o k2 -- python tools/perf_k2.py > gpurun_out/${TAG}_k2_h${HH}.log 2>&1; echo "k2 h$HH stats=$?"; tail -1 gpurun_out/${TAG}_k2_h${HH}.log
  db=$(find /tmp/k2p -name "*results.db" | head -1); python tools/prof_summary.py $db gpurun_out/${TAG}_k2_h${HH}_kernel_stats.csv > /dev/null
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/k2_$C; K2_H=$HH K2_HT=0 K2_REPS=4 timeout 120 rocprofv3 --kernel-trace --pmc $C -d /tmp/k2_$C -o k2 --output-format csv -- python tools/perf_k2.py > /dev/null 2>&1; echo "k2 h$HH $C=$?"
  done
  python tools/pmc_traffic.py k2 /tmp/k2_FETCH_SIZE /tmp/k2_WRITE_SIZE gpurun_out/${TAG}_k2_h${HH}_traffic.json $HH gpurun_out/${TAG}_k2_h${HH}_kernel_stats.csv
done
rm -rf /tmp/k2bp; K2_BWD=1 K2_REPS=600 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/k2bp -o k2b -- python tools/perf_k2.py > gpurun_out/${TAG}_k2b.log 2>&1; echo "k2b stats=$?"; tail -1 gpurun_out/${TAG}_k2b.log
db=$(find /tmp/k2bp -name "*results.db" | head -1); python tools/prof_summary.py $db gpurun_out/${TAG}_k2b_kernel_stats.csv > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/k2b_$C; K2_BWD=1 K2_REPS=4 timeout 120 rocprofv3 --kernel-trace --pmc $C -d /tmp/k2b_$C -o k2b --output-format csv -- python tools/perf_k2.py > /dev/null 2>&1; echo "k2b $C=$?"
done
python tools/pmc_traffic.py k2b /tmp/k2b_FETCH_SIZE /tmp/k2b_WRITE_SIZE gpurun_out/${TAG}_k2b_traffic.json 4
