#!/bin/bash
# engine clock / power while K2 runs back to back (the chip is power-managed: cycles per chunk and time per chunk do not
# move together).  Run on the GPU box after one warm-up run of tools/perf_k2.py.
K2_REPS=${1:-20000} python tools/perf_k2.py > gpurun_out/clk_k2.log 2>&1 &
pid=$!
for i in $(seq 1 40); do
  sleep 0.7
  kill -0 $pid 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|power" | tr '\n' ' '; echo
done
wait $pid
tail -1 gpurun_out/clk_k2.log
