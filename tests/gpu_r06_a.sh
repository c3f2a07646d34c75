#!/bin/bash
# round 6 session A: baselines on the starting tree -- vendor-GEMM yardstick, the tall kernels, engine groups at 64 rows
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/perf_gemm_yardstick.py 512 > gpurun_out/r06_gemm_yardstick.txt 2>&1; echo "yard=$?"
timeout 120 python tools/perf_gemm_yardstick.py 64 >> gpurun_out/r06_gemm_yardstick.txt 2>&1
cat gpurun_out/r06_gemm_yardstick.txt
timeout 200 python tools/perf_tall.py 512 200 > gpurun_out/r06_tall_base.txt 2>&1; cat gpurun_out/r06_tall_base.txt
for cfg in "64 1" "32 2" "16 4" "64 2" ; do
  timeout 300 python tools/probe_two_engines.py $cfg >> gpurun_out/r06_b64_engines_a.txt 2>&1
done
cat gpurun_out/r06_b64_engines_a.txt
