#!/bin/bash
# round 5 final session on the final tree: smoke, the full GPU suite (parity record r05), the driver's bench command, and the
# rocprofv3 kernel table + step timeline of the decode bench (B_total = 512 on one GPU)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke > gpurun_out/r05_smoke.log 2>&1; echo "smoke=$?"; tail -1 gpurun_out/r05_smoke.log
LINA_PARITY_TAG=r05 timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest=$?"; tail -4 gpurun_out/r05_pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; echo "bench=$?"; tail -2 gpurun_out/r05_bench.err
python tools/bench_summary.py gpurun_out/r05_bench.json
rm -rf /tmp/kp; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python bench.py --steps 300 --warmup 16 --no-train --no-cpu-baseline --no-chunk > gpurun_out/r05_bench_prof.log 2>&1; echo "bench_prof=$?"
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_summary.py $db gpurun_out/r05_bench_kernel_stats.csv
python tools/prof_step_timeline.py $db gpurun_out/r05_step_timeline.csv > gpurun_out/r05_step_timeline.txt; head -8 gpurun_out/r05_step_timeline.txt
