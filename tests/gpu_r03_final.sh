#!/bin/bash
# round 3, final session: smoke, full GPU suite (parity record), the bench line, rocprofv3 summaries for profiles/
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py --smoke > gpurun_out/r03_smoke.log 2>&1; echo "smoke=$?"; tail -2 gpurun_out/r03_smoke.log
LINA_PARITY_TAG=r03 timeout 1500 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/r03_pytest_gpu.log 2>&1; echo "pytest=$?"; tail -6 gpurun_out/r03_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "bench=$?"; tail -2 gpurun_out/r03_bench.err; cut -c1-400 gpurun_out/r03_bench.json
# decode step: kernel stats + launch timeline
rm -rf /tmp/kp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python bench.py --steps 300 --warmup 16 --no-train --no-cpu-baseline --no-chunk > gpurun_out/r03_bench_prof.log 2>&1
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_summary.py $db gpurun_out/r03_bench_kernel_stats.csv
python tools/prof_step_timeline.py $db gpurun_out/r03_step_timeline.csv > gpurun_out/r03_step_timeline.txt; head -3 gpurun_out/r03_step_timeline.txt
# K2 at the bench shape, training call (no final state): settled kernel stats + HBM traffic counters
K2_HT=0 bash tests/gpu_k2_prof.sh r03_traincall 2>&1 | grep -v "^W2026\|simple_timer" | tail -8
# train step kernel stats
bash tests/gpu_prof_train.sh r03_train 8 2>&1 | grep -v "^W2026\|simple_timer" | grep "^{" | tail -2
