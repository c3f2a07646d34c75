#!/bin/bash
# round 6, the train step on the round's LAST tree: wall time, same-box A/B of the late-round switches, kernel table of exactly 7 steps
# under rocprofv3, torch.profiler operator table
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r06t}
timeout 300 python tools/perf_train_step.py 10 > gpurun_out/${TAG}_train_step.json 2>/dev/null; echo "train_step=$?"; cat gpurun_out/${TAG}_train_step.json
{ echo "same box, same session: python tools/perf_train_step.py 10 with the round's train-path changes switched off one at a time";
  echo -n "final tree                         "; timeout 300 python tools/perf_train_step.py 10 2>/dev/null | cut -c80-170;
  echo -n "TRAIN_OPERANDS=0 (torch cat/cast/pad) "; TRAIN_OPERANDS=0 timeout 300 python tools/perf_train_step.py 10 2>/dev/null | cut -c80-170;
  echo -n "TRAIN_SPLIT_GEMM=0 (N = 4160 GEMMs) "; TRAIN_SPLIT_GEMM=0 timeout 300 python tools/perf_train_step.py 10 2>/dev/null | cut -c80-170;
  echo -n "TRAIN_TORCH_ADAMW=1 (torch fused)  "; TRAIN_TORCH_ADAMW=1 timeout 300 python tools/perf_train_step.py 10 2>/dev/null | cut -c80-170;
  echo -n "final tree again                   "; timeout 300 python tools/perf_train_step.py 10 2>/dev/null | cut -c80-170; } > gpurun_out/${TAG}_train_ab.txt 2>&1
cat gpurun_out/${TAG}_train_ab.txt
rm -rf /tmp/tp; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/tp -o tp -- python tools/perf_train_step.py 5 > /dev/null 2>&1; echo "train_prof=$?"
db=$(find /tmp/tp -name "*results.db" | head -1)
python tools/prof_summary.py $db gpurun_out/${TAG}_train_step_kernel_stats.csv
timeout 300 python tools/prof_train_ops.py > gpurun_out/${TAG}_train_step_ops.txt 2>/dev/null; echo "train_ops=$?"
