#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-k2}
python tools/perf_k2.py 2>&1 | tail -2
rocprofv3 -L > gpurun_out/${TAG}_counters.txt 2>&1
grep -c . gpurun_out/${TAG}_counters.txt
K2_REPS=2 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d gpurun_out/${TAG}_pmc1 -o ${TAG} --output-format csv -- python tools/perf_k2.py > gpurun_out/${TAG}_pmc1.log 2>&1; echo "pmc1=$?"
K2_REPS=2 timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM -d gpurun_out/${TAG}_pmc2 -o ${TAG} --output-format csv -- python tools/perf_k2.py > gpurun_out/${TAG}_pmc2.log 2>&1; echo "pmc2=$?"
tail -3 gpurun_out/${TAG}_pmc1.log gpurun_out/${TAG}_pmc2.log
find gpurun_out/${TAG}_pmc1 gpurun_out/${TAG}_pmc2 -type f | head
