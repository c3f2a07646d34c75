#!/bin/bash
# rerun of the two pieces of tests/gpu_r05_evidence.sh that failed in the first session: the bench line and the SQ passes
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=r05
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench=$?"; tail -2 gpurun_out/${TAG}_bench.err
python tools/bench_summary.py gpurun_out/${TAG}_bench.json
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
SQB="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
i=0
for SET in "$SQA" "$SQB"; do
  i=$((i+1))
  rm -rf /tmp/sq_k2_$i; K2_H=4 K2_HT=0 K2_REPS=4 timeout 150 rocprofv3 --kernel-trace --pmc $SET -d /tmp/sq_k2_$i -o k2 --output-format csv -- python tools/perf_k2.py > gpurun_out/${TAG}_sq_k2_$i.log 2>&1; echo "sq k2 $i=$?"
  rm -rf /tmp/sq_k2b_$i; K2_BWD=1 K2_REPS=4 timeout 150 rocprofv3 --kernel-trace --pmc $SET -d /tmp/sq_k2b_$i -o k2b --output-format csv -- python tools/perf_k2.py > gpurun_out/${TAG}_sq_k2b_$i.log 2>&1; echo "sq k2b $i=$?"
  rm -rf /tmp/sq_k1_$i; K1_REPS=8 timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/sq_k1_$i -o k1 --output-format csv -- python tools/perf_k1w.py > gpurun_out/${TAG}_sq_k1w_$i.log 2>&1; echo "sq k1w $i=$?"
done
python tools/pmc_sq.py gpurun_out/${TAG}_k2_sq.json "K2 forward H=4 B=64 T=4096|gla_chunk_bf16_h256_kernel<false, 1" -- /tmp/sq_k2_1 /tmp/sq_k2_2
python tools/pmc_sq.py gpurun_out/${TAG}_k2b_sq.json "K2b sweeps B=64 H=4 T=4096 (the three instantiations pooled)|gla_chunk_bf16_h256_kernel" -- /tmp/sq_k2b_1 /tmp/sq_k2b_2
python tools/pmc_sq.py gpurun_out/${TAG}_k1w_sq.json "K1w + K5 B=64|gla_decode_window_kernel" -- /tmp/sq_k1_1 /tmp/sq_k1_2
