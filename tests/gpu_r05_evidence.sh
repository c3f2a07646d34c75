#!/bin/bash
# round 5 evidence session: smoke, the driver's bench command, K1w traffic at B = 512, SQ counters of the CURRENT K2 / K2b / K1w
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r05}
timeout 300 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke=$?"; tail -1 gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench=$?"; tail -2 gpurun_out/${TAG}_bench.err
python tools/bench_summary.py gpurun_out/${TAG}_bench.json
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/k1_$C; K1_B=512 K1_REPS=8 timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/k1_$C -o k1 --output-format csv -- python tools/perf_k1w.py > gpurun_out/${TAG}_k1w_b512_$C.log 2>&1; echo "k1w b512 $C=$?"
done
python tools/pmc_traffic.py k1w /tmp/k1_FETCH_SIZE /tmp/k1_WRITE_SIZE gpurun_out/${TAG}_k1w_traffic_b512.json 512
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
# b = 8 (the training micro-batch): segment-parallel K2 forward and K2b, HBM-side traffic
for BWD in 0 1; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/b8_${BWD}_$C; K2_B=8 K2_HT=0 K2_BWD=$BWD K2_REPS=4 timeout 150 rocprofv3 --kernel-trace --pmc $C -d /tmp/b8_${BWD}_$C -o k2 --output-format csv -- python tools/perf_k2.py > /dev/null 2>&1; echo "b8 bwd=$BWD $C=$?"
  done
done
PMC_B=8 python tools/pmc_traffic.py k2seg /tmp/b8_0_FETCH_SIZE /tmp/b8_0_WRITE_SIZE gpurun_out/${TAG}_k2_b8_traffic.json 4
PMC_B=8 python tools/pmc_traffic.py k2b /tmp/b8_1_FETCH_SIZE /tmp/b8_1_WRITE_SIZE gpurun_out/${TAG}_k2b_b8_traffic.json 4
