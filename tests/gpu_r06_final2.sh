#!/bin/bash
# round 6, LAST evidence session (after the one-launch 256 x 512 chunk forward, the 16-byte bf16-state accesses of K1w and the
# train-path changes K15 / K16 / K17): the same passes as tests/gpu_r06_evidence.sh on the final tree, Dv = 512 as ONE launch,
# without the LINA_DMA_NT A/B (unchanged kernels: profiles/r06_k2_dma_nt_ab.txt stands).
# round 6 evidence session on the (near-)final tree: smoke, the full GPU suite (parity record r06), the driver's bench command, the
# rocprofv3 kernel table + step timeline of the decode bench, and EVERY counter figure the bench line quotes re-measured in this
# session: HBM traffic of K1w (256 / 512 / 64 rows), K2 at H = 4 / 8 / 16 and Dv = 512, K2b at B = 64,
# the b = 8 segment-parallel forward / backward; SQ counters of K2 / K2b / K1w / the tall projections; and the LINA_DMA_NT A/B on
# K2b and the b = 8 passes.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r06}
timeout 300 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke=$?"; tail -1 gpurun_out/${TAG}_smoke.log
LINA_PARITY_TAG=$TAG timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest=$?"; tail -4 gpurun_out/${TAG}_pytest_gpu.log
# ---- HBM traffic (one counter per pass)
for KB in 256 512 64; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/k1_${KB}_$C; K1_B=$KB K1_REPS=8 timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/k1_${KB}_$C -o k1 --output-format csv -- python tools/perf_k1w.py > /dev/null 2>&1; echo "k1w b$KB $C=$?"
  done
  python tools/pmc_traffic.py k1w /tmp/k1_${KB}_FETCH_SIZE /tmp/k1_${KB}_WRITE_SIZE gpurun_out/${TAG}_k1w_traffic_b$KB.json $KB
done
for HH in 4 8 16; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/k2_${HH}_$C; K2_H=$HH K2_HT=0 K2_REPS=4 timeout 150 rocprofv3 --kernel-trace --pmc $C -d /tmp/k2_${HH}_$C -o k2 --output-format csv -- python tools/perf_k2.py > /dev/null 2>&1; echo "k2 h$HH $C=$?"
  done
  python tools/pmc_traffic.py k2 /tmp/k2_${HH}_FETCH_SIZE /tmp/k2_${HH}_WRITE_SIZE gpurun_out/${TAG}_k2_h${HH}_traffic.json $HH
done
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/k2dv_$C; K2_H=4 K2_DV=512 K2_HT=0 K2_REPS=4 timeout 150 rocprofv3 --kernel-trace --pmc $C -d /tmp/k2dv_$C -o k2 --output-format csv -- python tools/perf_k2.py > /dev/null 2>&1; echo "k2 dv512 $C=$?"
  rm -rf /tmp/k2b_$C; K2_BWD=1 K2_REPS=4 timeout 150 rocprofv3 --kernel-trace --pmc $C -d /tmp/k2b_$C -o k2b --output-format csv -- python tools/perf_k2.py > /dev/null 2>&1; echo "k2b b64 $C=$?"
done
python tools/pmc_traffic.py k2dv512one /tmp/k2dv_FETCH_SIZE /tmp/k2dv_WRITE_SIZE gpurun_out/${TAG}_k2_dv512_traffic.json 4
python tools/pmc_traffic.py k2b /tmp/k2b_FETCH_SIZE /tmp/k2b_WRITE_SIZE gpurun_out/${TAG}_k2b_traffic.json 4
for BWD in 0 1; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/b8_${BWD}_$C; K2_B=8 K2_HT=0 K2_BWD=$BWD K2_REPS=4 timeout 150 rocprofv3 --kernel-trace --pmc $C -d /tmp/b8_${BWD}_$C -o k2 --output-format csv -- python tools/perf_k2.py > /dev/null 2>&1; echo "b8 bwd=$BWD $C=$?"
  done
done
PMC_B=8 python tools/pmc_traffic.py k2seg /tmp/b8_0_FETCH_SIZE /tmp/b8_0_WRITE_SIZE gpurun_out/${TAG}_k2_b8_traffic.json 4
PMC_B=8 python tools/pmc_traffic.py k2b /tmp/b8_1_FETCH_SIZE /tmp/b8_1_WRITE_SIZE gpurun_out/${TAG}_k2b_b8_traffic.json 4
# ---- SQ counters
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
SQB="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
i=0
for SET in "$SQA" "$SQB"; do
  i=$((i+1))
  rm -rf /tmp/sq_k2_$i; K2_H=4 K2_HT=0 K2_REPS=4 timeout 150 rocprofv3 --kernel-trace --pmc $SET -d /tmp/sq_k2_$i -o k2 --output-format csv -- python tools/perf_k2.py > /dev/null 2>&1; echo "sq k2 $i=$?"
  rm -rf /tmp/sq_k2b_$i; K2_BWD=1 K2_REPS=4 timeout 150 rocprofv3 --kernel-trace --pmc $SET -d /tmp/sq_k2b_$i -o k2b --output-format csv -- python tools/perf_k2.py > /dev/null 2>&1; echo "sq k2b $i=$?"
  rm -rf /tmp/sq_k1_$i; K1_B=256 K1_REPS=8 timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/sq_k1_$i -o k1 --output-format csv -- python tools/perf_k1w.py > /dev/null 2>&1; echo "sq k1w $i=$?"
  rm -rf /tmp/sq_tall_$i; timeout 200 rocprofv3 --kernel-trace --pmc $SET -d /tmp/sq_tall_$i -o tall --output-format csv -- python tools/perf_tall.py 512 8 > /dev/null 2>&1; echo "sq tall $i=$?"
done
python tools/pmc_sq.py gpurun_out/${TAG}_k2_sq.json "K2 forward H=4 B=64 T=4096|gla_chunk_bf16_h256_kernel<false, 1" -- /tmp/sq_k2_1 /tmp/sq_k2_2
python tools/pmc_sq.py gpurun_out/${TAG}_k2b_sq.json "K2b sweeps B=64 H=4 T=4096 (the three instantiations pooled)|gla_chunk_bf16_h256_kernel" -- /tmp/sq_k2b_1 /tmp/sq_k2b_2
python tools/pmc_sq.py gpurun_out/${TAG}_k1w_sq.json "K1w + K5 B=256|gla_decode_window_kernel" -- /tmp/sq_k1_1 /tmp/sq_k1_2
python tools/pmc_sq.py gpurun_out/${TAG}_tall_sq.json "inproj|gla_inproj_tall_kernel" "up / head|linear_tall_kernel" -- /tmp/sq_tall_1 /tmp/sq_tall_2
# ---- the bench line quotes the counter summaries above: put them where bench.py reads them (the same files are committed)
cp gpurun_out/${TAG}_*traffic*.json gpurun_out/${TAG}_*_sq.json profiles/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench=$?"; tail -2 gpurun_out/${TAG}_bench.err
python tools/bench_summary.py gpurun_out/${TAG}_bench.json
rm -rf /tmp/kp; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kp -o kp -- python bench.py --steps 300 --warmup 16 --no-train --no-cpu-baseline --no-chunk > gpurun_out/${TAG}_bench_prof.log 2>&1; echo "bench_prof=$?"
db=$(find /tmp/kp -name "*results.db" | head -1)
python tools/prof_summary.py $db gpurun_out/${TAG}_bench_kernel_stats.csv
python tools/prof_step_timeline.py $db gpurun_out/${TAG}_step_timeline.csv > gpurun_out/${TAG}_step_timeline.txt; head -8 gpurun_out/${TAG}_step_timeline.txt
# ---- train step (a-11): wall time, then the kernel table of exactly 7 steps under rocprofv3
timeout 300 python tools/perf_train_step.py 10 > gpurun_out/${TAG}_train_step.json 2>/dev/null; echo "train_step=$?"; cat gpurun_out/${TAG}_train_step.json
{ echo "same box, same session: python tools/perf_train_step.py 10 with the round's train-path changes switched off one at a time";
  echo -n "final tree                         "; timeout 300 python tools/perf_train_step.py 10 2>/dev/null | cut -c80-170;
  echo -n "TRAIN_OPERANDS=0 (torch cat/cast/pad) "; TRAIN_OPERANDS=0 timeout 300 python tools/perf_train_step.py 10 2>/dev/null | cut -c80-170;
  echo -n "TRAIN_TORCH_ADAMW=1 (torch fused)  "; TRAIN_TORCH_ADAMW=1 timeout 300 python tools/perf_train_step.py 10 2>/dev/null | cut -c80-170;
  echo -n "final tree again                   "; timeout 300 python tools/perf_train_step.py 10 2>/dev/null | cut -c80-170; } > gpurun_out/${TAG}_train_ab.txt 2>&1
cat gpurun_out/${TAG}_train_ab.txt
timeout 300 python tools/prof_train_ops.py > gpurun_out/${TAG}_train_step_ops.txt 2>/dev/null; echo "train_ops=$?"
rm -rf /tmp/tp; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/tp -o tp -- python tools/perf_train_step.py 5 > /dev/null 2>&1; echo "train_prof=$?"
db=$(find /tmp/tp -name "*results.db" | head -1)
python tools/prof_summary.py $db gpurun_out/${TAG}_train_step_kernel_stats.csv
