#!/bin/bash
# round 5, session L: SQ counters of the tall projection kernels (variant 0 and 1), micro script
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
SQB="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
for TV in 0 1; do
  i=0
  for SET in "$SQA" "$SQB"; do
    i=$((i+1))
    rm -rf /tmp/sqt_${TV}_$i; LINA_TALL_V=$TV timeout 120 rocprofv3 --kernel-trace --pmc $SET -d /tmp/sqt_${TV}_$i -o t --output-format csv -- python tools/perf_tall.py 512 4 > /dev/null 2>&1; echo "sq tall v$TV $i=$?"
  done
  python tools/pmc_sq.py gpurun_out/r05l_tall_v${TV}_sq.json "inproj|gla_inproj_tall_kernel" "up|linear_tall_kernel<unsigned short, true, true" "head|linear_tall_kernel<unsigned short, false, false" -- /tmp/sqt_${TV}_1 /tmp/sqt_${TV}_2
done
