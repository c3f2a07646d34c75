#!/bin/bash
# round 6: 256 x 512 heads (expand_v = 2) as ONE chunk-forward launch of two XCD-paired workgroups per head: parity, time against the
# two-launch form in the same session, HBM traffic (two --pmc passes, --kernel-trace only)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r06}
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "chunk" --timeout=600 > gpurun_out/${TAG}_dv512_pytest.log 2>&1; echo "pytest=$?"; tail -2 gpurun_out/${TAG}_dv512_pytest.log
{
for ONE in 1 0 1 0; do
  echo "dv512_one_launch=$ONE"; K2_DV512_ONE=$ONE K2_H=4 K2_DV=512 K2_HT=0 K2_REPS=100 timeout 120 python tools/perf_k2.py
done
echo "with final state:"; for ONE in 1 0; do K2_DV512_ONE=$ONE K2_H=4 K2_DV=512 K2_HT=1 K2_REPS=100 timeout 120 python tools/perf_k2.py; done
echo "Dv = 256 in the same session:"; K2_H=4 K2_HT=0 K2_REPS=100 timeout 120 python tools/perf_k2.py
echo "B = 32 (128 heads: 256 workgroups, one round):"; for ONE in 1 0; do K2_B=32 K2_DV512_ONE=$ONE K2_H=4 K2_DV=512 K2_HT=0 K2_REPS=100 timeout 120 python tools/perf_k2.py; done
} > gpurun_out/${TAG}_k2_dv512_ab.txt 2>&1
cat gpurun_out/${TAG}_k2_dv512_ab.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/k2dv_$C; K2_H=4 K2_DV=512 K2_HT=0 K2_REPS=4 timeout 150 rocprofv3 --kernel-trace --pmc $C -d /tmp/k2dv_$C -o k2 --output-format csv -- python tools/perf_k2.py > /dev/null 2>&1; echo "k2 dv512 $C=$?"
done
python tools/pmc_traffic.py k2dv512one /tmp/k2dv_FETCH_SIZE /tmp/k2dv_WRITE_SIZE gpurun_out/${TAG}_k2_dv512_traffic.json 4
