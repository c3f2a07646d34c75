#!/bin/bash
# round 4 final session: K2 clamp A/B, smoke, the full GPU suite (parity record), the bench line as the driver runs it, evidence
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r04}
{
for L in "" k2clamp "" k2clamp; do
  if [ -z "$L" ]; then P=""; N="product (no clamp in the optimistic scan)"; else P=tools/abl/liblina_$L.so; N="clamp always (round-3 form)"; fi
  echo -n "$N: "; LINA_GLA_LIB=$P K2_HT=0 K2_REPS=1500 timeout 200 python tools/perf_k2.py 2>&1 | tail -1
done
} > gpurun_out/${TAG}_k2_clamp_ab.txt 2>&1; cat gpurun_out/${TAG}_k2_clamp_ab.txt
timeout 300 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke=$?"; tail -2 gpurun_out/${TAG}_smoke.log
LINA_PARITY_TAG=${TAG} timeout 1500 python -m pytest tests -m gpu -q --timeout=600 --durations=8 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest=$?"; tail -16 gpurun_out/${TAG}_pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench=$?"; tail -3 gpurun_out/${TAG}_bench.err
python tools/bench_summary.py gpurun_out/${TAG}_bench.json
bash tests/gpu_evidence.sh ${TAG}
