#!/bin/bash
# round 4, session b: the new parity tests + decode with a 16-token state window against 8
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
LINA_PARITY_TAG=r04b timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -k "config4 or config5 or b64_free_running or decode_window or chunk_bwd_generic" > gpurun_out/r04b_pytest.log 2>&1; echo "pytest=$?"; tail -12 gpurun_out/r04b_pytest.log
for W in 8 16 8 16; do
  timeout 300 python bench.py --window $W --no-chunk --no-train --no-cpu-baseline > gpurun_out/r04b_bench_w$W.json 2> gpurun_out/r04b_bench_w$W.err; echo -n "window $W: "; python tools/bench_summary.py gpurun_out/r04b_bench_w$W.json | head -1
done
