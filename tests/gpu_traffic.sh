#!/bin/bash
# HBM traffic of the dominant kernel (K1d) from the L2 fabric counters, one counter per pass (guide: MI355X_MICROARCH.md HBM).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-tr}
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/${TAG}_$C -o ${TAG} --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-chunk > gpurun_out/${TAG}_$C.log 2>&1; echo "$C=$?"
done
