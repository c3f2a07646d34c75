#!/bin/bash
# round 5, session J: in-projection with prefetched conv caches, K2b at b = 8 with 4 / 8 segments, then the full GPU suite
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 60 python tools/perf_tall.py 512 40 2>/dev/null | tee gpurun_out/r05j_perf_tall.txt
for NS in 8 4 2; do K2B_NSEG=$NS K2_REPS=200 timeout 100 python tools/perf_k2b.py 2>/dev/null | tail -2 | sed "s/^/nseg=$NS: /"; done | tee gpurun_out/r05j_k2b_nseg.txt
for NS in 8 4; do K2_B=8 K2_HT=0 K2_NSEG=$NS K2_REPS=300 timeout 100 python tools/perf_k2.py 2>/dev/null | tail -1 | sed "s/^/fwd nseg=$NS: /"; done | tee -a gpurun_out/r05j_k2b_nseg.txt
LINA_PARITY_TAG=r05 timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest=$?"; tail -5 gpurun_out/r05_pytest_gpu.log
