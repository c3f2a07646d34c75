#!/bin/bash
# round 5, session Q: tall parity on the final thresholds; decode loop at B = 512 / 256 / 192 / 128; K1w state window 8 vs 16 at B = 512 / 256
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout=600 -x -k "tall or config4" > gpurun_out/r05q_pytest.log 2>&1; echo "pytest=$?"; tail -2 gpurun_out/r05q_pytest.log
for BB in 512 256 192 128; do timeout 300 python tools/perf_loop.py $BB 400 2>/dev/null; done | tee gpurun_out/r05q_loop.txt
for BB in 512 256; do timeout 300 python tools/perf_loop.py $BB 400 16 2>/dev/null; done | tee -a gpurun_out/r05q_loop.txt
for BB in 512; do timeout 300 python tools/perf_loop.py $BB 400 4 2>/dev/null; done | tee -a gpurun_out/r05q_loop.txt
