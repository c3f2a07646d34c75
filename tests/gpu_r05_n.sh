#!/bin/bash
# round 5, session N: variant 2 of the tall kernels (weights through the LDS ring, A through a register ring)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
LINA_TALL_V=2 timeout 600 python -m pytest tests -m gpu -q --timeout=600 -x -k "tall" > gpurun_out/r05n_pytest.log 2>&1; echo "pytest=$?"; tail -2 gpurun_out/r05n_pytest.log
for TV in 0 2; do LINA_TALL_V=$TV timeout 60 python tools/perf_tall.py 512 40 2>/dev/null; done | tee gpurun_out/r05n_perf_tall.txt
for MM in 256; do LINA_TALL=1 LINA_TALL_V=2 timeout 60 python tools/perf_tall.py $MM 40 2>/dev/null; done | tee -a gpurun_out/r05n_perf_tall.txt
LINA_TALL_V=2 timeout 300 python tools/perf_loop.py 512 2>/dev/null | tee -a gpurun_out/r05n_perf_tall.txt
