#!/bin/bash
# round 5, session M: tall kernels after the loop restructuring (no accumulator copies, statistics on the matrix pipe) -- parity, micro timing, loop
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout=600 -x -k "tall" > gpurun_out/r05m_pytest.log 2>&1; echo "pytest=$?"; tail -2 gpurun_out/r05m_pytest.log
for TV in 0 1; do LINA_TALL_V=$TV timeout 60 python tools/perf_tall.py 512 40 2>/dev/null; done | tee gpurun_out/r05m_perf_tall.txt
for MM in 256 384; do LINA_TALL=1 LINA_TALL_V=0 timeout 60 python tools/perf_tall.py $MM 40 2>/dev/null; LINA_TALL=0 timeout 60 python tools/perf_tall.py $MM 40 2>/dev/null; done | tee -a gpurun_out/r05m_perf_tall.txt
timeout 300 python tools/perf_loop.py 512 2>/dev/null | tee -a gpurun_out/r05m_perf_tall.txt
