#!/bin/bash
# round 5, session P: row threshold of the tall kernels with 64-row workgroups: tall vs 64-row split-K kernels at M = 128 .. 512
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for MM in 128 192 256 384 512; do for TL in 0 1; do LINA_TALL=$TL timeout 60 python tools/perf_tall.py $MM 40 2>/dev/null; done; done | tee gpurun_out/r05p_threshold.txt
