#!/bin/bash
# round 3, call i: bench under a live RCCL process group (1 GPU), K2 after the variant clean-up
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bench_launch.py -m gpu -q -x 2>&1 | tail -40 | cut -c1-400
for r in 1 2; do K2_HT=0 K2_REPS=3000 python tools/perf_k2.py | tail -1; done
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "chunk" 2>&1 | tail -2
