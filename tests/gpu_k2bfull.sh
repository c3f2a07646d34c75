cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "chunk_bwd or chunk" 2>&1 | tail -8
for ns in "" 1 4 16; do K2B_NSEG=$ns timeout 120 python tools/perf_k2b.py; done
K2B_PATH=sweeps timeout 120 python tools/perf_k2b.py
K2_B=64 K2_REPS=50 timeout 120 python tools/perf_k2b.py
K2_B=64 K2_REPS=20 K2B_PATH=sweeps timeout 120 python tools/perf_k2b.py
