#!/bin/bash
# round 4: eight-wave K1w -- parity (bit-identical to the sixteen-wave kernel) and A/B of the decode loop
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "eight_waves or decode_window" --timeout=300 2>&1 | tail -3
timeout 600 python tools/probe_k1w_waves.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04h_k1w_waves.txt
