#!/bin/bash
# round 4: the one-launch mixer (in-projection + K1w + K5): kernel parity, A/B of the decode loop, launch timeline
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "inproj_window or decode_window or inproj" --timeout=300 > gpurun_out/r04e_pytest.log 2>&1; echo "pytest=$?"; tail -5 gpurun_out/r04e_pytest.log
timeout 600 python tools/probe_one_launch.py > gpurun_out/r04e_probe.txt 2>&1; echo "probe=$?"; cat gpurun_out/r04e_probe.txt | grep -v amdgpu.ids
