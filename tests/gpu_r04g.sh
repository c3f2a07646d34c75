#!/bin/bash
# round 4: time stamps inside the one-launch mixer + the A/B of the loop
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/probe_iw_prof.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04g_iw_prof.txt
timeout 600 python tools/probe_one_launch.py 2>&1 | grep -v "amdgpu.ids\|^tokens of" | tee gpurun_out/r04g_probe.txt
