#!/bin/bash
# round 4: where do the workgroups of the one-launch mixer land? (dispatcher census) + kernel durations of the A/B probe
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp

timeout 600 python tools/probe_one_launch.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04f_probe.txt
