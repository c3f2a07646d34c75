#!/bin/bash
# round 3, call q: where does the projections' post-barrier time go (slot 7 = partial sums added)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/probe_skinny_prof.py 2>&1 | grep -v amdgpu.ids | head -34 | tee gpurun_out/r03q_skprof.log
