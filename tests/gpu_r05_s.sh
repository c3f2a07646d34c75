#!/bin/bash
# round 5, session S: two / four independent engines on separate HIP streams vs one engine of all rows
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ timeout 200 python tools/probe_two_engines.py 256 2; timeout 200 python tools/probe_two_engines.py 128 4; timeout 200 python tools/perf_loop.py 512; } 2>/dev/null | tee gpurun_out/r05s_two_engines.txt
