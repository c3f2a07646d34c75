#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 3 4 0 3 4; do LINA_TALL_V=$v timeout 300 python tools/perf_loop.py 512 400 2>&1 | tail -1 | sed "s/^/V=$v /"; done
for v in 0 3 4; do LINA_TALL_V=$v timeout 300 python tools/perf_loop.py 384 400 2>&1 | tail -1 | sed "s/^/V=$v /"; done
