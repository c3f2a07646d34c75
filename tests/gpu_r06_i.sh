#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "128 2" "192 2" "256 2"; do timeout 300 python tools/probe_two_engines.py $cfg 2>&1 | tail -1; done
for b in 256 384; do timeout 300 python tools/perf_loop.py $b 400 2>&1 | tail -1; done
