#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
K2_HT=0 K2_REPS=200 timeout 200 python tools/perf_k2.py 2>&1 | tail -1
LINA_GLA_LIB=tools/abl/liblina_k2prof.so K2_PROF=1 K2_HT=0 K2_REPS=20 timeout 200 python tools/perf_k2.py 2>&1 | tail -20
