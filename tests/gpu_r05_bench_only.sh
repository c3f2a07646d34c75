#!/bin/bash
# the driver's bench command on its own (after the capture guard: a dropped engine's hipGraphs must not be destroyed inside another capture)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; echo "bench=$?"; tail -2 gpurun_out/r05_bench.err
python tools/bench_summary.py gpurun_out/r05_bench.json
