#!/bin/bash
# round 6 session F: two engines as the default at 512 rows -- bit-equality test, the bench line with it
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "two_engines or picks_two" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-train --no-cpu-baseline --no-chunk > gpurun_out/r06_f_bench.json 2> gpurun_out/r06_f_bench.err; echo "bench=$?"; tail -3 gpurun_out/r06_f_bench.err
python tools/bench_summary.py gpurun_out/r06_f_bench.json 2>&1 | head -40
