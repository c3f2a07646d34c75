#!/bin/bash
# round 6 session C: the 128-row fused in-projection (variant 3): parity on the GPU, per-launch time, the decode loop with / without it
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "inproj_tall" > gpurun_out/r06_c_pytest.log 2>&1; echo "pytest=$?"; tail -3 gpurun_out/r06_c_pytest.log
for v in 0 3; do
  LINA_TALL_V=$v timeout 200 python tools/perf_tall.py 512 200 2>&1 | grep inproj >> gpurun_out/r06_tall_v3.txt
  LINA_TALL_V=$v timeout 200 python tools/perf_tall.py 384 200 2>&1 | grep inproj >> gpurun_out/r06_tall_v3.txt
done
cat gpurun_out/r06_tall_v3.txt
for v in 0 3; do
  LINA_TALL_V=$v timeout 300 python tools/perf_loop.py 512 400 2>&1 | tail -1 | sed "s/^/V=$v /" >> gpurun_out/r06_loop_v3.txt
done
LINA_TALL_V=3 timeout 300 python tools/perf_loop.py 384 400 2>&1 | tail -1 | sed "s/^/V=3 /" >> gpurun_out/r06_loop_v3.txt
LINA_TALL_V=0 timeout 300 python tools/perf_loop.py 384 400 2>&1 | tail -1 | sed "s/^/V=0 /" >> gpurun_out/r06_loop_v3.txt
cat gpurun_out/r06_loop_v3.txt
