#!/bin/bash
# round 6 session H: the opt-in bf16 recurrent state -- kernel + engine parity on the GPU, the bench block
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
LINA_PARITY_TAG=r06h timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "bf16_state or decode_window" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-train --no-cpu-baseline > gpurun_out/r06_h_bench.json 2> gpurun_out/r06_h_bench.err; echo "bench=$?"; tail -3 gpurun_out/r06_h_bench.err
python tools/bench_summary.py gpurun_out/r06_h_bench.json 2>&1 | head -40
