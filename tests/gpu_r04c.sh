#!/bin/bash
# round 4, session c: the model-level GPU suite with the fused cross-attention tail + the fixed config-4 / config-5 tests; decode bench
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
LINA_PARITY_TAG=r04c timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -q --timeout=600 -k "not chunk and not vocoder and not config3 and not conv and not rmsnorm" > gpurun_out/r04c_pytest.log 2>&1; echo "pytest=$?"; tail -8 gpurun_out/r04c_pytest.log
for i in 1 2; do
  timeout 300 python bench.py --no-chunk --no-train --no-cpu-baseline > gpurun_out/r04c_bench_$i.json 2> gpurun_out/r04c_bench_$i.err; python tools/bench_summary.py gpurun_out/r04c_bench_$i.json | head -1
done
