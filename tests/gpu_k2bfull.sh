cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "chunk or train or config5 or grad" 2>&1 | tail -4
timeout 120 python tools/perf_k2b.py
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02r_bench.json 2> gpurun_out/r02r_bench.err; tail -3 gpurun_out/r02r_bench.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r02r_bench.json"))
print(j["value"], j["ms_per_step"])
for k in ("chunk_bwd_kernel","chunk_bwd_kernel_b64","train_step"):
    d=j.get(k,{}); print(k, {x:d[x] for x in d if x in ("ms","frac","ms_per_step","tokens_per_s","step_ms")})
PY
