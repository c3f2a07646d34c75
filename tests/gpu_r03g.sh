#!/bin/bash
# round 3, call g: full GPU suite (wave-uniform fix), parity record, the full bench line
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
LINA_PARITY_TAG=r03g timeout 1500 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/r03g_pytest_gpu.log 2>&1; echo "pytest=$?"; tail -12 gpurun_out/r03g_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err; echo "bench=$?"; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r03g_bench.json").read().strip().splitlines()[-1])
keep = {k: j[k] for k in ("value", "ms_per_step", "steps", "sustained", "sampled_decode") if k in j}
print(json.dumps(keep))
print("roofline", {k: j["roofline"][k] for k in ("frac", "us_per_launch", "frac_back_to_back")})
print("step_roofline", j["step_roofline"]["frac"], "chunk", j.get("chunk_kernel", {}).get("frac"), "train", j.get("train_step"))
print("cpu", j.get("cpu_baseline"))
PY
tail -3 gpurun_out/r03g_bench.err
