cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 --durations=15 -k "decode_window or bf16_engine or chunk_bwd_at or chunk_bwd_long or engine_device_side or smoke" > gpurun_out/r02b_pytest.log 2>&1; echo "pytest=$?"; tail -32 gpurun_out/r02b_pytest.log
tools/micro/load_pattern2 > gpurun_out/r02b_load_pattern2.log 2>&1; cat gpurun_out/r02b_load_pattern2.log
for W in 8 1 4; do
timeout 300 python bench.py --window $W --no-train --no-cpu-baseline --no-chunk > gpurun_out/r02b_bench_w$W.json 2> gpurun_out/r02b_bench_w$W.err; echo "bench w$W=$?"; python -c "
import json;j=json.load(open('gpurun_out/r02b_bench_w$W.json'));print(j['value'],j['ms_per_step'],j['roofline'],j['step_roofline']['frac'])"; tail -3 gpurun_out/r02b_bench_w$W.err
done
