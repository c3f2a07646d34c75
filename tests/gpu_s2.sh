cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${TAG:-r02c}
timeout 600 python -m pytest tests -m gpu -q --timeout=600 --durations=8 -k "${KEXPR:-decode_window}" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest=$?"; tail -15 gpurun_out/${T}_pytest.log
for W in ${WINDOWS:-8}; do
timeout 300 python bench.py --window $W --no-train --no-cpu-baseline --no-chunk > gpurun_out/${T}_bench_w$W.json 2> gpurun_out/${T}_bench_w$W.err; echo "bench w$W=$?"; python -c "
import json;j=json.load(open('gpurun_out/${T}_bench_w$W.json'));r=j['roofline'];print(j['value'],j['ms_per_step'],r['us_per_launch'],r['frac'],r.get('immediate_form',{}).get('equivalent_GBs'),j['step_roofline']['frac'])"; tail -2 gpurun_out/${T}_bench_w$W.err
done
