cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${TAG:-r02e}
if [ -n "${KEXPR}" ]; then
timeout 600 python -m pytest tests -m gpu -q --timeout=600 --durations=8 -k "${KEXPR}" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest=$?"; tail -12 gpurun_out/${T}_pytest.log
fi
timeout 300 python bench.py ${BENCH_ARGS:---no-train --no-cpu-baseline --no-chunk} > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench=$?"; python -c "
import json;j=json.load(open('gpurun_out/${T}_bench.json'));r=j['roofline'];print('tok/s',j['value'],'ms',j['ms_per_step'],'k1 us',r['us_per_launch'],'frac',r['frac'],'step frac',j['step_roofline']['frac'])"; tail -2 gpurun_out/${T}_bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof -o ${T} -- python bench.py --steps 400 --warmup 10 --no-cpu-baseline --no-train --no-chunk > gpurun_out/${T}_bench_prof.json 2> gpurun_out/${T}_bench_prof.err; echo "prof=$?"
db=$(find gpurun_out/${T}_prof -name "*results.db" | head -1); [ -n "$db" ] && python tools/prof_summary.py "$db" gpurun_out/${T}_kernel_stats.csv && python - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/${T}_kernel_stats.csv')):
    n=r['Name']
    if int(r['Calls'])>300: print(f"{n.split('(')[0][:80]:82s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:7.2f} min {float(r['MinNs'])/1e3:7.2f} max {float(r['MaxNs'])/1e3:7.2f}")
PY
rm -rf gpurun_out/${T}_prof
