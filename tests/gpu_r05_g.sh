#!/bin/bash
# round 5, session G: tall kernels with the XCD-aware tile order -- micro timing, FETCH_SIZE, then the decode loop
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for TV in 0 1; do LINA_TALL_V=$TV timeout 60 python tools/perf_tall.py 512 40 2>/dev/null; done | tee gpurun_out/r05g_perf_tall.txt
for MM in 256 128; do for TL in 0 1; do LINA_TALL=$TL timeout 60 python tools/perf_tall.py $MM 40 2>/dev/null; done; done | tee -a gpurun_out/r05g_perf_tall.txt
rm -rf /tmp/pm; LINA_TALL_V=0 timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pm -o pm --output-format csv -- python tools/perf_tall.py 512 4 > gpurun_out/r05g_FETCH.log 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee gpurun_out/r05g_pmc_FETCH_SIZE.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values()))[:4]:
    print(k.ljust(72), "  ".join(f"{c}: n={len(v)} mean={sum(v)/len(v):.1f}" for c, v in d.items()))
PY
for BB in 512; do timeout 300 python tools/perf_loop.py $BB 2>/dev/null | tee -a gpurun_out/r05g_perf_tall.txt; done
