#!/bin/bash
# round 5, session F: HBM-side bytes (FETCH_SIZE) and L2 hit / miss of the tall projection kernels, micro script (few launches)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for TV in 0 1; do LINA_TALL_V=$TV timeout 60 python tools/perf_tall.py 512 40 2>/dev/null; done | tee gpurun_out/r05f_perf_tall.txt
LINA_TALL=0 timeout 60 python tools/perf_tall.py 512 40 2>/dev/null | tee -a gpurun_out/r05f_perf_tall.txt
for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  rm -rf /tmp/pm; LINA_TALL_V=0 timeout 120 rocprofv3 --kernel-trace --pmc $C -d /tmp/pm -o pm --output-format csv -- python tools/perf_tall.py 512 4 > gpurun_out/r05f_$T.log 2>&1; echo "$T=$?"
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' | tee gpurun_out/r05f_pmc_$T.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values()))[:8]:
    print(k.ljust(72), "  ".join(f"{c}: n={len(v)} mean={sum(v)/len(v):.1f}" for c, v in d.items()))
PY
done
