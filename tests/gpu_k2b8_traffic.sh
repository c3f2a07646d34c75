#!/bin/bash
# HBM traffic of the segment-parallel K2 forward at the training micro-batch (b=8, 8 segments): FETCH_SIZE / WRITE_SIZE per kernel
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  K2_B=8 K2_HT=0 K2_REPS=4 timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/k2b8_$c -o k2 --output-format csv -- python tools/perf_k2.py > gpurun_out/r02_k2_b8_$c.log 2>&1; echo "$c=$?"
done
python - <<'PY'
import csv, glob, json, collections, re
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/k2b8_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            m = re.search(r"gla_chunk_bf16_h256_kernel<([^>]*)>", k)
            key = m.group(1).replace(" ", "") if m else ("combine" if "combine" in k else None)
            if key:
                res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"n": len(v), "mean_KiB": sum(v) / len(v)} for c, v in d.items()} for k, d in res.items()}
json.dump(out, open("gpurun_out/r02_k2_b8_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
