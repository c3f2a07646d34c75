#!/bin/bash
# K2b on the full-head kernel: HBM traffic counters and SQ activity per sweep (separate --pmc passes; MI355X_MICROARCH.md).
# B = 64 (one workgroup per head) and b = 8 (8 segments).  Output -> gpurun_out/<tag>_k2bfull_counters.json
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02}
run_pmc() {  # batch, name, counters...
  local b=$1 name=$2; shift; shift
  K2_B=$b K2_REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/k2bf_${b}_$name -o ${TAG} --output-format csv -- python tools/perf_k2b.py > gpurun_out/${TAG}_k2bfull_${b}_$name.log 2>&1; echo "$b $name=$?"
}
for b in 64 8; do
  run_pmc $b fetch FETCH_SIZE
  run_pmc $b write WRITE_SIZE
done
run_pmc 64 sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run_pmc 64 sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU
python - "$TAG" <<'PY'
import csv, glob, json, sys, collections, re
tag = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob("/tmp/k2bf_*"):
    b = d.split("_")[1]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            m = re.search(r"gla_chunk_bf16_h256_kernel<([^>]*)>", k)
            key = m.group(1).replace(" ", "") if m else ("combine" if "combine" in k else None)
            if key:
                res[f"B{b}:{key}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in sorted(res.items())}
json.dump(out, open(f"gpurun_out/{tag}_k2bfull_counters.json", "w"), indent=1)
for k, d in out.items():
    print(k, "  ".join(f"{c}={v['mean']:.4g}" for c, v in d.items()))
PY
