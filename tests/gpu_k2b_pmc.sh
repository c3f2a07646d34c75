#!/bin/bash
# K2b (chunk backward) alone at the training shape b=8,H=4,T=4096: kernel stats under settled clocks, HBM traffic
# counters and SQ activity counters (separate --pmc passes; MI355X_MICROARCH.md).  Outputs -> gpurun_out/<tag>_k2b_*
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02}
K2_REPS=300 timeout 120 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_k2bprof -o ${TAG} -- python tools/perf_k2b.py > gpurun_out/${TAG}_k2b_prof.log 2>&1; echo "stats=$?"; tail -1 gpurun_out/${TAG}_k2b_prof.log
db=$(find gpurun_out/${TAG}_k2bprof -name "*results.db" | head -1); [ -n "$db" ] && python tools/prof_summary.py "$db" gpurun_out/${TAG}_k2b_kernel_stats.csv; rm -rf gpurun_out/${TAG}_k2bprof
run_pmc() {  # name, counters...
  local name=$1; shift
  K2_REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/${TAG}_k2b_$name -o ${TAG} --output-format csv -- python tools/perf_k2b.py > gpurun_out/${TAG}_k2b_$name.log 2>&1; echo "$name=$?"
}
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run_pmc sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SALU
python - "$TAG" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("fetch", "write", "sq1", "sq2"):
    for f in glob.glob(f"gpurun_out/{tag}_k2b_{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            for key in ("gla_bwd_sweeps", "gla_bwd_dg_totals", "gla_bwd_dg_final"):
                if key in k:
                    res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in res.items()}
json.dump(out, open(f"gpurun_out/{tag}_k2b_counters.json", "w"), indent=1)
for k, d in out.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:32s} n={v['n']:3d} mean={v['mean']:.4g}")
PY
for n in fetch write sq1 sq2; do rm -rf gpurun_out/${TAG}_k2b_$n; done
