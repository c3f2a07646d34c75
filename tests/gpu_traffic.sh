#!/bin/bash
# HBM traffic of the dominant kernel of the decode step (K1w + K5) from the L2 fabric counters, one counter per pass
# (guide: MI355X_MICROARCH.md, HBM: separate --pmc passes, FETCH_SIZE doubled on gfx950 for 16-B/lane streaming reads).
# Writes gpurun_out/<tag>_k1w_traffic.json (copy to profiles/).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02}
for C in FETCH_SIZE WRITE_SIZE; do
  K1_REPS=16 timeout 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/${TAG}_pmc_$C -o ${TAG} --output-format csv -- python tools/perf_k1w.py > gpurun_out/${TAG}_pmc_$C.log 2>&1; echo "$C=$?"
done
python - "$TAG" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/{tag}_pmc_{c}/**/*counter_collection.csv", recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if "gla_decode_window_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == c]
    v = [float(r["Counter_Value"]) for r in rows]
    v.sort()
    out[c] = {"dispatches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": v[0], "max_KiB": v[-1],
              "median_KiB": v[len(v) // 2]}
    print(c, out[c])
rd = 2 * out["FETCH_SIZE"]["mean_KiB"] * 1024          # gfx950: 128-B streaming read requests tallied at 64 B
wr = out["WRITE_SIZE"]["mean_KiB"] * 1024
res = {"kernel": "lina::gla_decode_window_kernel<256, 4, bf16, float> (K1w + K5, window 8)",
       "command": "rocprofv3 --kernel-trace --pmc <COUNTER> -- python tools/perf_k1w.py (the 13 layers' real buffers, all 8 window positions in turn; one counter per pass; tests/gpu_traffic.sh)",
       "counters": out,
       "correction": "gfx950 FETCH_SIZE counts the 128-B requests of a 16-B/lane streaming read at 64 B: doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE taken as is",
       "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
       "note": "mean over all launches of the run = all 8 window positions (7 read-only, 1 write-back)"}
json.dump(res, open(f"gpurun_out/{tag}_k1w_traffic.json", "w"), indent=1)
print(json.dumps(res)[:600])
PY
rm -rf gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE
