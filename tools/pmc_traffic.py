#!/usr/bin/env python
"""HBM traffic of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, tests/gpu_evidence.sh),
corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes for gfx950: FETCH_SIZE tallies the 128-byte requests of a
16-B/lane streaming read at 64 B -> doubled; WRITE_SIZE as reported.  Writes a JSON summary.

    python tools/pmc_traffic.py k1w <fetch_dir> <write_dir> out.json [rows per launch, default 64]
    python tools/pmc_traffic.py k2  <fetch_dir> <write_dir> out.json <heads> [kernel_stats.csv]
    python tools/pmc_traffic.py k2b <fetch_dir> <write_dir> out.json <heads>      (sum over the sweeps of one backward call)
"""
import csv, glob, json, os, sys

BK = int(os.environ.get("PMC_B", 64))        # batch rows of the K2 / K2b shape the passes ran (tools/perf_k2.py K2_B)

which, fdir, wdir, out = sys.argv[1:5]
pat = {"k1w": "gla_decode_window_kernel", "k2": "gla_chunk_bf16_h256", "k2b": "gla_chunk_bf16_h256", "k2seg": "gla_",
       "k2dv512": "gla_chunk_bf16_h256", "k2dv512one": "gla_chunk_bf16_h256"}[which]
res = {}
for c, d in (("FETCH_SIZE", fdir), ("WRITE_SIZE", wdir)):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if pat in r.get("Kernel_Name", "") and r.get("Counter_Name") == c]
    v = sorted(float(r["Counter_Value"]) for r in rows)
    res[c] = {"dispatches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": v[0], "max_KiB": v[-1], "median_KiB": v[len(v) // 2]}
    if which in ("k2b", "k2seg"):            # one backward call = one dispatch of every sweep instantiation: add their means
        by = {}
        for r in rows:
            by.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        calls = min(len(x) for x in by.values())
        res[c]["per_call_KiB"] = sum(sum(x) for x in by.values()) / calls
        res[c]["sweeps"] = {k[:90]: {"dispatches": len(x), "mean_KiB": sum(x) / len(x)} for k, x in by.items()}
rd, wr = 2 * res["FETCH_SIZE"]["mean_KiB"] * 1024, res["WRITE_SIZE"]["mean_KiB"] * 1024
o = {"counters": res,
     "correction": "gfx950 FETCH_SIZE counts the 128-B requests of a 16-B/lane streaming read at 64 B: doubled "
                   "(MI355X_MICROARCH.md, HBM); WRITE_SIZE taken as is; one counter per rocprofv3 pass, --kernel-trace only",
     "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr}
if which == "k2dv512":     # expand_v = 2: one call = two launches of the 256-column kernel (q, k, g read by both)
    rd, wr = 2 * rd, 2 * wr
    heads = int(sys.argv[5])
    D = 1024 // heads
    alg = BK * heads * 4096 * 2 * (3 * D + 2 * 2 * D)
    o.update(hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, traffic_bytes_per_launch=rd + wr,
             kernel="lina::gla_chunk_bf16_h256_kernel<false, 1> x 2 (one launch per 256-column block of v / o), per CALL",
             shape={"B": BK, "H": heads, "T": 4096, "Dk": D, "Dv": 2 * D},
             command=f"K2_B={BK} K2_H={heads} K2_DV={2 * D} K2_HT=0 rocprofv3 --kernel-trace --pmc <COUNTER> -- python tools/perf_k2.py (tests/gpu_r06_evidence.sh)",
             algorithmic_bytes_per_launch=alg, traffic_over_algorithmic=(rd + wr) / alg)
elif which == "k2dv512one":  # expand_v = 2 since round 6: ONE launch, two workgroups per head paired on an XCD
    heads = int(sys.argv[5])
    D = 1024 // heads
    alg = BK * heads * 4096 * 2 * (3 * D + 2 * 2 * D)
    o.update(kernel="lina::gla_chunk_bf16_h256_kernel<false, 1, 0, false, false, false, NCB = 2> (both 256-column blocks of v / o in one launch)",
             shape={"B": BK, "H": heads, "T": 4096, "Dk": D, "Dv": 2 * D},
             command=f"K2_B={BK} K2_H={heads} K2_DV={2 * D} K2_HT=0 rocprofv3 --kernel-trace --pmc <COUNTER> -- python tools/perf_k2.py (tests/gpu_r06_dv512.sh)",
             algorithmic_bytes_per_launch=alg, traffic_over_algorithmic=(rd + wr) / alg)
elif which == "k2seg":       # segment-parallel forward: state-only pass + combine + full pass, summed per call
    rd, wr = 2 * res["FETCH_SIZE"]["per_call_KiB"] * 1024, res["WRITE_SIZE"]["per_call_KiB"] * 1024
    heads = int(sys.argv[5])
    D = 1024 // heads
    alg = BK * heads * 4096 * 2 * 5 * D
    o.update(hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, traffic_bytes_per_launch=rd + wr,
             kernel="lina_gla_chunk_fwd_seg: state-only pass + combine + full pass (segment-parallel K2 forward), per CALL",
             shape={"B": BK, "H": heads, "T": 4096, "Dk": D, "Dv": D},
             command=f"K2_B={BK} K2_HT=0 rocprofv3 --kernel-trace --pmc <COUNTER> -- python tools/perf_k2.py (tests/gpu_r05_evidence.sh)",
             algorithmic_bytes_per_launch=alg, traffic_over_algorithmic=(rd + wr) / alg)
elif which == "k2b":
    rd, wr = 2 * res["FETCH_SIZE"]["per_call_KiB"] * 1024, res["WRITE_SIZE"]["per_call_KiB"] * 1024
    heads = int(sys.argv[5])
    D = 1024 // heads
    alg = BK * heads * 4096 * 2 * 9 * D
    o.update(hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, traffic_bytes_per_launch=rd + wr,
             kernel="lina_gla_chunk_bwd_full: gla_chunk_bf16_h256_kernel<MODE, REV, DG> x 3 sweeps (K2b), per backward CALL",
             shape={"B": BK, "H": heads, "T": 4096, "Dk": D, "Dv": D},
             command=f"K2_BWD=1 K2_H={heads} rocprofv3 --kernel-trace --pmc <COUNTER> -- python tools/perf_k2.py (tests/gpu_evidence.sh)",
             note="algorithmic = q, k, v, g, do in + dq, dk, dv, dg out (9 tensor passes); the three sweeps make 18",
             algorithmic_bytes_per_launch=alg, traffic_over_algorithmic=(rd + wr) / alg)
elif which == "k1w":
    B, H, Dk, Dv, W = (int(sys.argv[5]) if len(sys.argv) > 5 else 64), 4, 256, 256, 8
    alg = int(B * (4 * H * Dk * Dv * (1 + 1 / W) + 2 * (2 * H * Dk + 2 * H * Dv) + 4 * H * Dk + 4 * H * (2 * Dk + Dv)
                   + 4 * H * (2 * Dk + Dv) * (W - 1) / 2))
    o.update(kernel="lina::gla_decode_window_kernel<256, 4, 1, bf16, float> (K1w + K5, window 8)",
             command="rocprofv3 --kernel-trace --pmc <COUNTER> -- python tools/perf_k1w.py (the 13 layers' real buffers, all 8 "
                     "window positions in turn; tests/gpu_evidence.sh)",
             note="mean over all launches of the run = all 8 window positions (7 read-only, 1 write-back)", rows_per_launch=B,
             algorithmic_bytes_per_launch=alg, traffic_over_algorithmic=(rd + wr) / alg)
else:
    heads = int(sys.argv[5])
    D = 1024 // heads
    alg = BK * heads * 4096 * 2 * 5 * D
    o.update(kernel=f"lina::gla_chunk_bf16_h256_kernel<false, G={256 // D}> (K2 forward, training call: no final state)",
             shape={"B": BK, "H": heads, "T": 4096, "Dk": D, "Dv": D},
             command=f"K2_B={BK} K2_H={heads} K2_HT=0 rocprofv3 --kernel-trace --pmc <COUNTER> -- python tools/perf_k2.py (tests/gpu_evidence.sh)",
             algorithmic_bytes_per_launch=alg, traffic_over_algorithmic=(rd + wr) / alg)
    if len(sys.argv) > 6:
        rows = [r for r in csv.DictReader(open(sys.argv[6])) if pat in r.get("name", r.get("Name", ""))]
        if rows:
            o["kernel_stats_row"] = rows[0]
json.dump(o, open(out, "w"), indent=1)
print(out, "traffic/algorithmic =", round(o["traffic_over_algorithmic"], 4))
