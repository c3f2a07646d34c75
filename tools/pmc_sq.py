#!/usr/bin/env python
"""Collect the SQ counter passes of tests/gpu_r05_evidence.sh into one JSON: per kernel, the mean of every counter over the
dispatches of its pass (the counters are SUMS over all waves / SIMDs of a launch) + the derived shares the docs quote:
  mfma_util      = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)        (matrix-pipe busy cycles summed over the 1024 SIMDs / (elapsed
                   cycles x 1024 SIMDs); SQ_BUSY_CYCLES is summed over the chip's 32 SQ instances (8 XCDs x 4 shader engines), so
                   elapsed = SQ_BUSY_CYCLES / 32 for a kernel that keeps every engine busy, and 1024 / 32 = 32.  Same figure as the
                   gfx94x derived metric MfmaUtil = MFMA_BUSY / (GRBM_GUI_ACTIVE x CUs x 4).  Cross-check on K2: 18.9 M MFMAs x 16 cycles)
  wait_share     = SQ_WAIT_ANY / SQ_WAVE_CYCLES                           (wave parked: s_waitcnt / s_barrier)
  issue_stall    = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
    python tools/pmc_sq.py out.json '<label>|<kernel name pattern>' ... -- <pass dir> <pass dir> ..."""
import csv, glob, json, sys

args = sys.argv[2:]
cut = args.index("--")
pats = dict(a.split("|", 1) for a in args[:cut])
res = {k: {} for k in pats}
for d in args[cut + 1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            for label, pat in pats.items():
                if pat in r.get("Kernel_Name", ""):
                    acc.setdefault((label, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (label, c), v in acc.items():
            res[label][c] = {"n": len(v), "mean": sum(v) / len(v)}
for label, d in res.items():
    g = lambda c: d.get(c, {}).get("mean")
    der = {}
    if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("SQ_BUSY_CYCLES"):
        der["mfma_util"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (32.0 * g("SQ_BUSY_CYCLES"))
    if g("SQ_WAVE_CYCLES"):
        for name, c in (("wait_share", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"), ("active_share", "SQ_ACTIVE_INST_ANY"),
                        ("valu_share", "SQ_ACTIVE_INST_VALU"), ("lds_share", "SQ_ACTIVE_INST_LDS")):
            if g(c) is not None:
                der[name] = g(c) / g("SQ_WAVE_CYCLES")
    if g("SQ_LDS_IDX_ACTIVE"):
        der["lds_conflict"] = (g("SQ_LDS_BANK_CONFLICT") or 0.0) / g("SQ_LDS_IDX_ACTIVE")
    d["derived"] = der
json.dump({"what": "rocprofv3 --kernel-trace --pmc passes (SQ counters only, one set per pass): means per launch; derived shares "
                   "as defined in tools/pmc_sq.py", "kernels": res}, open(sys.argv[1], "w"), indent=1)
for label, d in res.items():
    print(label, {k: round(v, 4) for k, v in d["derived"].items()})
