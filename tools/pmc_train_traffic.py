#!/usr/bin/env python
"""HBM-side traffic of the train step's memory-bound kernels from two rocprofv3 --pmc passes over tools/perf_train_step.py
(FETCH_SIZE, WRITE_SIZE: one counter per pass), against the bytes each launch must read / write once at L169, b = 8 x 4096.
FETCH_SIZE is printed raw AND doubled: on gfx950 it reports half the bytes of a wide coalesced streaming read
(MI355X_MICROARCH.md, HBM); other access widths are uncalibrated -- the contiguous 8-byte-per-lane kernels below (SwiGLU forward:
one pass over u) calibrate it for this pattern.
    python tools/pmc_train_traffic.py <fetch_dir> <write_dir> [out.txt]"""
import csv
import glob
import sys

N = 32768
ZW = 4160                                                             # row width of the stacked projection (mixer.py)
ALG = {   # kernel name fragment -> (label, read bytes, write bytes) per launch
    "short_conv_fwd_kernel": ("K3 conv fwd q|k|v", N * 3072 * 2, N * 3072 * 2),
    "short_conv_bwd_kernel": ("K3b conv bwd", 2 * N * 3072 * 2, N * 3072 * 2),
    "rmsnorm_gate_kernel": ("K5 norm-gate fwd", 2 * N * 1024 * 2, N * 1024 * 2),
    "rmsnorm_gate_bwd_kernel": ("K5b norm-gate bwd", 3 * N * 1024 * 2, 2 * N * 1024 * 2),
    "layernorm_fwd_kernel<float": ("K10 LayerNorm + residual fwd", N * 1024 * (4 + 2), N * 1024 * (4 + 2)),
    "layernorm_bwd_kernel<float": ("K10b LayerNorm bwd", N * 1024 * (2 + 4 + 4), N * 1024 * (4 + 2)),
    "swiglu_rows_kernel": ("K11 SwiGLU fwd", N * 2816 * 2, N * 1408 * 2),
    "swiglu_bwd_colsum_kernel": ("K11c SwiGLU bwd", N * (1408 + 2816) * 2, N * 2816 * 2),
    "gate_lowrank_mfma_kernel<false>": ("K12c gate fwd", N * 16 * 2, N * 1024 * 2),
    "gate_lowrank_mfma_kernel<true>": ("K12c gate bwd", N * (16 + 1024) * 2, N * 1024 * 2),
    "cross_entropy_kernel<unsigned short, 17, false>": ("K14 CE fwd", N * 4099 * 2, N * 8),
    "cross_entropy_kernel<unsigned short, 17, true>": ("K14 CE bwd", N * 4099 * 2, N * 4099 * 2),
    "gla_chunk_bf16_h256_kernel<false, 1, 0, false": ("K2 full pass (8 segments)", 4 * N * 1024 * 2, N * 1024 * 2),
    "gla_chunk_bf16_h256_kernel<true, 1, 0, false": ("K2 state-only pass", 3 * N * 1024 * 2, 0),
}


def load(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    by = {}
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") == counter:
            by.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return by


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
lines = [f"{'kernel':34s} {'n':>5s} {'read MB':>8s} {'FETCH MB':>9s} {'x2':>7s} {'x2/read':>8s} {'write MB':>9s} {'WRITE MB':>9s} {'/write':>7s}"]
for key, (label, rb, wb) in ALG.items():
    ks = [k for k in fe if key in k and (key != "rmsnorm_gate_kernel" or "bwd" not in k)]
    if not ks:
        continue
    f = [v for k in ks for v in fe[k]]
    w = [v for k in ks for v in wr.get(k, [])]
    if key.startswith("layernorm") or key.startswith("swiglu_bwd"):   # the text encoder's smaller calls: keep the big ones
        f = [v for v in f if v > 0.5 * max(f)]
        w = [v for v in w if v > 0.5 * max(w)]
    fm, wm = sum(f) / len(f) * 1024, (sum(w) / len(w) * 1024 if w else 0.0)
    lines.append(f"{label:34s} {len(f):5d} {rb / 1e6:8.1f} {fm / 1e6:9.1f} {2 * fm / 1e6:7.1f} {2 * fm / rb:8.2f} {wb / 1e6:9.1f} {wm / 1e6:9.1f} "
                 f"{(wm / wb if wb else 0):7.2f}")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(__doc__.split("\n    python")[0] + "\n\n" + out + "\n")
