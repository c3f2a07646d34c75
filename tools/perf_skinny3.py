#!/usr/bin/env python
"""Tile sweep of lina_linear_skinny at the decode shapes (M = 64), timed inside a hipGraph.  Run once per tile:
    LINA_SKINNY_TILE=4,2 python tools/perf_skinny3.py
hot = one weight buffer (L2 / MALL resident), cold = cycling through > 400 MB of weights."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lina_speech_amd import ops
from perf_skinny2 import timed_graph  # noqa
dev, M = "cuda", 64
def bench(N, K, nset, ln=False, sw=0, hot=False):
    ws = [torch.randn(2 * sw if sw else N, K, device=dev).bfloat16() for _ in range(1 if hot else nset)] * (nset if hot else 1)
    a = torch.randn(M, K, device=dev).bfloat16()
    nw = ws[0].shape[0]
    c1 = torch.randn(nw, device=dev) if ln else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    def fn():
        for w in ws: ops.linear_skinny(a, w, c1, c1, out=out, ln_dim=K if ln else 0, swiglu_hidden=sw, n_out=N)
    return timed_graph(fn, nset)
tile = os.environ.get("LINA_SKINNY_TILE", "auto")
res = []
for N, K in ((5120, 1024), (1024, 1024), (4112, 1024)):
    res.append(f"N={N} K={K}: hot {bench(N, K, 30, hot=True):.2f} cold {bench(N, K, min(60, int(400e6 // (N * K * 2)))):.2f}")
res.append(f"swiglu N=1376: cold+LN {bench(1376, 1024, 60, ln=True, sw=1365):.2f}")
print(f"tile {tile}: " + " | ".join(res))
