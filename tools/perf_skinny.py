#!/usr/bin/env python
"""Micro-benchmark of lina_linear_skinny at the decode shapes; weights cycle through > 256 MiB so they stream
from HBM as in the real step.  Used under rocprofv3 for PMC passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lina_speech_amd import ops

dev = "cuda"
M = 64
shapes = {"inproj": (4112, 1024, True), "oproj": (1024, 1024, False), "down": (1024, 1376, False), "head": (4099, 1024, False)}
which = os.environ.get("SK", "inproj,oproj,down,head").split(",")
for name in which:
    N, K, ln = shapes[name]
    nset = max(2, int(400e6 // (N * K * 2)))
    ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(nset)]
    a = torch.randn(M, K, device=dev).bfloat16()
    c1 = torch.randn(N, device=dev) if ln else None
    c2 = torch.randn(N, device=dev) if ln else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for w in ws[:2]:
        ops.linear_skinny(a, w, c1, c2, out=out, ln_dim=K if ln else 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        for w in ws:
            ops.linear_skinny(a, w, c1, c2, out=out, ln_dim=K if ln else 0)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * nset)
    print(f"{name}: N={N} K={K} ln={ln}: {us:.2f} us/launch (back-to-back)  W stream {N*K*2/us/1e6:.0f} GB/s")
