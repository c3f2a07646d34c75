#!/usr/bin/env python
"""How do the decode-step projection kernels (tiled for M <= 64 rows) compare with the GEMM library at M = 512 rows -- the
metric's batch on ONE GPU?  us per call, back to back (weights L2 / cache warm for both)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lina_speech_amd import ops

dev = torch.device("cuda", 0)
ops.get_backend().lib
g = torch.Generator().manual_seed(0)


def t_us(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for M in (64, 256, 512):
    for name, N, K in (("o-proj", 1024, 1024), ("in-proj", 4112, 1024), ("up (2 x 1365)", 2730, 1024), ("down", 1024, 1408)):
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        a_p, w_p = ops.pack_rows(a), ops.pack_rows(w)
        us_sk = t_us(lambda: ops.linear_skinny(a, w, out=out))
        us_pk = t_us(lambda: ops.linear_skinny_packed(a_p, w_p, M, N, K, out=out))
        us_mm = t_us(lambda: torch.mm(a, w.t(), out=out))
        print(f"M={M:4d} {name:14s} N={N:5d} K={K:5d}: skinny {us_sk:7.1f} us | packed {us_pk:7.1f} us | torch.mm {us_mm:7.1f} us", flush=True)
