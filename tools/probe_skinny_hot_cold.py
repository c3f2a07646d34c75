"""Premise check for prefetching the decode step's projection weights into L2: a dependent chain of 40 o-projection-shaped
launches (M = 64, 1024 x 1024 weights, residual epilogue) inside ONE hipGraph with (hot) the SAME 2 MB weight every launch,
(cold) 40 different weights = 84 MB per pass with a 1 GB state-sized read between graph replays (nothing survives in L2 / the
Infinity Cache), us per launch.  Also the up-projection shape (2730 x 1024 + SwiGLU).
    python tools/probe_skinny_hot_cold.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
bf = torch.bfloat16
M, N_CHAIN = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 40
g = torch.Generator().manual_seed(0)
flush = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)     # 1 GB


def chain(n_out, k, weights, swiglu=0):
    x = torch.randn(M, k, generator=g).to(bf).to(dev)
    outs = [torch.empty(M, swiglu if swiglu else n_out, dtype=bf, device=dev) for _ in range(2)]
    res = torch.zeros(M, n_out, dtype=bf, device=dev)

    def body():
        a = x
        for i in range(N_CHAIN):
            w = weights[i % len(weights)]
            if swiglu:
                ops.linear_skinny(a, w, out=outs[i & 1], swiglu_hidden=swiglu)
            else:
                ops.linear_skinny(a, w, resid=res, out=outs[i & 1])
                a = outs[i & 1] if n_out == k else x
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            body()
    return gr


def timed(gr, cold, reps=30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(reps):
        if cold:
            flush.add_(1.0)
        a.record()
        gr.replay()
        b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / reps / N_CHAIN * 1e3


for name, n_out, k, sw in (("o / down shape 1024 x 1024 + residual", 1024, 1024, 0), ("up shape 2730 x 1024 + SwiGLU", 2730, 1024, 1365)):
    ws = [(torch.randn(n_out, k, generator=g) * 0.03).to(bf).to(dev) for _ in range(N_CHAIN)]
    hot = chain(n_out, k, ws[:1], sw)
    cold = chain(n_out, k, ws, sw)
    print(f"M = {M}: {name}: hot (same weight, L2-resident) {timed(hot, False):.2f} us per launch | cold (40 weights, caches flushed "
          f"between replays) {timed(cold, True):.2f} us | 40 weights, no flush {timed(cold, False):.2f} us")
