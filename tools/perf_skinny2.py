#!/usr/bin/env python
"""linear_skinny: fixed overhead vs HBM streaming, timed inside a hipGraph (no host launch cost).
hot = one weight buffer reused (L2/MALL resident), cold = cycling through > 256 MiB of weights."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lina_speech_amd import ops
dev = "cuda"; M = 64
def timed_graph(fn, n):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)
def bench(N, K, nset, ln=False, sw=0):
    ws = [torch.randn(2 * sw if sw else N, K, device=dev).bfloat16() for _ in range(nset)]
    a = torch.randn(M, K, device=dev).bfloat16()
    nw = ws[0].shape[0]
    c1 = torch.randn(nw, device=dev) if ln else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    def fn():
        for w in ws: ops.linear_skinny(a, w, c1, c1, out=out, ln_dim=K if ln else 0, swiglu_hidden=sw, n_out=N)
    return timed_graph(fn, nset)
def main():
    x = torch.zeros(64, device=dev)
    print(f"trivial torch add_ in graph: {timed_graph(lambda: [x.add_(1) for _ in range(50)], 50):.2f} us")
    for N, K in ((1024, 1024), (4112, 1024), (1024, 32), (4112, 32), (1024, 256), (4112, 256), (1024, 1376)):
        nset_cold = min(max(2, int(400e6 // (N * K * 2))), 60)
        print(f"N={N} K={K}: hot {bench(N, K, 40 if N*K*2 < 4e6 else 20) if False else bench(N, K, 1):.2f} us   cold {bench(N, K, nset_cold):.2f} us   "
              f"cold+LN {bench(N, K, nset_cold, ln=True):.2f} us")
    print(f"up-proj swiglu N=1376 K=1024 cold+LN: {bench(1376, 1024, 60, ln=True, sw=1365):.2f} us")

if __name__ == "__main__":
    main()
