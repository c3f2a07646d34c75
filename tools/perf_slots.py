#!/usr/bin/env python
"""What does one launch of the decode step's small kernels cost inside a hipGraph chain of 64 dependent launches of the
SAME kernel (no heavy neighbours)?  Compared with their slots inside the real step (rocprofv3) this separates the
kernel's own latency chain from what its neighbours leave behind at the boundary."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lina_speech_amd import ops

dev = torch.device("cuda", 0)
bf = torch.bfloat16
B, d, Tn = 64, 1024, 64

def timed_chain(fn, n=64, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)

x = torch.randn(B, d, device=dev).to(bf)
one = torch.zeros(1, dtype=torch.long, device=dev)
print(f"torch add_ (1 element)            : {timed_chain(lambda: one.add_(1)):6.2f} us")
sc = torch.randn(B, Tn, device=dev)
att = torch.zeros(B, 2, 1, Tn, device=dev, dtype=bf)
attc = torch.zeros(B, 64, device=dev, dtype=bf)
print(f"softmax_rows                      : {timed_chain(lambda: ops.softmax_rows(sc, 1.0, att[:, 0, 0], attc, Tn)):6.2f} us")
w = (torch.randn(d, d, device=dev) / 32).to(bf)
x_p, w_p = ops.pack_rows(x), ops.pack_rows(w)
og_p = ops.pack_rows(torch.randn(B, d, device=dev).to(bf))
print(f"linear_skinny<1,1> packed, in-place residual (o_proj shape): "
      f"{timed_chain(lambda: ops.linear_skinny_packed(og_p, w_p, B, d, d, resid=x_p, out_packed=x_p, out_packed_width=d)):6.2f} us")
out = torch.empty(B, d, device=dev, dtype=bf)
print(f"linear_skinny<1,1> row-major                                : {timed_chain(lambda: ops.linear_skinny(x, w, out=out)):6.2f} us")
ws = [ops.pack_rows((torch.randn(d, d, device=dev) / 32).to(bf)) for _ in range(64)]
it = iter(range(10 ** 9))
def cold():
    ops.linear_skinny_packed(og_p, ws[next(it) % 64], B, d, d, resid=x_p, out_packed=x_p, out_packed_width=d)
print(f"linear_skinny<1,1> packed, 64 different weights (128 MB)    : {timed_chain(cold):6.2f} us")
vv = torch.randn(B, Tn, d, device=dev).to(bf)
sc2 = torch.randn(B, 64, device=dev).to(bf)
print(f"softmax_weighted_rows_add         : {timed_chain(lambda: ops.softmax_weighted_rows_add(sc2, 0.03, att[:, 1, 0], vv, x)):6.2f} us")
