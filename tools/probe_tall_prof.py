#!/usr/bin/env python
"""Inside the fused in-projection at a large batch: per-workgroup time stamps (tools/skinny_prof.sh build, -DLINA_SKINNY_PROF) of
the 64-row kernel with gate workgroups (LINA_TALL_V=0) and of the 128-row kernel with the gate folded in (LINA_TALL_V=3).
    python tools/probe_tall_prof.py [M]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LINA_GLA_LIB", os.path.join(ROOT, "tools", "abl", "liblina_skprof.so"))
import numpy as np
import torch
from lina_speech_amd import ops, _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev, dt = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(0)
mk = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dt).to(dev)
K, Kd, Vd, R = 1024, 1024, 1024, 16
x_p = ops.pack_rows(mk(M, K))
layers = []
for _ in range(40):                     # 40 x 8.4 MB of weights: a launch does not find its weights in the Infinity Cache
    w_in = mk(2 * Kd + 2 * Vd + R, K) / 32
    layers.append(dict(w_in=ops.pack_rows(w_in), c1=w_in.float().sum(1).contiguous(), c2=torch.randn(w_in.shape[0], generator=g).to(dev)))
wq, wk, wv, w2, b2 = mk(Kd, 4), mk(Kd, 4), mk(Vd, 4), mk(Kd, R), mk(Kd)
caches = [(mk(M, Kd, 4), mk(M, Kd, 4), mk(M, Vd, 4)) for _ in range(40)]     # (and 40 sets of conv caches: HBM-cold, as in the step)
qkv, go = torch.empty(M, 2 * Kd + Vd, dtype=dt, device=dev), torch.empty(M, Vd, dtype=dt, device=dev)
gk = torch.empty(M, Kd, dtype=torch.float32, device=dev)
lib = _lib.load()                       # (binds to the HIP runtime of the process first)


def run(i):
    P, (cq, ck, cv) = layers[i % 40], caches[i % 40]
    ops.gla_decode_inproj_packed(x_p, P["w_in"], M, K, P["c1"], P["c2"], wq, wk, wv, cq, ck, cv, w2, b2, qkv, go, gk, w_stream=True)


for v in ("0", "3", "4"):
    os.environ["LINA_TALL_V"] = v
    for i in range(80):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    lib.lina_inproj_prof_read(buf.ctypes.data_as(ctypes.c_void_p))
    a = buf.reshape(1024, 8)
    if v != "0":
        a = a[:64 * ((M + 127) // 128) * (1 if v == "3" else 2)]         # (the buffer is not cleared between the two kernels: this one's workgroups only)
    a = a[a[:, 1] > 0].astype(np.float64)
    wall0 = a[:, 0].min()
    clk_per_us = np.median((a[:, 5] - a[:, 1]) / np.maximum(a[:, 6] - a[:, 0], 1.0)) * 100.0
    q = lambda x: f"{np.min(x):6.2f}/{np.median(x):6.2f}/{np.max(x):6.2f}"
    rel = lambda col: (a[:, col] - a[:, 1]) / clk_per_us
    print(f"LINA_TALL_V={v} M={M}: {us:.2f} us per launch (HBM-cold weights and conv caches); {len(a)} workgroups, shader clock ~{clk_per_us:.0f} MHz, "
          f"first start -> last end {(a[:, 6].max() - wall0) / 100.0:.2f} us")
    print(f"   start after the first workgroup (min/med/max us): {q((a[:, 0] - wall0) / 100.0)}")
    if v != "0":
        print(f"   entry -> prefetch issued {q(rel(7))} -> main loop done {q(rel(2))} -> tile staged {q(rel(3))} -> conv / g stored {q(rel(4))} -> end {q(rel(5))}")
    else:
        print(f"   entry -> main loop done {q(rel(2)[a[:, 2] > 0])} -> end {q(rel(5))}")
    print(f"   end (wall) after the first start: {q((a[:, 6] - wall0) / 100.0)}")
