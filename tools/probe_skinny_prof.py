#!/usr/bin/env python
"""Inside the decode step's projection launches: time stamps of every workgroup (tools/skinny_prof.sh build,
LINA_GLA_LIB=tools/abl/liblina_skprof.so): dispatch skew, first load round, main loop, reduction, epilogue."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LINA_GLA_LIB", os.path.join(ROOT, "tools", "abl", "liblina_skprof.so"))
import numpy as np
import torch
from lina_speech_amd import ops, _lib
from lina_speech_amd.configs import l169
from lina_speech_amd.decode import DecodeEngine

dev = torch.device("cuda", 0)
B = 64
torch.manual_seed(0)
model = l169().eval().to(dev, torch.bfloat16)
texts = torch.randint(3, 256, (B, 64), generator=torch.Generator().manual_seed(1234)).to(dev)
with torch.inference_mode():
    eng = DecodeEngine(model, model.txt_encoder(model.txt_embed(texts)), batch_size=B)
    eng.begin_greedy(700)
    for _ in range(600):
        eng.greedy_step()
    torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)


def report(name, a):
    a = a[a[:, 1] > 0].astype(np.float64)
    if not len(a):
        print(name, ": no data"); return
    wall0 = a[:, 0].min()
    span_us = (a[:, 6].max() - wall0) / 100.0                       # 100 MHz wall clock
    clk_per_us = np.median((a[:, 5] - a[:, 1]) / np.maximum(a[:, 6] - a[:, 0], 1.0)) * 100.0
    rel = lambda col: (a[:, col] - a[:, 1]) / clk_per_us
    start = (a[:, 0] - wall0) / 100.0
    q = lambda x: f"{np.min(x):5.2f}/{np.median(x):5.2f}/{np.max(x):5.2f}"
    print(f"{name}: {len(a)} workgroups, first start -> last end {span_us:.2f} us, shader clock ~{clk_per_us:.0f} MHz")
    print(f"   start after the first workgroup (min/med/max us): {q(start)}")
    print(f"   entry -> first load round consumed: {q(rel(2))}   -> main loop done: {q(rel(3))}   -> reduction barrier: {q(rel(4))}"
          f"   -> end: {q(rel(5))}")
    print(f"   end (wall) after the first start: {q((a[:, 6] - wall0) / 100.0)}")
    if a[:, 7].max() > 0:
        print(f"   entry -> partial sums added (slot 7): {q(rel(7))}")


buf = np.zeros(1024 * 8, dtype=np.uint64)
rc = lib.lina_inproj_prof_read(buf.ctypes.data_as(ctypes.c_void_p))
report("in-projection (last launch)", buf.reshape(1024, 8))
a_ = buf.reshape(1024, 8)[:192].astype(np.float64)                  # L169: q | k | v | g (32 workgroups each) | gate (64)
if a_[:, 1].min() > 0:
    w0 = a_[:, 0].min()
    for nm, lo, cnt in (("q tiles", 0, 32), ("k tiles", 32, 32), ("v tiles", 64, 32), ("g tiles", 96, 32), ("gate tiles", 128, 64)):
        e_ = (a_[lo:lo + cnt, 6] - w0) / 100.0
        print(f"   {nm:10s}: end after the first start min/med/max {e_.min():5.2f}/{np.median(e_):5.2f}/{e_.max():5.2f} us")
buf = np.zeros(4 * 1024 * 8, dtype=np.uint64)
rc = lib.lina_skinny_prof_read(buf.ctypes.data_as(ctypes.c_void_p))
for kind, nm in enumerate(("LN-2 + up + SwiGLU", "o-projection (K=1024, residual)", "down (residual)", "other (head / cross)")):
    report(nm, buf.reshape(4, 1024, 8)[kind])
if hasattr(lib, "lina_cross_prof_read"):
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    lib.lina_cross_prof_read(buf.ctypes.data_as(ctypes.c_void_p))
    print("(cross_scores slots: 2 = query + LayerNorm parameters arrived, 3 = LayerNorm done, 4 = dot products done)")
    report("cross_scores", buf.reshape(1024, 8))
