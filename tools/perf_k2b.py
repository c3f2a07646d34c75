#!/usr/bin/env python
"""Micro-benchmark of K2b (chunk backward: three sweeps + dg scan) at the training shape, settled clocks."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lina_speech_amd import ops

B, H, T, Dk, Dv = int(os.environ.get("K2_B", 8)), 4, int(os.environ.get("K2_T", 4096)), 256, 256
reps = int(os.environ.get("K2_REPS", 300))
dev = "cuda"
g = torch.Generator().manual_seed(0)
mk = lambda D: torch.randn(B, T, H * D, generator=g).to(torch.bfloat16).to(dev).view(B, T, H, D).transpose(1, 2)
q, k, v, do = mk(Dk), mk(Dk), mk(Dv), mk(Dv)
gk = (torch.nn.functional.logsigmoid(torch.randn(B, T, H * Dk, generator=g)) / 16).to(torch.bfloat16).to(dev)
gk = gk.view(B, T, H, Dk).transpose(1, 2)
scale = Dk ** -0.5
PATH = os.environ.get("K2B_PATH", "full")                  # full | sweeps
NSEG = os.environ.get("K2B_NSEG") or None                        # segments of the full-head sweeps (default: ops.chunk_segments)
nseg = ops.chunk_segments(B * H, T) if NSEG is None else int(NSEG)
kept = []
if PATH == "full" and nseg > 1 and os.environ.get("K2B_STATES", "1") != "0":   # as in training: the forward's boundary states
    ops._gla_launch("lina_gla_chunk_fwd", q, k, v, gk, scale, None, False, nseg=nseg, keep_seg_states=kept)
seg_ws = kept[0][0] if kept else None
fn = lambda: ops.gla_chunk_bwd(q, k, v, gk, do, scale, nseg=nseg, path=PATH, seg_states=seg_ws)
fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record()
torch.cuda.synchronize()
dt = e0.elapsed_time(e1) * 1e-3 / reps
nbytes = B * H * T * 2 * (5 * Dk + 4 * Dv)
print(f"K2b[{PATH},nseg={nseg},fwd_states={seg_ws is not None}] B={B} T={T}: {dt*1e3:.3f} ms  {nbytes/dt/1e9:.1f} GB/s ({nbytes/dt/8e12*100:.1f}% of 8 TB/s)")
