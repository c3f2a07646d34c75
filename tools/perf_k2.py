#!/usr/bin/env python
"""Micro-benchmark of K2 (chunk kernel) at the training shape; used under rocprofv3 for PMC passes."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lina_speech_amd import ops

H = int(os.environ.get("K2_H", 4))                        # heads of the 1024-wide model: head dimension 1024 / H
B, T, Dk = int(os.environ.get("K2_B", 64)), int(os.environ.get("K2_T", 4096)), 1024 // H
Dv = int(os.environ.get("K2_DV", Dk))                     # K2_DV=512: expand_v = 2 (two 256-column launches per call)
reps = int(os.environ.get("K2_REPS", 5))
if os.environ.get("K2_DV512_ONE") is not None:            # 256 x 512 heads: "1" one launch of two workgroups per head (default), "0" two launches
    ops.POLICY.dv512_one_launch = os.environ["K2_DV512_ONE"] != "0"
dev = "cuda"
g = torch.Generator().manual_seed(0)
mk = lambda D: torch.randn(B, T, H * D, generator=g).to(torch.bfloat16).to(dev).view(B, T, H, D).transpose(1, 2)
q, k, v = mk(Dk), mk(Dk), mk(Dv)
gk = (torch.nn.functional.logsigmoid(torch.randn(B, T, H * Dk, generator=g)) / 16).to(torch.bfloat16).to(dev)
gk = gk.view(B, T, H, Dk).transpose(1, 2)
HT = os.environ.get("K2_HT", "1") != "0"                   # also return the final state (67 MB more at B = 64)
NSEG = os.environ.get("K2_NSEG") or None
run = lambda: ops.chunk_gla(q, k, v, gk, output_final_state=HT, nseg=None if NSEG is None else int(NSEG))
BWD = os.environ.get("K2_BWD", "0") != "0"                 # K2b instead: the three sweeps of lina_gla_chunk_bwd_full (no segments at B*H >= 256)
if BWD:
    do = mk(Dv)
    run = lambda: ops.gla_chunk_bwd(q, k, v, gk, do, Dk ** -0.5)
run()
torch.cuda.synchronize()
if reps >= 100:                                            # settle the clocks first
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        for _ in range(10):
            run()
        torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
dt = e0.elapsed_time(e1) * 1e-3 / reps
nbytes = B * H * T * 2 * ((5 * Dk + 4 * Dv) if BWD else (3 * Dk + 2 * Dv))
print(f"{'K2b' if BWD else 'K2'}[final_state={HT}] B={B} T={T}: {dt*1e3:.3f} ms  {nbytes/dt/1e9:.1f} GB/s ({nbytes/dt/8e12*100:.1f}% of 8 TB/s)  "
      f"{dt/(T/32)*2.4e9:.0f} clk/chunk @2.4GHz")
if os.environ.get("K2_PROF"):
    import ctypes, numpy as np
    from lina_speech_amd import _lib
    lib = _lib.load()
    ops.chunk_gla(q, k, v, gk, output_final_state=True)
    torch.cuda.synchronize()
    buf = np.zeros(256 + 3 * 1024, dtype=np.uint64)
    rc = lib.lina_k2_prof_read(buf.ctypes.data_as(ctypes.c_void_p))
    chunks = T / 32 / max(1, ops.chunk_segments(B * H, T) if NSEG is None else int(NSEG))   # chunks ONE workgroup walks (segments split T)
    a = buf[:256].reshape(16, 16).astype(np.float64) / chunks
    names = ["phaseA", "bar(2)", "flags/roll", "maskA(w<4)", "step1 qS", "step4 upd", "bar(1')+dma", "rawrd+step3", "wait_vmem",
             "bar(3)", "o stores", "(unused)"]   # o stores: of the previous chunk, at the end of phase A
    print("clk/chunk per phase (shader clock), waves 0, 3, 4, 15 and mean:  rc =", rc)
    for i, nm in enumerate(names):
        print(f"  {nm:12s} " + " ".join(f"{a[w, i]:8.0f}" for w in (0, 3, 4, 15)) + f"   mean {a[:, i].mean():8.0f}")
    print(f"  total        {a[0, :12].sum():8.0f}")
    wg = buf[256:256 + 3 * B * H].reshape(-1, 3).astype(np.float64) / chunks
    print("per-workgroup clk/chunk (wave 0): total min/median/max", np.min(wg[:, 0]), np.median(wg[:, 0]), np.max(wg[:, 0]),
          " wait_vmem median/max", np.median(wg[:, 1]), np.max(wg[:, 1]), " bar(3) median/max", np.median(wg[:, 2]), np.max(wg[:, 2]))
    order = np.argsort(wg[:, 0])
    print("slowest workgroups:", order[-8:], wg[order[-8:], 0].round(), " fastest:", order[:8], wg[order[:8], 0].round())
