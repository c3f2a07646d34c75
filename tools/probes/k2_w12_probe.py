#!/usr/bin/env python
"""Timing of tools/probes/k2_w12_probe.hip (12-wave K2 with the DMA and phase A on four utility waves; results NOT checked
beyond finiteness) against the product K2, same inputs (B=64, H=4, T=4096), same process."""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from lina_speech_amd import ops

so = "/tmp/k2_w12_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-strict-aliasing",
                       "-Wno-inline-asm", "-I", os.path.join(ROOT, "lina-speech_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                       "-shared", "-fPIC", os.path.join(ROOT, "tools", "probes", "k2_w12_probe.hip"), "-o", so])
dev = "cuda"
B, H, T, D = 64, 4, 4096, 256
g = torch.Generator().manual_seed(0)
mk = lambda: torch.randn(B, T, H * D, generator=g).to(torch.bfloat16).to(dev).view(B, T, H, D).transpose(1, 2)
q, k, v = mk(), mk(), mk()
gk = (torch.nn.functional.logsigmoid(torch.randn(B, T, H * D, generator=g)) / 16).to(torch.bfloat16).to(dev).view(B, T, H, D).transpose(1, 2)
o = torch.empty(B, T, H, D, dtype=torch.bfloat16, device=dev).transpose(1, 2)
ops.get_backend().lib
lib = ctypes.CDLL(so)
lib.lina_k2_w12_probe.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
st = (ctypes.c_int64 * 15)(*[s for t in (q, k, v, gk, o) for s in (t.stride(0), t.stride(1), t.stride(2))])


def timed(fn, reps=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


stream = torch.cuda.current_stream().cuda_stream
nbytes = B * H * T * 2 * 5 * D
for rnd in range(2):
    ms = timed(lambda: ops.chunk_gla(q, k, v, gk, output_final_state=False))
    print(f"round {rnd}: product K2 (16 waves)            {ms:.4f} ms  {nbytes / ms / 1e9 / 8 * 100:.1f} % of 8 TB/s", flush=True)
    for ilp in (1, 2):
        fn = lambda: lib.lina_k2_w12_probe(q.data_ptr(), k.data_ptr(), v.data_ptr(), gk.data_ptr(), o.data_ptr(), B, H, T, st,
                                           D ** -0.5, ilp, stream)
        ms = timed(fn)
        fin = bool(torch.isfinite(o.float()).all())
        print(f"round {rnd}: 12-wave probe, phase A ilp={ilp}      {ms:.4f} ms  {nbytes / ms / 1e9 / 8 * 100:.1f} % of 8 TB/s  (finite output: {fin})", flush=True)
