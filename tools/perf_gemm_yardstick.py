"""Yardstick only (NOT a product path): the vendor GEMM (hipBLASLt / rocBLAS through torch.matmul, bf16, fp32 accumulate) at the
shapes of the decode step's projections, timed the way tools/perf_tall.py times this repo's kernels (four weight sets in
rotation so that successive launches do not find their weights in L2; HIP events on the current stream).  The vendor kernels
do NOT do the fused work of ours (LayerNorm fold, conv step, gate epilogues, SwiGLU, residual) -- the figure is a floor for the
bare contraction at that shape, nothing more.
    python tools/perf_gemm_yardstick.py [M] [reps]"""
import sys

import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev, dt = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(0)
shapes = [("inproj", 4112, 1024), ("up", 2730, 1024), ("o", 1024, 1024), ("down", 1024, 1376), ("head", 4099, 1024)]
for name, N, K in shapes:
    x = (torch.randn(M, K, generator=g) * 0.5).to(dt).to(dev)
    ws = [(torch.randn(N, K, generator=g) / 32).to(dt).to(dev) for _ in range(4)]
    out = torch.empty(M, N, dtype=dt, device=dev)
    for layout, fn in (("x @ W^T (W row-major [N,K])", lambda w: torch.matmul(x, w.t(), out=out)),
                       ("x @ Wt   (W stored [K,N])", None)):
        if fn is None:
            wts = [w.t().contiguous() for w in ws]
            run = lambda i: torch.matmul(x, wts[i % 4], out=out)
        else:
            run = lambda i, fn=fn: fn(ws[i % 4])
        for i in range(8):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"M={M} {name:7s} N={N} K={K} {layout:30s} {us:7.2f} us  {2 * M * N * K / us * 1e-6:7.1f} TFLOP/s")
