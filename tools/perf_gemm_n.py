"""Rate of the library GEMM [32768, K] x [K, N] (bf16) over N and K: which widths get the fast kernels.   python tools/perf_gemm_n.py"""
import torch

dev = torch.device("cuda", 0)
M = 32768
g = torch.Generator().manual_seed(0)


def timed(fn, iters=40, warm=15):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for K in (1024, 1408, 2816, 4096):
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    line = []
    for N in (512, 768, 1024, 1280, 1408, 1536, 2048, 2816, 3072, 4096):
        w = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).to(dev)
        us = timed(lambda: torch.mm(x, w.t()))
        line.append(f"N={N}: {us:6.1f} us {2.0 * M * N * K / us / 1e9:5.2f} PF/s")
        del w
    print(f"K = {K}\n   " + "\n   ".join(line))
    del x
