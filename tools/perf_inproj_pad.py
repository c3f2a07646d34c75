"""The stacked in-projection's three GEMMs at different paddings of its output width (4112 real columns: q | k | v | g | 16):
what the GEMM library makes of N = 4112 / 4160 / 4224 / 4352 / 4608.   python tools/perf_inproj_pad.py"""
import torch

dev = torch.device("cuda", 0)
M, K = 32768, 1024
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)


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


for _ in range(2):
    for N in (4096, 4112, 4160, 4224, 4352, 4608):
        w = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).to(dev)
        dy = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
        f = timed(lambda: torch.mm(x, w.t()))
        dx = timed(lambda: torch.mm(dy, w))
        dw = timed(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32))
        dw4 = timed(lambda: torch.bmm(dy.view(4, M // 4, N).transpose(1, 2), x.view(4, M // 4, K), out_dtype=torch.float32).sum(0))
        print(f"N = {N}: forward {f:6.1f} us   dX {dx:6.1f}   dW one GEMM {dw:6.1f}   dW token-split 4 + sum {dw4:6.1f}   sum {f + dx + min(dw, dw4):7.1f}")
        del w, dy
