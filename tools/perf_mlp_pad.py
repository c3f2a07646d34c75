"""The channel mixer's six GEMMs at two paddings of its hidden width (H = 1365; Hp = 1408 today, 1536 = 6 x 256): does the GEMM
library prefer the 256-multiple enough to pay for 9 % more flops?   python tools/perf_mlp_pad.py"""
import torch

dev = torch.device("cuda", 0)
M, d = 32768, 1024
g = torch.Generator().manual_seed(0)
x = torch.randn(M, d, generator=g).to(torch.bfloat16).to(dev)
dyo = torch.randn(M, d, generator=g).to(torch.bfloat16).to(dev)


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
    for Hp in (1408, 1536, 1792, 2048):
        wi = (torch.randn(2 * Hp, d, generator=g) * 0.03).to(torch.bfloat16).to(dev)
        bi = torch.zeros(2 * Hp, dtype=torch.bfloat16, device=dev)
        wo = (torch.randn(d, Hp, generator=g) * 0.03).to(torch.bfloat16).to(dev)
        du = torch.randn(M, 2 * Hp, generator=g).to(torch.bfloat16).to(dev)
        h = torch.randn(M, Hp, generator=g).to(torch.bfloat16).to(dev)
        t = {}
        t["up fwd"] = timed(lambda: torch.addmm(bi, x, wi.t()))
        t["up dX"] = timed(lambda: torch.mm(du, wi))
        t["up dW s8"] = timed(lambda: torch.bmm(du.view(8, M // 8, 2 * Hp).transpose(1, 2), x.view(8, M // 8, d), out_dtype=torch.float32).sum(0))
        t["down fwd"] = timed(lambda: torch.mm(h, wo.t()))
        t["down dX"] = timed(lambda: torch.mm(dyo, wo))
        t["down dW s8"] = timed(lambda: torch.bmm(dyo.view(8, M // 8, d).transpose(1, 2), h.view(8, M // 8, Hp), out_dtype=torch.float32).sum(0))
        print(f"Hp = {Hp}: " + "  ".join(f"{k} {v:6.1f}" for k, v in t.items()) + f"   sum {sum(t.values()):7.1f} us")
        del wi, wo, du, h
