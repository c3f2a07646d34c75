"""Token-split factor of the train path's weight-gradient GEMMs (dW = dY^T X over 32768 tokens, fp32 partial products + sum) at the
shapes of the final tree.   python tools/perf_dw_split.py"""
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


for name, n_out, n_in in (("in-proj main", 4096, 1024), ("up", 2816, 1024), ("down", 1024, 1408), ("o-proj", 1024, 1024), ("tail", 64, 1024)):
    dy = torch.randn(M, n_out, generator=g).to(torch.bfloat16).to(dev)
    x = torch.randn(M, n_in, generator=g).to(torch.bfloat16).to(dev)
    out = [f"one GEMM {timed(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32)):6.1f}"]
    for S in (2, 4, 8, 16, 32):
        f = lambda S=S: torch.bmm(dy.view(S, M // S, n_out).transpose(1, 2), x.view(S, M // S, n_in), out_dtype=torch.float32).sum(0)
        ft = lambda S=S: torch.bmm(x.view(S, M // S, n_in).transpose(1, 2), dy.view(S, M // S, n_out), out_dtype=torch.float32).sum(0)
        out.append(f"S={S}: {timed(f):6.1f} / transposed {timed(ft):6.1f}")
    print(f"{name:13s} [{n_out}, {n_in}]  " + "   ".join(out))
    del dy, x
