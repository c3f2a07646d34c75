"""Do the SwiGLU GEMMs of the L169 train step (hidden 1365 = 1024*4//3: rows of 2730 / 1365 elements, not 16-byte aligned) run
faster with the hidden dimension padded to a multiple of 8 / 16 / 64?  One JSON line per (GEMM, hidden)."""
import json
import sys

import torch


def timed(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev = torch.device("cuda", 0)
    M, d = 32768, 1024
    bf = torch.bfloat16
    x = torch.randn(M, d, device=dev, dtype=bf)
    dyd = torch.randn(M, d, device=dev, dtype=bf)
    for H in (1365, 1368, 1376, 1408):
        w_in = torch.randn(2 * H, d, device=dev, dtype=bf)
        b_in = torch.randn(2 * H, device=dev, dtype=bf)
        w_out = torch.randn(d, H, device=dev, dtype=bf)
        b_out = torch.randn(d, device=dev, dtype=bf)
        u = torch.randn(M, 2 * H, device=dev, dtype=bf)
        h = torch.randn(M, H, device=dev, dtype=bf)
        S = 8
        cases = {
            "up fwd (addmm)": (lambda: torch.nn.functional.linear(x, w_in, b_in), 2.0 * M * d * 2 * H),
            "up fwd (mm, no bias)": (lambda: torch.mm(x, w_in.t()), 2.0 * M * d * 2 * H),
            "up dX": (lambda: torch.mm(u, w_in), 2.0 * M * d * 2 * H),
            "up dW split 8": (lambda: torch.bmm(u.view(S, M // S, 2 * H).transpose(1, 2), x.view(S, M // S, d),
                                                out_dtype=torch.float32).sum(0), 2.0 * M * d * 2 * H),
            "down fwd (addmm)": (lambda: torch.nn.functional.linear(h, w_out, b_out), 2.0 * M * d * H),
            "down fwd (mm, no bias)": (lambda: torch.mm(h, w_out.t()), 2.0 * M * d * H),
            "down dX": (lambda: torch.mm(dyd, w_out), 2.0 * M * d * H),
            "down dW split 8": (lambda: torch.bmm(dyd.view(S, M // S, d).transpose(1, 2), h.view(S, M // S, H),
                                                  out_dtype=torch.float32).sum(0), 2.0 * M * d * H),
            "down dW split 8 transposed": (lambda: torch.bmm(h.view(S, M // S, H).transpose(1, 2), dyd.view(S, M // S, d),
                                                             out_dtype=torch.float32).sum(0).t(), 2.0 * M * d * H),
        }
        for name, (fn, fl) in cases.items():
            us = timed(fn)
            print(json.dumps({"gemm": name, "hidden": H, "us": round(us, 1), "TFLOP/s": round(fl / us / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    sys.exit(main())
