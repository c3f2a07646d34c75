"""The big GEMMs of the L169 train step (M = 32768 tokens): does the operand layout (weight stored [N, K] vs [K, N]) or the
BLAS backend change what the library delivers?  One JSON line per (backend, GEMM, layout)."""
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
    M = 32768
    bf = torch.bfloat16
    shapes = [("in_proj", 4112, 1024), ("in_proj 4096", 4096, 1024), ("o_proj", 1024, 1024), ("up", 2816, 1024), ("down", 1024, 1408),
              ("head", 4099, 1024), ("head 4224", 4224, 1024)]
    backends = ["default"]
    for name in ("hipblaslt", "rocblas", "ck"):
        backends.append(name)
    for be in backends:
        if be != "default":
            try:
                torch.backends.cuda.preferred_blas_library(be)
            except Exception as e:                              # noqa: BLE001
                print(json.dumps({"backend": be, "error": str(e)[:100]}))
                continue
        for name, N, K in shapes:
            x = torch.randn(M, K, device=dev, dtype=bf)
            dy = torch.randn(M, N, device=dev, dtype=bf)
            w = torch.randn(N, K, device=dev, dtype=bf)          # nn.Linear layout
            wt = w.t().contiguous()                              # [K, N]
            fl = 2.0 * M * N * K
            cases = {"fwd  x @ W[N,K].t()": lambda: torch.mm(x, w.t()),
                     "fwd  x @ Wt[K,N]": lambda: torch.mm(x, wt),
                     "dX   dy @ W[N,K]": lambda: torch.mm(dy, w),
                     "dX   dy @ Wt[K,N].t()": lambda: torch.mm(dy, wt.t()),
                     "dW   dy.t() @ x": lambda: torch.mm(dy.t(), x),
                     "dWt  x.t() @ dy": lambda: torch.mm(x.t(), dy)}
            for cn, fn in cases.items():
                try:
                    us = timed(fn)
                    print(json.dumps({"backend": be, "gemm": name, "case": cn, "us": round(us, 1),
                                      "TFLOP/s": round(fl / us / 1e6, 1)}), flush=True)
                except Exception as e:                          # noqa: BLE001
                    print(json.dumps({"backend": be, "gemm": name, "case": cn, "error": str(e)[:100]}), flush=True)


if __name__ == "__main__":
    sys.exit(main())
