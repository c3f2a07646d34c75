"""Weight-gradient GEMMs of the L169 train step (dW = dY^T X, reduction over 32768 tokens, small outputs): the form autograd
issues against re-shaped forms of the same product -- transposed problem, split over the token axis as a batched GEMM + sum,
fp32 output.  Prints one JSON line per (shape, variant)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
    T = 32768
    g = torch.Generator().manual_seed(0)
    for name, n_out, n_in in [("o_proj", 1024, 1024), ("down", 1024, 1365), ("up", 2730, 1024), ("in_proj", 4112, 1024),
                              ("gk1", 1024, 16)]:
        dy = torch.randn(T, n_out, generator=g).to(torch.bfloat16).to(dev)
        x = torch.randn(T, n_in, generator=g).to(torch.bfloat16).to(dev)
        ref = (dy.float().t() @ x.float())
        flops = 2.0 * T * n_out * n_in
        variants = {"autograd: dy.t() @ x": lambda: dy.t().mm(x),
                    "transposed: (x.t() @ dy).t()": lambda: x.t().mm(dy).t()}
        for S in (4, 8, 16, 32):
            def splitk(S=S):
                return torch.bmm(dy.view(S, T // S, n_out).transpose(1, 2), x.view(S, T // S, n_in)).sum(0)
            variants[f"split S={S}: bmm + sum"] = splitk
            def splitk_t(S=S):
                return torch.bmm(x.view(S, T // S, n_in).transpose(1, 2), dy.view(S, T // S, n_out)).sum(0).t()
            variants[f"split S={S} transposed"] = splitk_t
        if n_in % 8:
            xp = torch.nn.functional.pad(x, (0, 8 - n_in % 8))
            variants["padded x (in multiple of 8)"] = lambda: dy.t().mm(xp)[:, :n_in]
            variants["padded x, pad included"] = lambda: dy.t().mm(torch.nn.functional.pad(x, (0, 8 - n_in % 8)))[:, :n_in]
            for S in (4, 8):
                def splitk_p(S=S):
                    return torch.bmm(dy.view(S, T // S, n_out).transpose(1, 2), xp.view(S, T // S, -1)).sum(0)[:, :n_in]
                variants[f"padded split S={S}"] = splitk_p
        try:
            torch.mm(dy.t(), x, out_dtype=torch.float32)
            variants["fp32 out: mm(out_dtype=f32)"] = lambda: torch.mm(dy.t(), x, out_dtype=torch.float32)
        except Exception as e:                                  # noqa: BLE001
            print(json.dumps({"shape": name, "variant": "fp32 out", "error": str(e)[:120]}))
        for vn, fn in variants.items():
            try:
                out = fn()
                err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
                us = timed(fn)
                print(json.dumps({"shape": name, "out": [n_out, n_in], "variant": vn, "us": round(us, 1),
                                  "TFLOP/s": round(flops / us / 1e6, 1), "rel_err": round(err, 5)}), flush=True)
            except Exception as e:                              # noqa: BLE001
                print(json.dumps({"shape": name, "variant": vn, "error": str(e)[:160]}), flush=True)


if __name__ == "__main__":
    main()
