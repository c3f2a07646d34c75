"""Timing of the training-path kernels on a real MI355X: K2 forward + K2b backward at the L169 training shape, and
one full L169 train step (fwd + bwd + AdamW, bf16 autocast).  Prints JSON lines; torch events on the current stream
(the ops enqueue on it)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd import configs, ops  # noqa: E402
from lina_speech_amd.train import TrainStep, synthetic_batch  # noqa: E402


def timed(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=8)
    ap.add_argument("--T", type=int, default=4096)
    ap.add_argument("--no-step", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, H, T, D = args.b, 4, args.T, 256
    g = torch.Generator().manual_seed(0)
    mk = lambda: torch.randn(B, T, H, D, generator=g).to(torch.bfloat16).to(dev).transpose(1, 2)
    q, k, v, do = mk(), mk(), mk(), mk()
    gk = (torch.nn.functional.logsigmoid(torch.randn(B, T, H, D, generator=g)) / 16).to(torch.bfloat16).to(dev).transpose(1, 2)
    scale = D ** -0.5
    e = 2
    fwd_bytes = B * H * T * e * (3 * D + 2 * D)
    bwd_bytes = B * H * T * e * (5 * D + 4 * D)          # q,k,v,g,do in; dq,dk,dv,dg out
    t_f = timed(lambda: ops.chunk_gla(q, k, v, gk))
    t_b = timed(lambda: ops.gla_chunk_bwd(q, k, v, gk, do, scale))
    gk32 = gk.float()
    t_g = timed(lambda: ops.chunk_gla(q, k, v, gk32))
    print(json.dumps({"kernel": "K2 fwd, V-sliced generic kernel (fp32 gates)", "B": B, "H": H, "T": T, "ms": t_g * 1e3}))
    print(json.dumps({"kernel": "K2 fwd", "B": B, "H": H, "T": T, "ms": t_f * 1e3, "GB/s": fwd_bytes / t_f / 1e9}))
    print(json.dumps({"kernel": "K2b bwd (3 sweeps + dg)", "B": B, "H": H, "T": T, "ms": t_b * 1e3,
                      "GB/s": bwd_bytes / t_b / 1e9, "algorithmic_bytes": bwd_bytes}))
    if args.no_step:
        return
    torch.manual_seed(0)
    ts = TrainStep(configs.l169(), device=dev, ddp=False)
    batch = synthetic_batch(b=B, n=T + 1, t_txt=64, seed=1).to(dev)
    t_s = timed(lambda: ts.step(batch), iters=3, warmup=2)
    print(json.dumps({"what": "L169 train step (fwd+bwd+AdamW, bf16 autocast)", "b": B, "T": T, "ms": t_s * 1e3,
                      "tokens/s": B * T / t_s, "max_mem_GB": torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == "__main__":
    main()
