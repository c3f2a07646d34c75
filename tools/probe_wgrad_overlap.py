"""Do the train step's weight-gradient GEMMs (MFMA-bound, persistent stream-K kernels) overlap with its memory-bound
kernels when they sit on a second HIP stream?  Serial vs two-stream time of {dW GEMMs of one block} beside {conv bwd, LN,
norm-gate, K2b} at the L169 b = 8 x 4096 shapes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
M = 32768
bf = torch.bfloat16
g = torch.Generator(device="cpu").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g).to(bf).to(dev)
dy_in, x_in = rn(M, 4112), rn(M, 1024)
dy_up, dy_dn, h_dn = rn(M, 2816), rn(M, 1024), rn(M, 1408)


def wgrads():
    torch.mm(dy_in.t(), x_in, out_dtype=torch.float32)
    torch.bmm(dy_up.view(8, 4096, 2816).transpose(1, 2), x_in.view(8, 4096, 1024), out_dtype=torch.float32).sum(0)
    torch.bmm(dy_dn.view(8, 4096, 1024).transpose(1, 2), h_dn.view(8, 4096, 1408), out_dtype=torch.float32).sum(0)
    torch.bmm(dy_dn.view(8, 4096, 1024).transpose(1, 2), x_in.view(8, 4096, 1024), out_dtype=torch.float32).sum(0)


xa, xb = torch.randn(M, 1024, device=dev), torch.randn(M, 1024, device=dev)
q, k, v, do = (rn(8, 4096, 4, 256).transpose(1, 2) for _ in range(4))
gk = (torch.nn.functional.logsigmoid(torch.randn(8, 4096, 4, 256, generator=g)) / 16).to(bf).to(dev).transpose(1, 2)


def stream_kernels():                      # ~ the memory-bound stretch of one block's backward
    for _ in range(6):
        torch.add(xa, xb, out=xa)          # 0.4 GB each: LN / conv / gate-like passes


def k2b():
    ops.gla_chunk_bwd(q, k, v, gk, do, 256 ** -0.5)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


side = torch.cuda.Stream()


def both(main_fn):
    def f():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            wgrads()
        main_fn()
        cur.wait_stream(side)
    return f


tw = timed(wgrads)
for name, fn in (("6 x 0.4 GB streaming passes", stream_kernels), ("K2b (b = 8)", k2b)):
    tm = timed(fn)
    tb = timed(both(fn))
    print(f"wgrad GEMMs {tw:.3f} ms | {name} {tm:.3f} ms | serial {tw + tm:.3f} | two streams {tb:.3f} ms "
          f"(hidden {tw + tm - tb:.3f} ms = {(tw + tm - tb) / min(tw, tm) * 100:.0f} % of the shorter)")
