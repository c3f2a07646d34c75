"""K5 / K5b (norm-gate forward / backward) and K10 / K10b (LayerNorm + residual) on their own at the L169 train shapes
(b = 8 x 4096): us per launch and TB/s against their algorithmic bytes.  LINA_GLA_LIB picks an A/B build of the library.
    python tools/perf_norm.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd import ops  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda", 0)
bf = torch.bfloat16
B, T, H, D = 8, 4096, 4, 256
g = torch.Generator().manual_seed(0)
z = torch.randn(B, T, 4112, generator=g).to(bf).to(dev)                      # the stacked projection: g = columns 3072..4095
o = torch.randn(B, T, H, D, generator=g).to(bf).to(dev).requires_grad_()
gate = z[..., 3072:4096].view(B, T, H, D).requires_grad_()
w = torch.ones(D, device=dev, requires_grad=True)
dy = torch.randn(B, T, H, D, generator=g).to(bf).to(dev)


def timed(fn, reps=REPS):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


n = B * T * H * D
with torch.no_grad():
    t_f = timed(lambda: ops.rmsnorm_swish_gate(o, gate, w, 1e-5))
print(f"K5  norm-gate fwd  {t_f:7.1f} us  {3 * n * 2 / t_f / 1e6:5.2f} TB/s (x, g in; y out)")
y = ops.rmsnorm_swish_gate(o, gate, w, 1e-5)
t_fb = timed(lambda: torch.autograd.grad(y, (o, gate, w), dy, retain_graph=True))
print(f"K5b norm-gate bwd  {t_fb:7.1f} us incl. the partial sum  {5 * n * 2 / t_fb / 1e6:5.2f} TB/s (x, g, dy in; dx, dg out)")
x32 = torch.randn(B * T, 1024, generator=g).to(dev).requires_grad_()
r = torch.randn(B * T, 1024, generator=g).to(bf).to(dev).requires_grad_()
gam, bet = torch.ones(1024, device=dev, requires_grad=True), torch.zeros(1024, device=dev, requires_grad=True)
with torch.no_grad():
    t_l = timed(lambda: ops.layer_norm(x32, gam, bet, 1e-5, residual=r, out_dtype=bf))
m = B * T * 1024
print(f"K10 LayerNorm + residual fwd {t_l:7.1f} us  {m * (4 + 2 + 4 + 2) / t_l / 1e6:5.2f} TB/s")
yl, xs = ops.layer_norm(x32, gam, bet, 1e-5, residual=r, out_dtype=bf)
dyl, dxs = torch.randn_like(yl), torch.randn_like(xs)
t_lb = timed(lambda: torch.autograd.grad((yl, xs), (x32, r, gam, bet), (dyl, dxs), retain_graph=True))
print(f"K10b LayerNorm bwd {t_lb:7.1f} us incl. the partial sums  {m * (2 + 4 + 4 + 4 + 2) / t_lb / 1e6:5.2f} TB/s")
