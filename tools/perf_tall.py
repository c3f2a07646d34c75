"""The projection kernels of the decode step on their own at a large batch (L169 shapes, bf16, packed operands): the fused
input side of a mixer, LN-2 + up-projection + SwiGLU, the codec head.  A handful of launches each -- cheap enough for a
rocprofv3 --pmc pass (one counter set per pass; thousands of dispatches under --pmc take minutes).
    [LINA_TALL=0|1] [LINA_TALL_V=0|1] python tools/perf_tall.py [M] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev, dt = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(0)
mk = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dt).to(dev)
K, Kd, Vd, R, hid, hid_pad, L = 1024, 1024, 1024, 16, 1365, 1376, 4099
x_p = ops.pack_rows(mk(M, K))
layers = []
for _ in range(4):                      # four sets of weights (34 MB each set): successive launches do not find them in L2
    w_in = mk(2 * Kd + 2 * Vd + R, K) / 32
    up = torch.zeros(2, 1408, K, dtype=dt, device=dev)
    up[:, :hid] = mk(2, hid, K) / 32
    layers.append(dict(w_in=ops.pack_rows(w_in), c1=w_in.float().sum(1).contiguous(), c2=torch.randn(w_in.shape[0], generator=g).to(dev),
                       up=torch.cat([ops.pack_rows(up[0]), ops.pack_rows(up[1])]), uc1=torch.randn(2 * hid, generator=g).to(dev),
                       uc2=torch.randn(2 * hid, generator=g).to(dev), head=ops.pack_rows(mk(L, K) / 32)))
wq, wk, wv, w2, b2 = mk(Kd, 4), mk(Kd, 4), mk(Vd, 4), mk(Kd, R), mk(Kd)
cq, ck, cv = mk(M, Kd, 4), mk(M, Kd, 4), mk(M, Vd, 4)
qkv, go = torch.empty(M, 2 * Kd + Vd, dtype=dt, device=dev), torch.empty(M, Vd, dtype=dt, device=dev)
gk = torch.empty(M, Kd, dtype=torch.float32, device=dev)
s_p = torch.zeros(ops.packed_numel(M, hid_pad), dtype=dt, device=dev)
logits = torch.empty(M, L, dtype=dt, device=dev)


def inproj(P):
    ops.gla_decode_inproj_packed(x_p, P["w_in"], M, K, P["c1"], P["c2"], wq, wk, wv, cq, ck, cv, w2, b2, qkv, go, gk)


def up(P):
    ops.linear_skinny_packed(x_p, P["up"], M, hid_pad, K, P["uc1"], P["uc2"], out_packed=s_p, out_packed_width=hid_pad,
                             swiglu_hidden=hid, ln_dim=K, w_half_rows=1408)


def head(P):
    ops.linear_skinny_packed(x_p, P["head"], M, L, K, out=logits)


for name, fn in (("inproj", inproj), ("up", up), ("head", head)):
    for i in range(4):
        fn(layers[i % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(layers[i % 4])
    e1.record()
    torch.cuda.synchronize()
    print(f"M={M} {name:7s} {e0.elapsed_time(e1) / reps * 1e3:8.2f} us per launch  (LINA_TALL={os.environ.get('LINA_TALL', 'default')}, "
          f"V={os.environ.get('LINA_TALL_V', 'default')})")
