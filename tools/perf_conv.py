"""K3 / K3b (the fused q | k | v short convolution, D = 3072 column slices of the stacked projection) on their own at the L169
train shape b = 8 x 4096, cycling 4 operand sets (2.4 GB: nothing survives in the Infinity Cache); run under
`rocprofv3 --kernel-trace` + tools/prof_pick.py for kernel times.   python tools/perf_conv.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd import ops, _lib  # noqa: E402

_lib.CONV_BWD_TT = int(os.environ.get("LINA_CONV_BWD_TT", _lib.CONV_BWD_TT))     # (A/B builds of the library: -DLINA_CONV_BWD_TT)

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
bf = torch.bfloat16
B, T, Kd = 8, 4096, 1024
ZW = int(os.environ.get("CONV_ZW", 4112))        # row width of the stacked projection (4112 = 8224-byte rows: not a multiple of 128 B)
g = torch.Generator().manual_seed(0)
sets = []
for _ in range(4):
    z = torch.randn(B, T, ZW, generator=g).to(bf).to(dev).requires_grad_()
    ws = [torch.randn(Kd, 1, 4, generator=g).to(dev).requires_grad_() for _ in range(3)]
    dys = torch.randn(B, T, 3 * Kd, generator=g).to(bf).to(dev)
    sets.append((z, ws, dys))
for r in range(REPS):
    z, ws, dys = sets[r % 4]
    (q, k, v, g_, lr), slab = ops.split_slab(z, [Kd, Kd, Kd, Kd, ZW - 4 * Kd])
    out = ops.short_conv3((q, k, v), ws, [None] * 3, None, "silu", grad_slab=(slab, 0))
    torch.autograd.grad(out, [z] + ws, list(dys.split(Kd, dim=-1)), allow_unused=True)
torch.cuda.synchronize()
print("done")
