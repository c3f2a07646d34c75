#!/usr/bin/env python
"""Inside ONE launch of the one-launch mixer (tools/iw_prof.sh build, LINA_GLA_LIB=tools/abl/liblina_iwprof.so): wall-clock
stamps of every workgroup of the last launch of a decode run -- when the K1w workgroups have issued their prefetch, see their
tiles, finish; when the in-projection workgroups finish their load rounds and hand their tile over."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LINA_GLA_LIB", os.path.join(ROOT, "tools", "abl", "liblina_iwprof.so"))
import numpy as np
import torch
from lina_speech_amd import ops, _lib
from lina_speech_amd.configs import l169
from lina_speech_amd.decode import DecodeEngine

dev = torch.device("cuda", 0)
B = 64
torch.manual_seed(0)
model = l169().eval().to(dev, torch.bfloat16)
texts = torch.randint(3, 256, (B, 64), generator=torch.Generator().manual_seed(1234)).to(dev)
torch.zeros(1, device=dev)
lib = ops.get_backend().lib                      # (the HIP runtime is in the process once torch has initialised the device)
q = lambda x: f"{np.min(x):6.2f} /{np.median(x):6.2f} /{np.max(x):6.2f}"
for n_pre, pace, steps, dl in ((24, 0, 603, 0), (24, 0, 603, 400), (24, 0, 603, 550), (24, 0, 603, 700)):
    os.environ['LINA_IW_DELAY'] = str(dl)
    with torch.inference_mode():
        eng = DecodeEngine(model, model.txt_encoder(model.txt_embed(texts)), batch_size=B, one_launch_mixer=True, n_pre=n_pre, pace=pace)
        eng.begin_greedy(700)
        eng.greedy_steps(steps)
        torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    lib.lina_iw_prof_read(buf.ctypes.data_as(ctypes.c_void_p))
    a = buf.reshape(1024, 8).astype(np.float64)
    kw, ip = a[:256], a[256:448]
    t0 = min(kw[:, 0].min(), ip[:, 0].min())
    us = lambda col: (col - t0) / 100.0
    print(f"--- n_pre={n_pre} pace={pace} delay={dl/100:.1f} us, last step at window position {(steps - 1) % 8} (7 = write-back)   [min / median / max us after the first start]")
    print(f"  K1w  entry               {q(us(kw[:, 0]))}")
    print(f"  K1w  prefetch issued     {q(us(kw[:, 1]))}")
    print(f"  K1w  tiles seen          {q(us(kw[:, 2]))}")
    print(f"  K1w  bookkeeping done    {q(us(kw[:, 3]))}")
    print(f"  K1w  late loads issued   {q(us(kw[:, 4]))}")
    print(f"  K1w  state pass done     {q(us(kw[:, 5]))}")
    print(f"  K1w  end                 {q(us(kw[:, 6]))}")
    for nm, lo, cnt in (("direct tiles", 0, 128), ("gate tiles", 128, 64)):
        x = ip[lo:lo + cnt]
        print(f"  in-projection {nm:12s} entry {q(us(x[:, 0]))} | first round {q(us(x[:, 2]))} | loop done {q(us(x[:, 3]))} | reduced {q(us(x[:, 4]))} | handed over {q(us(x[:, 5]))}")
