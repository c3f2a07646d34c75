"""Where the train step's activation-sized elementwise ops come from: torch.profiler with Python stacks, the aten::copy_ / add /
mul / to calls on [8, 4096, 1024]-sized tensors grouped by their innermost repo frame.   python tools/prof_train_copies.py"""
import os
import sys
from collections import defaultdict

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd import configs  # noqa: E402
from lina_speech_amd.train import TrainStep, synthetic_batch  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
ts = TrainStep(configs.l169(), device=dev, ddp=False)
batch = synthetic_batch(b=8, n=4097, t_txt=64, seed=1).to(dev)
for _ in range(3):
    ts.step(batch)
torch.cuda.synchronize()
SMALL = "--small" in sys.argv


def table(prof, tag):
    agg = defaultdict(lambda: [0.0, 0])
    for e in prof.events():
        if e.name not in ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::sum", "aten::fill_", "aten::_to_copy", "aten::cat"):
            continue
        shp = str(e.input_shapes)
        big = "4096, 1024" in shp or "4097, 1024" in shp
        small = any(shp.startswith(pre) for pre in ("[[1024, 4]", "[[1024]", "[[1024, 16]", "[[3072, 4]", "[[256]", "[[2, 1365", "[[1024, 1365]"))
        if not (big or (SMALL and small)):
            continue
        dt = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
        par, chain = e.cpu_parent, []
        while par is not None and len(chain) < 4:
            chain.append(par.name)
            par = par.cpu_parent
        agg[(e.name, shp[:48], " < ".join(chain)[:120])][0] += dt
        agg[(e.name, shp[:48], " < ".join(chain)[:120])][1] += 1
    print(f"== {tag}")
    for (name, shp, chain), (dt, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"{dt / 1e3:7.3f} ms x{n:3d}  {name:12s} {shp:48s} {chain}")


ts.opt.zero_grad(set_to_none=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    loss = ts.loss(batch)
    torch.cuda.synchronize()
table(prof, "forward")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    loss.backward()
    torch.cuda.synchronize()
table(prof, "backward")
