"""Which torch operators the L169 train step's non-lina kernels come from: one step under torch.profiler, device time per
(aten op, input shapes), copies / casts / fills / reductions first.   python tools/prof_train_ops.py > gpurun_out/rNN_train_ops.txt"""
import os
import sys

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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    ts.step(batch)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = []
for e in ka:
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0:
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total self device time {tot / 1e3:.2f} ms over one step")
for dt, n, key, shp in rows[:120]:
    print(f"{dt / 1e3:8.3f} ms  x{n:4d}  {key[:60]:60s} {shp}")
