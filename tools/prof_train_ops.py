"""Which torch ops the kernels of one L169 train step come from: torch.profiler over 2 steps, self device time per aten op
and input shape (the rocprof kernel table names the kernels, this names their callers).  Writes gpurun_out/train_ops.txt."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd import configs  # noqa: E402
from lina_speech_amd.train import TrainStep, synthetic_batch  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ts = TrainStep(configs.l169(), device=dev, ddp=False)
    batch = synthetic_batch(b=8, n=4097, t_txt=64, seed=1).to(dev)
    for _ in range(2):
        ts.step(batch)
    torch.cuda.synchronize()
    steps = 2
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(steps):
            ts.step(batch)
        torch.cuda.synchronize()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/train_ops.txt", "w") as f:
        f.write(f"# {steps} steps; self device time per op and input shape\n")
        f.write(prof.key_averages(group_by_input_shape=True).table(sort_by="self_device_time_total", row_limit=120,
                                                                   max_name_column_width=70, max_shapes_column_width=90))
        f.write("\n\n# per op\n")
        f.write(prof.key_averages().table(sort_by="self_device_time_total", row_limit=60, max_name_column_width=70))
    print("wrote gpurun_out/train_ops.txt")


if __name__ == "__main__":
    main()
