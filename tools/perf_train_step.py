"""The L169 train step (fwd + CE + bwd + AdamW, bf16 autocast, b = 8 x 4096) on its own: wall time per step, HOST time to
issue a step (the Python call returns when everything is enqueued) and, under `rocprofv3 --kernel-trace --stats`, the kernel
table of exactly WARM + STEPS steps (tools/prof_summary.py -> profiles/r05_train_step_kernel_stats.csv: launches per step =
dispatches / (WARM + STEPS), GPU-busy time per step = total kernel time / (WARM + STEPS)).
    python tools/perf_train_step.py [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd import configs  # noqa: E402
from lina_speech_amd.train import TrainStep, synthetic_batch  # noqa: E402

if os.environ.get("TRAIN_OPERANDS") == "0":               # A/B: the torch-op construction of the stacked / padded weight operands
    from lina_speech_amd import ops as _ops
    _ops.POLICY.one_pass_operands = False
if os.environ.get("TRAIN_SPLIT_GEMM") == "0":             # A/B: the stacked projection's forward / dW as single GEMMs (N = 4160)
    from lina_speech_amd import ops as _ops2
    _ops2.POLICY.split_stacked_gemm = False
if os.environ.get("TRAIN_WIDE_DX") == "0":                # A/B: the down-projection's dX on the 1408-column operand
    from lina_speech_amd import ops as _ops3
    _ops3.POLICY.wide_down_dx = False
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
WARM = 2
dev = torch.device("cuda", 0)
torch.manual_seed(0)
ts = TrainStep(configs.l169(), device=dev, ddp=False)
if os.environ.get("TRAIN_TORCH_ADAMW") == "1":            # A/B: torch's own fused AdamW instead of K17
    ts.opt = torch.optim.AdamW(ts.model.parameters(), lr=5e-4, weight_decay=0.1, betas=(0.9, 0.999), fused=True)
    ts.sched = None
batch = synthetic_batch(b=8, n=4097, t_txt=64, seed=1).to(dev)
for _ in range(WARM):
    ts.step(batch)
torch.cuda.synchronize()
host, t0 = [], time.perf_counter()
for _ in range(STEPS):
    h0 = time.perf_counter()
    loss = ts.step(batch)
    host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / STEPS
print(json.dumps({"what": "L169 train step b=8 x 4096, bf16 autocast, eager launches", "steps": STEPS, "warmup_steps": WARM,
                  "ms_per_step": wall * 1e3, "host_issue_ms_per_step": sum(host) / STEPS * 1e3,
                  "host_issue_ms_min_max": [min(host) * 1e3, max(host) * 1e3], "loss": float(loss),
                  "max_mem_GB": torch.cuda.max_memory_allocated() / 1e9}))
