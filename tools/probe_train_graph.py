"""Whole-step capture of the L169 train step (TrainStep(graph=True)): does it capture, do captured and eager steps agree, what
does it buy.  Prints JSON lines."""
import json
import os
import sys
import time
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd import configs  # noqa: E402
from lina_speech_amd.train import TrainStep, synthetic_batch  # noqa: E402


def run(graph, steps=6, b=8, T=4096):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ts = TrainStep(configs.l169(), device=dev, ddp=False, graph=graph)
    batch = synthetic_batch(b=b, n=T + 1, t_txt=64, seed=1).to(dev)
    losses = [float(ts.step(batch)) for _ in range(3)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = ts.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    losses.append(float(loss))
    per_step = []
    for _ in range(4):                                       # step by step: loss and time of each
        t1 = time.perf_counter()
        l = float(ts.step(batch))
        per_step.append((round((time.perf_counter() - t1) * 1e3, 1), l))
    print(json.dumps({"graph": graph, "per_step_ms_loss": per_step}), flush=True)
    out = {"graph": graph, "captured": ts._graph is not None, "ms_per_step": dt * 1e3, "losses": losses,
           "lr": float(ts.opt.param_groups[0]["lr"]), "max_mem_GB": torch.cuda.max_memory_allocated() / 1e9}
    del ts
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    for g in ((True,) if os.environ.get("PROBE_GRAPH_ONLY") else (False, True)):
        try:
            print(json.dumps(run(g)), flush=True)
        except Exception:                                   # noqa: BLE001
            print(json.dumps({"graph": g, "error": traceback.format_exc()[-1500:]}), flush=True)
