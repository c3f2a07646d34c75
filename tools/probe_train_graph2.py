"""Step-by-step diagnostics of the captured train step (TrainStep(graph=True)) at the config-5 shape: loss, time and the
finiteness of the parameters after every replay, printed as it goes (a run that degrades shows where)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd import configs  # noqa: E402
from lina_speech_amd.train import TrainStep, synthetic_batch  # noqa: E402

if os.environ.get("PROBE_BLAS"):                     # e.g. hipblas (= rocBLAS path: no hipBLASLt stream-K kernels / memsets)
    torch.backends.cuda.preferred_blas_library(os.environ["PROBE_BLAS"])
dev = torch.device("cuda", 0)
torch.manual_seed(0)
b, T = int(os.environ.get("PROBE_B", "8")), int(os.environ.get("PROBE_T", "4096"))
ts = TrainStep(configs.l169(), device=dev, ddp=False, graph=True)
batch = synthetic_batch(b=b, n=T + 1, t_txt=64, seed=1).to(dev)
for i in range(int(os.environ.get("PROBE_STEPS", "14"))):
    t0 = time.perf_counter()
    loss = float(ts.step(batch))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    bad = [n for n, p in ts.model.named_parameters() if not bool(torch.isfinite(p).all())]
    badg = [n for n, p in ts.model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    print(json.dumps({"step": i, "ms": round(dt, 1), "loss": loss, "nonfinite_params": bad[:4], "n_bad_params": len(bad),
                      "nonfinite_grads": badg[:6], "n_bad_grads": len(badg)}), flush=True)
    if bad or loss != loss:
        break
