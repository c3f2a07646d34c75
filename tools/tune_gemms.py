"""Vendor-GEMM algorithm selection for the train step's library GEMMs (torch's TunableOp over hipBLASLt): ONE tuning step on
the L169 shapes (b = 8 x 4096), the result table written to FILE, then the step timed with and without the table.
    python tools/tune_gemms.py FILE [steps]          (tune + time)
    python tools/tune_gemms.py FILE [steps] --use    (time only, with FILE as a read-only table)"""
import json
import os
import sys
import time

os.environ.setdefault("PYTORCH_TUNABLEOP_ROCBLAS_ENABLED", "0")
import torch
import torch.cuda.tunable as tunable

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd import configs  # noqa: E402
from lina_speech_amd.train import TrainStep, synthetic_batch  # noqa: E402

FILE = sys.argv[1]
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
USE = "--use" in sys.argv
dev = torch.device("cuda", 0)
torch.manual_seed(0)
ts = TrainStep(configs.l169(), device=dev, ddp=False)
batch = synthetic_batch(b=8, n=4097, t_txt=64, seed=1).to(dev)


def timed(n):
    for _ in range(2):
        ts.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        loss = ts.step(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(loss)


out = {}
if not USE:
    out["untuned_ms"], out["untuned_loss"] = timed(STEPS)
    print(json.dumps(out), flush=True)
tunable.enable(True)
tunable.set_filename(FILE)
if USE:
    tunable.tuning_enable(False)
    tunable.read_file(FILE)
else:
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(30)
    tunable.set_max_tuning_iterations(20)
    t0 = time.perf_counter()
    ts.step(batch)
    torch.cuda.synchronize()
    out["tuning_step_s"] = time.perf_counter() - t0
    tunable.tuning_enable(False)
    print(json.dumps(out), flush=True)
out["tuned_ms"], out["tuned_loss"] = timed(STEPS)
out["entries"] = len(tunable.get_results())
print(json.dumps(out), flush=True)
