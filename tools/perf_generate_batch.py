"""generate_batch end to end vs the bare device loop, L169 bf16 (bench.measure_generate_batch on its own).
    python tools/perf_generate_batch.py [B]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lina_speech_amd.configs import l169  # noqa: E402
from lina_speech_amd.decode import DecodeEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = l169().eval().to(dev, torch.bfloat16)
texts = torch.randint(3, 256, (B, 64), generator=torch.Generator().manual_seed(1234)).to(dev)
with torch.inference_mode():
    eng = DecodeEngine(m, m.txt_encoder(m.txt_embed(texts)), batch_size=B)
    out = {}
    for log_att in (False, True):
        eng.begin_greedy(1000, log_att=log_att)
        eng.greedy_steps(200)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.greedy_steps(600)
        torch.cuda.synchronize()
        out[f"loop_ms_log_att_{log_att}"] = (time.perf_counter() - t0) / 600 * 1e3
    del eng
    torch.cuda.empty_cache()
out["generate_batch"] = bench.measure_generate_batch(m, texts, dev, out["loop_ms_log_att_True"])
print(json.dumps(out, indent=1))
