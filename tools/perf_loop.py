"""ms per token of the device-side decode loop (the one generate_batch runs), L169 bf16, at a given batch.
    [LINA_TALL=0|1] python tools/perf_loop.py [B] [steps] [window]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lina_speech_amd.configs import l169  # noqa: E402
from lina_speech_amd.decode import DecodeEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
window = int(sys.argv[3]) if len(sys.argv) > 3 else None          # state window of K1w (default 8)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = l169().eval().to(dev, torch.bfloat16)
texts = torch.randint(3, 256, (B, 64), generator=torch.Generator().manual_seed(1234)).to(dev)
with torch.inference_mode():
    eng = DecodeEngine(m, m.txt_encoder(m.txt_embed(texts)), batch_size=B, window=window)
    eng.begin_greedy(steps + 208, log_att=True)
    eng.greedy_steps(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.greedy_steps(steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    toks = eng.greedy_tokens()
print(f"B={B} window={eng.window} LINA_TALL={os.environ.get('LINA_TALL', 'default')}: {ms:.4f} ms per token, {B / ms:.1f} k tok/s, "
      f"token checksum {int(toks.sum())}")
