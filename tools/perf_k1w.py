#!/usr/bin/env python
"""K1w + K5 (windowed decode update) alone at the decode shape of the bench: all 13 layers' real buffers, every window
position in turn; used under rocprofv3 --pmc for the HBM traffic counters (tests/gpu_traffic.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from lina_speech_amd.configs import l169
from lina_speech_amd.decode import DecodeEngine

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = l169().eval().to(dev, torch.bfloat16)
B = int(os.environ.get("K1_B", 64))
texts = torch.randint(3, 256, (B, 64), generator=torch.Generator().manual_seed(1234)).to(dev)
with torch.inference_mode():
    eng = DecodeEngine(model, model.txt_encoder(model.txt_embed(texts)), batch_size=B, use_graph=False)
    dt, entry = bench.measure_k1(eng, reps=int(os.environ.get("K1_REPS", 16)))
P = eng.packs[0]
nb = bench.k1w_algorithmic_bytes(B, P.H, P.Dk, P.Dv, 2, 4, eng.window)
print(f"{entry}: {dt * 1e6:.2f} us per launch back to back, {nb} algorithmic bytes -> {nb / dt / 1e9:.0f} GB/s")
