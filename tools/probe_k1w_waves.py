#!/usr/bin/env python
"""A/B of the decode loop with the sixteen-wave K1w (default) and the eight-wave form (LINA_K1W_WAVES=8, LINA_K1W_NPRE = state
vectors requested at entry): ms per token of the bench engine (L169, B = 64, bf16), variants interleaved, tokens compared."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lina_speech_amd import ops
from lina_speech_amd.configs import l169
from lina_speech_amd.decode import DecodeEngine

dev = torch.device("cuda", 0)
B = 64
torch.manual_seed(0)
model = l169().eval().to(dev, torch.bfloat16)
texts = torch.randint(3, 256, (B, 64), generator=torch.Generator().manual_seed(1234)).to(dev)
ops.get_backend().lib
with torch.inference_mode():
    eng = DecodeEngine(model, model.txt_encoder(model.txt_embed(texts)), batch_size=B)


def timed(n=400, warm=100):
    with torch.inference_mode():
        eng.begin_greedy(n + warm + 8)            # (captures the step graph: the knobs are read by the launcher at capture)
        eng.greedy_steps(warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.greedy_steps(n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3


t_pre = time.perf_counter()
while time.perf_counter() - t_pre < 1.5:
    timed(200, 10)
toks = {}
for rnd in range(3):
    for waves, npre in (("16", "0"), ("8", "32"), ("8", "24"), ("8", "16")):
        os.environ["LINA_K1W_WAVES"], os.environ["LINA_K1W_NPRE"] = waves, npre
        ms = timed()
        toks[(waves, npre)] = eng.greedy_tokens().clone()
        print(f"round {rnd}: K1w waves={waves:2s} n_pre={npre:2s}  {ms:.4f} ms/token  {B / ms:.1f} k tok/s", flush=True)
ref = toks[("16", "0")]
print("tokens equal to the sixteen-wave loop's:", {k: bool((v == ref).all()) for k, v in toks.items()})
