#!/usr/bin/env python
"""A/B of the decode loop with the mixer's input side + K1w as TWO launches (lina_gla_decode_inproj_packed +
lina_gla_decode_window) or ONE (lina_gla_decode_inproj_window, n_pre = 16 / 20 / 24 state vectors requested before the hand-off):
ms per token of the bench engine (L169, B = 64, bf16), the variants interleaved twice, same tokens checked."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lina_speech_amd import ops
from lina_speech_amd.configs import l169
from lina_speech_amd.decode import DecodeEngine

dev = torch.device("cuda", 0)
B = int(os.environ.get("PROBE_B", "64"))
torch.manual_seed(0)
model = l169().eval().to(dev, torch.bfloat16)
texts = torch.randint(3, 256, (B, 64), generator=torch.Generator().manual_seed(1234)).to(dev)
ops.get_backend().lib


def build(**kw):
    with torch.inference_mode():
        return DecodeEngine(model, model.txt_encoder(model.txt_embed(texts)), batch_size=B, **kw)


def timed(eng, n=400, warm=100):
    with torch.inference_mode():
        eng.begin_greedy(n + warm + 8)
        eng.greedy_steps(warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.greedy_steps(n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3


variants = [("two launches", dict(one_launch_mixer=False), 0)] + \
           [(f"one launch, n_pre=24 delay={dl / 100:.1f} us", dict(one_launch_mixer=True, n_pre=24, pace=0), dl)
            for dl in (0, 200, 300, 400, 500, 650)]
engines = [(name, build(**kw), dl) for name, kw, dl in variants]
t_pre = time.perf_counter()
while time.perf_counter() - t_pre < 1.5:
    timed(engines[0][1], 200, 10)
toks = {}
for rnd in range(2):
    for name, eng, dl in engines:
        os.environ["LINA_IW_DELAY"] = str(dl)    # read by the launcher at every call: frozen into the graph begin_greedy captures
        ms = timed(eng)
        toks[name] = eng.greedy_tokens().clone()
        bad = sum(int(P.sync[32]) for P in eng._all_packs())
        print(f"round {rnd}: {name:40s} {ms:.4f} ms/token  {B / ms:.1f} k tok/s  hand-off timeouts: {bad}", flush=True)
ref = toks["two launches"]
for name, t in toks.items():
    same = (t == ref).float().mean().item()
    print(f"tokens of '{name}' equal to the two-launch loop's: {same * 100:.2f} % of {t.numel()} positions (random-init weights: near-ties may flip)")
