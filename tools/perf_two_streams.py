#!/usr/bin/env python
"""Experiment: does the decode step gain from running TWO independent half-batches (32 rows each, own hipGraph each) on
two HIP streams, so that one half's latency-bound launch chain overlaps the other's?  (Forked branches inside ONE graph
did not overlap)  Prints tokens/s for 1 x 64 rows and for 2 x 32 / 4 x 16 rows on separate streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lina_speech_amd.configs import l169
from lina_speech_amd.decode import DecodeEngine

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = l169().eval().to(dev, torch.bfloat16)
B, N = 64, int(os.environ.get("STEPS", 480))
texts = torch.randint(3, 256, (B, 64), generator=torch.Generator().manual_seed(1234)).to(dev)
with torch.inference_mode():
    x_enc = model.txt_encoder(model.txt_embed(texts))
    for parts in (1, 2, 4):
        rows = B // parts
        streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
        engs = []
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                e = DecodeEngine(model, x_enc[i * rows:(i + 1) * rows], batch_size=rows)
                e.begin_greedy(N + 80)
                e.greedy_steps(16)
                engs.append(e)
        torch.cuda.synchronize()
        def run(n):
            for k in range(0, n, 8):                      # interleave the replays of the parts
                for e, st in zip(engs, streams):
                    with torch.cuda.stream(st):
                        e.greedy_steps(8)
        run(64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(N)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{parts} x {rows} rows on {parts} stream(s): {dt / N * 1e3:.4f} ms per step of all 64 rows, {B * N / dt:.0f} tok/s")
        del engs
        torch.cuda.empty_cache()
