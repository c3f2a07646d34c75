#!/usr/bin/env python
"""Where does the decode step's time go?  The captured step graph of the bench engine, timed in variants that change ONE
thing each (results are garbage in the shared-buffer variants -- only the time is read):
  base            the bench's loop (one graph replay per token)
  no_update       without the 13 K1w launches
  shared_state    every block streams block 0's state (67 MB: stays in the 256 MB Infinity Cache)
  shared_weights  every block uses block 0's weights (cache-hot weights)
  shared_both
  prefetch_N      a side-stream `touch` kernel (N workgroups) reads block L+1's state while block L's projections run
"""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lina_speech_amd import ops, _lib
from lina_speech_amd.configs import l169
from lina_speech_amd.decode import DecodeEngine

dev = torch.device("cuda", 0)
B = 64
torch.manual_seed(0)
model = l169().eval().to(dev, torch.bfloat16)
texts = torch.randint(3, 256, (B, 64), generator=torch.Generator().manual_seed(1234)).to(dev)

ops.get_backend().lib                                            # the process-wide HIP runtime first
so = os.path.join(ROOT, "tools", "micro", "libtouch.so")
touch = ctypes.CDLL(so)
touch.touch_launch.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device=dev)


def build():
    with torch.inference_mode():
        return DecodeEngine(model, model.txt_encoder(model.txt_embed(texts)), batch_size=B)


def timed(eng, n=400, warm=100):
    with torch.inference_mode():
        eng.begin_greedy(n + warm + 8)
        for _ in range(warm):
            eng.greedy_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.greedy_step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3


def exec_order(eng):
    p = eng.parts[0].packs
    return p[:eng.n_enc] + [p[-1]] + p[eng.n_enc:-1]


MODE = os.environ.get("PROBE", "all")          # base: only the bench loop (for A/B of library builds via LINA_GLA_LIB)
res = {}
eng = build()
t_pre = time.perf_counter()
while time.perf_counter() - t_pre < 1.5:
    timed(eng, 200, 10)
res["base"] = timed(eng)
if MODE == "sampled":                           # the reference's default generation mode (k=100, first quantizer sampled)
    def timed_s(eng, n=400, warm=100, **kw):
        with torch.inference_mode():
            eng.begin_greedy(n + warm + 8, **kw)
            for _ in range(warm):
                eng.greedy_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                eng.greedy_step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
    res["sampled_fused_K6e"] = timed_s(eng, k=100, temp=1.0, seed=1, first_greedy_quant=1)
    toks_f = eng.greedy_tokens().clone()
    eng._fused_pick = False                       # (DecodeEngine(fused_pick=False): the unfused sampled epilogue)
    res["sampled_unfused"] = timed_s(eng, k=100, temp=1.0, seed=1, first_greedy_quant=1)
    toks_u = eng.greedy_tokens().clone()
    eng._fused_pick = True
    res["greedy_again"] = timed(eng)
    print(" ".join(f"{k_}={v_:.4f}" for k_, v_ in res.items()), "ms/step; fused vs unfused tokens equal:",
          bool(torch.equal(toks_f, toks_u)), " first divergence step:",
          (int((toks_f != toks_u).any(dim=(0, 1)).float().argmax()) if not torch.equal(toks_f, toks_u) else None))
    sys.exit(0)
if MODE == "base":
    res["base_again"] = timed(eng)
    print(" ".join(f"{k_}={v_:.4f}" for k_, v_ in res.items()), "ms/step  lib=", os.environ.get("LINA_GLA_LIB", "(product)"),
          " narrow_inproj=", os.environ.get("LINA_INPROJ_NARROW"))
    sys.exit(0)
eng._skip_update = True
res["no_update"] = timed(eng)
eng._skip_update = False

def share_state(e):
    P0 = e.packs[0]
    for P in e.packs[1:]:
        P.S, P.hk, P.hc, P.hv = P0.S, P0.hk, P0.hc, P0.hv

def share_weights(e):
    P0 = e.packs[0]
    for P in e.packs[1:]:
        for a in ("w_in_p", "w_o_p", "w_up_p", "w_down_p", "c1_in", "c2_in", "c1_up", "c2_up", "wq", "wk", "wv", "w2", "b2", "gnw"):
            setattr(P, a, getattr(P0, a))

e = build(); share_state(e); res["shared_state"] = timed(e)
if MODE == "state":
    for k_, v_ in res.items():
        print(f"{k_:24s} {v_:.4f} ms/step   lib=", os.environ.get("LINA_GLA_LIB", "(product)"))
    sys.exit(0)
e._skip_update = True; res["shared_state_no_update"] = timed(e); del e
e = build(); share_weights(e); res["shared_weights"] = timed(e); del e
e = build(); share_state(e); share_weights(e); res["shared_both"] = timed(e); del e

orig = ops.gla_decode_window
for nblk in (64, 256, 1024):
    e = build()
    order = exec_order(e)
    nxt = {order[i].S.data_ptr(): order[i + 1].S for i in range(len(order) - 1)}
    side = torch.cuda.Stream(device=dev)
    pending = [False]

    def wrapped(q, k, v, gk, state, *a, **kw):
        main = torch.cuda.current_stream(dev)
        if pending[0]:
            main.wait_stream(side)
            pending[0] = False
        out = orig(q, k, v, gk, state, *a, **kw)
        S2 = nxt.get(state.data_ptr())
        if S2 is not None:
            side.wait_stream(main)
            touch.touch_launch(S2.data_ptr(), S2.numel() * 4, nblk, sink.data_ptr(), side.cuda_stream)
            pending[0] = True
        return out

    ops.gla_decode_window = wrapped
    try:
        res[f"prefetch_{nblk}"] = timed(e)
    finally:
        ops.gla_decode_window = orig
    del e

for k_, v_ in res.items():
    print(f"{k_:24s} {v_:.4f} ms/step")
