"""Module / model level checks against the golden vectors captured from the reference's own
in-tree modules (tests/golden/make_golden.py).  `dev` = "cpu" (emulator) or "cuda" (HIP).

Tolerance: golden activations were produced in fp32 by the reference glue + CPU oracle, the
kernels accumulate in a different order -> 2e-4 relative to max|golden| (fp32 model);
token ids, stop flags, shapes: exact.
"""
import os

import numpy as np
import torch

from lina_speech_amd.attentive import AttentiveGLA
from lina_speech_amd.blocks import TextEncoder
from lina_speech_amd.lina_model import LinaModel
from lina_speech_amd.mixer import GatedLinearAttention
from lina_speech_amd.modules import Cache

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL = 2e-4


def load_golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name))


def golden_state_dict(g):
    sd = {}
    for k in g.files:
        if not k.startswith("sd::"):
            continue
        name = k[4:]
        if name.endswith("::first_rows"):
            base = name[: -len("::first_rows")]
            full = int(g["sd::" + base + "::full_rows"])
            rows = torch.from_numpy(g[k])
            w = torch.zeros(full, rows.shape[1])
            w[: rows.shape[0]] = rows
            sd[base] = w
        elif name.endswith("::full_rows"):
            continue
        else:
            sd[name] = torch.from_numpy(g[k])
    return sd


def close(got, ref, what, rel=REL):
    got = got.detach().float().cpu()
    ref = torch.as_tensor(ref).float()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    err = (got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)
    from kernel_cases import record_parity
    record_parity(what, err, rel)
    assert err <= rel, f"{what}: rel err {err:.3e} > {rel:.1e}"


def build_lina(d=64, n_layer=1, heads=1, ev=1.0, n_codebook=253, txt_layers=1):
    rnn = AttentiveGLA(d_model=d, n_layer=n_layer, heads=heads, blind=True, use_short_conv=True, expand_k=1.0,
                       expand_v=ev, pos_type="convolutional")
    txt = TextEncoder(d, heads, n_layers=txt_layers, dropout=0.0, rotary=False)
    return LinaModel(rnn, d_model=d, n_quant=1, n_codebook=n_codebook, n_special_token_in=3, n_special_token_out=3,
                     n_txt_vocab=256, txt_encoder=txt)


def check_mixer_golden(dev):
    g = load_golden("mixer_d128.npz")
    m = GatedLinearAttention(mode="fused_chunk", hidden_size=128, num_heads=2, expand_k=1.0, expand_v=2.0,
                             use_short_conv=True, layer_idx=0)
    missing = m.load_state_dict(golden_state_dict(g), strict=True)   # same keys as the reference module
    m = m.to(dev).eval()
    x = torch.from_numpy(g["x"]).to(dev)
    T = 21
    with torch.no_grad():
        for mode in ("fused_chunk", "chunk", "fused_recurrent"):
            m.mode = mode
            close(m(x[:, :T]), g["o_" + mode], f"mixer {mode}")
        m.mode = "fused_chunk"
        close(m(x[:, :T], reset_mask=torch.from_numpy(g["reset_mask"]).to(dev)), g["o_reset"], "mixer reset")
        cache = Cache()
        cache.update(m.init_state(3), 0, offset=0)
        m.mode = "fused_recurrent"
        close(m(x[:, :T], past_key_values=cache, use_cache=True), g["o_prefill_cached"], "mixer cached prefill")
        for i in range(3):
            close(m(x[:, T + i:T + i + 1], past_key_values=cache, use_cache=True), g[f"o_step{i}"], f"mixer step {i}")
        for j, s in enumerate(cache.states[0]):
            close(s, g[f"cache_after_{j}"], f"mixer cache[{j}]")


def check_lina_golden(dev, engine=None):
    g = load_golden("lina_d64.npz")
    model = build_lina()
    model.load_state_dict(golden_state_dict(g), strict=True)
    model = model.to(dev).eval()
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    with torch.no_grad():
        if engine is None:
            logits, loss, att, _, _ = model(t("x"), t("y"), t("encoder_mask"), t("crossatt_mask"),
                                            logits_mask=t("logits_mask"))
            close(logits, g["fwd_logits"], "forward logits")
            close(loss, g["fwd_loss"], "forward loss")
            close(att, g["fwd_att"], "forward att")
        qs, atts, stops, cuts = model.generate_batch(t("gen_x"), batch_size=3, max_seqlen=12, k=1,
                                                     first_greedy_quant=0, force_max_seqlen=True, device=dev,
                                                     engine=engine)
        assert torch.equal(qs.cpu(), torch.from_numpy(g["gen_qs"])), "greedy token ids differ from the reference"
        assert torch.equal(stops.cpu(), torch.from_numpy(g["gen_stop_tokens"]))
        close(atts, g["gen_atts"], "generate atts")
        assert [c[0].shape[-1] for c in cuts] == list(g["gen_cut_lens"])
        qs2, atts2, st2, _ = model.generate_batch(t("gen_x"), batch_size=3, prompt=t("gen_prompt"), max_seqlen=9,
                                                  k=1, first_greedy_quant=0, force_max_seqlen=True, device=dev,
                                                  engine=engine)
        assert torch.equal(qs2.cpu(), torch.from_numpy(g["gen_prompt_qs"])), "prompted token ids differ"
        close(atts2, g["gen_prompt_atts"], "prompted atts")
        if engine is None:
            # teacher-forced step loop == reference step logits, and the cache layout/content
            x_enc = model.txt_encoder(model.txt_embed(t("x")))
            state = model.attentive_rnn.init_state(batch_size=3)
            y_embd = model.rvq_embed.embed_sum(t("y").permute(2, 0, 1))
            outs = []
            for i in range(y_embd.shape[1] - 1):
                h, a, state = model.attentive_rnn.step(y_embd[:, i:i + 1], x_enc, i, state)
                outs.append(model.logits_head(h))
            close(torch.cat(outs, 1), g["step_logits"], "step logits")
            assert len(state.states) == 3
            for li, st in enumerate(state.states):
                assert len(st) == 4
                for j, s in enumerate(st):
                    close(s, g[f"cache_{li}_{j}"], f"cache[{li}][{j}]")


def check_lina_train_golden(dev, rel=5e-4):
    """a-11: one teacher-forced training forward + backward in train() mode reproduces the reference's loss and
    EVERY parameter gradient (golden captured from the reference modules under torch autograd,
    tests/golden/make_golden.py).  fp32 model; tolerance 5e-4 of max|golden gradient| per tensor (gradients
    pass through K2b/K3b/K5b whose summation order differs from the oracle's)."""
    g = load_golden("lina_d64.npz")
    model = build_lina()
    model.load_state_dict(golden_state_dict(g), strict=True)
    model = model.to(dev).train()
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    logits, loss, att, _, _ = model(t("x"), t("y"), t("encoder_mask"), t("crossatt_mask"), logits_mask=t("logits_mask"))
    close(loss, g["train_loss"], "train loss", 1e-5)
    loss.backward()
    names = [k[6:] for k in g.files if k.startswith("grad::")]
    assert names, "golden has no gradients"
    params = dict(model.named_parameters())
    checked = 0
    for name in names:
        ref = g["grad::" + name]
        p = params[name]
        assert p.grad is not None, f"no gradient for {name}"
        if np.abs(ref).max() < 1e-8:        # analytically zero (e.g. a key bias under softmax): rounding noise only
            assert float(p.grad.abs().max()) < 1e-7, name
            continue
        close(p.grad, ref, f"grad {name}", rel)
        checked += 1
    assert checked >= len(names) - 4
    return loss


def check_engine_sampling(dev, n_steps=10, k=20, temp=0.9, seed=77):
    """f-2: the device-side SAMPLED decode loop (top-k + temperature, K6c inside the captured step).  The CPU oracle is
    teacher-forced with the engine's tokens; at every step the engine's token must be the inverse-CDF pick of the
    oracle's logits for the same hashed uniform number (wherever that number is not within 1e-4 of a CDF edge --
    fp32 logits of two implementations differ by ~1e-5), and a different seed must give a different stream."""
    from oracle import gla_oracle as O
    from oracle.lina_decode_oracle import OracleLina
    from lina_speech_amd.decode import DecodeEngine
    g = load_golden("lina_d64.npz")
    model = build_lina()
    sd = golden_state_dict(g)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    x = torch.from_numpy(g["gen_x"]).unsqueeze(0).expand(3, -1).contiguous()
    with torch.no_grad():
        x_enc = model.txt_encoder(model.txt_embed(x.to(dev)))
        toks = DecodeEngine(model, x_enc, batch_size=3).run_greedy(n_steps, k=k, temp=temp, seed=seed,
                                                                   first_greedy_quant=1).cpu()
        toks_b = DecodeEngine(model, x_enc, batch_size=3).run_greedy(n_steps, k=k, temp=temp, seed=seed + 1,
                                                                     first_greedy_quant=1).cpu()
        toks_g = DecodeEngine(model, x_enc, batch_size=3).run_greedy(n_steps, k=k, temp=temp, seed=seed,
                                                                     first_greedy_quant=0).cpu()
    assert toks.shape == (1, 3, n_steps)
    assert not torch.equal(toks, toks_b), "another seed must give another token stream"
    assert torch.equal(toks_g, torch.from_numpy(g["gen_qs"])[:, :, :n_steps]), "first_greedy_quant=0 must stay greedy"
    orc = OracleLina(sd, n_layer=1, heads=1)
    _, logits, _, _ = orc.generate_greedy(x, n_steps, teacher=toks)           # [B,n,q,l]
    compared = 0
    for t in range(n_steps):
        u = torch.tensor([O.hash_uniform(seed, t, r, 3) for r in range(3)], dtype=torch.float64)
        rt, margin, p = O.topk_sample_inverse_cdf(logits[:, t, 0].float(), k, temp, u)
        for r in range(3):
            assert p[r, toks[0, r, t]] > 0, "sampled token outside the oracle's top-k set"
            if margin[r] > 1e-4:
                assert int(toks[0, r, t]) == int(rt[r]), (t, r)
                compared += 1
    assert compared >= 2 * n_steps


def check_generate_batch_loop(dev, rel=REL, full=True):
    """a-10: ``LinaModel.generate_batch`` with NO engine argument runs the device-side loop (decode.DecodeEngine.generate:
    picks, stop flags, attention log and next-token embedding inside the step, stop test read back every few steps) and
    returns what the per-token module path returns with the reference's per-step stop test (model/modeling_lina.py:152-192):
    token ids, stop flags and cut lengths exactly, attention rows within ``rel`` -- on a model whose stop-token head row is
    scaled up so that rows stop early and at DIFFERENT steps, with and without a codec prompt; the engine (packed weights,
    captured graphs) is built once per (batch, text length) and re-armed for other texts, and rebuilt when a weight changes."""
    g = load_golden("lina_d64.npz")
    model = build_lina()
    model.load_state_dict(golden_state_dict(g), strict=True)
    with torch.no_grad():
        model.logits_head.weight[0, 2] *= 6.0
    model = model.to(dev).eval()
    B = 4
    seen_n, engines = set(), []
    cases = ((1, 0, 4), (4, 0, 4), (2, 3, 8), (3, 0, 16), (1, 0, 1)) if full else ((4, 0, 4), (2, 3, 16))   # (seed, prompt length, check every)
    for ci, (seed, prompt_len, every) in enumerate(cases):
        gen = torch.Generator().manual_seed(seed)
        x = torch.randint(3, 256, (B, 9), generator=gen).to(dev)
        prompt = torch.randint(0, 250, (1, 1, prompt_len), generator=gen).to(dev) if prompt_len else None
        kw = dict(batch_size=B, prompt=prompt, max_seqlen=48 if full else 20, k=1, first_greedy_quant=0, device=dev)
        ref = model.generate_batch(x, engine="module", stop_check_every=1, **kw)
        got = model.generate_batch(x, stop_check_every=every, **kw)
        engines.append(next(reversed(model._decode_engines.values())))
        assert torch.equal(got[0], ref[0]), f"token ids differ (seed {seed})"
        assert torch.equal(got[2], ref[2]), "stop flags differ"
        close(got[1], ref[1].cpu(), "generate_batch loop atts", rel)
        assert [c[0].shape for c in got[3]] == [c[0].shape for c in ref[3]]
        for cg, cr in zip(got[3], ref[3]):
            assert torch.equal(cg[0], cr[0]) and cg[1].shape == cr[1].shape
        seen_n.add(int(got[0].shape[-1]))
        assert got[0].shape[-1] < 20, "this model must stop early"
        if ci and not full:
            continue
        forced = model.generate_batch(x, force_max_seqlen=True, **{**kw, "max_seqlen": 11})
        assert forced[0].shape[-1] == 11 and torch.equal(forced[0][:, :, :min(11, got[0].shape[-1])],
                                                        got[0][:, :, :11])
    assert len(seen_n) >= (3 if full else 2), seen_n
    assert all(e is engines[0] for e in engines), "same (batch, text length, weights): one engine, re-armed"
    with torch.no_grad():
        model.logits_head.weight[0, 5].mul_(1.5)                 # a weight changed: the cached engine is stale
    x = torch.randint(3, 256, (B, 9), generator=torch.Generator().manual_seed(1)).to(dev)
    kw = dict(batch_size=B, max_seqlen=10 if full else 5, k=1, first_greedy_quant=0, device=dev)
    got = model.generate_batch(x, **kw)
    assert next(reversed(model._decode_engines.values())) is not engines[0]
    assert torch.equal(got[0], model.generate_batch(x, engine="module", stop_check_every=1, **kw)[0])
    if not full:
        return                                               # (the emulator is slow: the sampled-mode part runs on the GPU)
    # the reference's default mode (k = 100, first quantizer sampled): reproducible under torch.manual_seed, another seed differs
    torch.manual_seed(5)
    a = model.generate_batch(x, batch_size=B, max_seqlen=6, device=dev, force_max_seqlen=True)
    torch.manual_seed(5)
    b = model.generate_batch(x, batch_size=B, max_seqlen=6, device=dev, force_max_seqlen=True)
    c = model.generate_batch(x, batch_size=B, max_seqlen=6, device=dev, force_max_seqlen=True)
    assert torch.equal(a[0], b[0]) and not torch.equal(a[0], c[0])


def check_generate_batch_group(dev, rel=REL):
    """``generate_batch(..., n_engines=2)``: the batch cut into two row ranges, one DecodeEngine (and one HIP stream) each
    (decode.DecodeEngineGroup) == the per-token module path with the reference's per-step stop test -- rows never interact, and
    the loop must end at the step where the LAST of the two engines has seen all of its rows stop."""
    from lina_speech_amd.decode import DecodeEngineGroup
    g = load_golden("lina_d64.npz")
    model = build_lina()
    model.load_state_dict(golden_state_dict(g), strict=True)
    with torch.no_grad():
        model.logits_head.weight[0, 2] *= 6.0
    model = model.to(dev).eval()
    B = 5                                                   # 3 + 2 rows
    for seed, every in (((4, 4), (1, 16)) if dev != "cpu" else ((4, 4),)):      # (the emulator is slow: one case there)
        x = torch.randint(3, 256, (B, 9), generator=torch.Generator().manual_seed(seed)).to(dev)
        kw = dict(batch_size=B, max_seqlen=20, k=1, first_greedy_quant=0, device=dev)
        ref = model.generate_batch(x, engine="module", stop_check_every=1, **kw)
        got = model.generate_batch(x, stop_check_every=every, n_engines=2, **kw)
        eng = next(reversed(model._decode_engines.values()))
        assert isinstance(eng, DecodeEngineGroup) and [hi - lo for lo, hi in eng.ranges] == [3, 2]
        assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2]), f"tokens / stop flags differ (seed {seed})"
        close(got[1], ref[1].cpu(), "generate_batch with two engines: atts", rel)
        assert [c[0].shape for c in got[3]] == [c[0].shape for c in ref[3]]
        assert got[0].shape[-1] < 20
    forced = model.generate_batch(x, force_max_seqlen=True, n_engines=2, **{**kw, "max_seqlen": 9})
    assert forced[0].shape[-1] == 9 and torch.equal(forced[0][:, :, :got[0].shape[-1]], got[0][:, :, :9])
    if dev == "cpu":
        return
    torch.manual_seed(7)
    a = model.generate_batch(x, batch_size=B, max_seqlen=5, device=dev, force_max_seqlen=True, n_engines=2)
    torch.manual_seed(7)
    b = model.generate_batch(x, batch_size=B, max_seqlen=5, device=dev, force_max_seqlen=True, n_engines=2)
    assert torch.equal(a[0], b[0])


def check_generate_batch_outputs_are_fresh(dev):
    """(1) the tensors a ``generate_batch`` call returns are the caller's -- not views of the cached engine's static
    logs, which the next call of the same shape overwrites (the case that bit: max_seqlen a multiple of 64 and no early stop, where
    the trimmed log IS the whole log); (2) a write through ``param.data`` (an EMA swap) is seen by the engine cache key (content
    fingerprint), so the next call decodes with the new weights; (3) the per-engine set of captured loop configurations stays
    bounded when a caller sweeps the temperature."""
    from lina_speech_amd.decode import DecodeEngine
    g = load_golden("lina_d64.npz")
    model = build_lina()
    model.load_state_dict(golden_state_dict(g), strict=True)
    model = model.to(dev).eval()
    B, n = (1 if dev == "cpu" else 2), 64                     # (the emulator is slow)
    kw = dict(batch_size=B, max_seqlen=n, k=1, first_greedy_quant=0, device=dev, force_max_seqlen=True)
    xa = torch.randint(3, 256, (B, 9), generator=torch.Generator().manual_seed(11)).to(dev)
    xb = torch.randint(3, 256, (B, 9), generator=torch.Generator().manual_seed(12)).to(dev)
    a = model.generate_batch(xa, **kw)
    eng = next(reversed(model._decode_engines.values()))
    assert isinstance(eng, DecodeEngine) and eng._loop.cap == n
    assert a[1].data_ptr() != eng._loop.att_log.data_ptr(), "generate_batch returned the engine's static attention log"
    keep = [t.clone() for t in (a[0], a[1], a[2])] + [c[1].clone() for c in a[3]]
    # same shape: the cached engine is re-armed and overwrites its logs (the emulator is slow: 8 steps there -- the first 8 log
    # entries are rewritten all the same)
    kw2 = kw if dev != "cpu" else {**kw, "max_seqlen": 8}
    b = model.generate_batch(xb, **kw2)
    assert next(reversed(model._decode_engines.values())) is eng
    assert not torch.equal(b[1][:, :, :8], keep[1][:, :, :8])
    for was, now in zip(keep, [a[0], a[1], a[2]] + [c[1] for c in a[3]]):
        assert torch.equal(was, now), "a later generate_batch call changed the tensors an earlier one returned"
    # (2) EMA-style swap through param.data: neither the storage nor the version counter changes
    with torch.no_grad():
        model.logits_head.weight.data.mul_(-1.0)
    c = model.generate_batch(xb, **{**kw, "max_seqlen": 8})
    assert next(reversed(model._decode_engines.values())) is not eng, "stale packed weights: the cache key missed a param.data write"
    ref = model.generate_batch(xb, engine="module", **{**kw, "max_seqlen": 8})
    assert torch.equal(c[0], ref[0])
    with torch.no_grad():
        model.logits_head.weight.data.mul_(-1.0)
    # (3) bounded loop cache
    with torch.inference_mode():
        eng2 = DecodeEngine(model, model.txt_encoder(model.txt_embed(xa)), batch_size=B)
        for t in range(DecodeEngine.MAX_LOOPS + 3):
            eng2.begin_greedy(8, k=4, temp=1.0 + 0.1 * t, first_greedy_quant=1)
            assert len(eng2._loops) <= DecodeEngine.MAX_LOOPS
        eng2.close()
    model.clear_decode_cache()
    assert not model.__dict__.get("_decode_engines")


def check_engine_hidden_log(dev, n=5):
    """``begin_greedy(log_hidden=True)``: the pre-head hidden state filed by the device loop (fragment-major inside the loop)
    == the residual stream the generic step API leaves behind when it is teacher-forced with the loop's tokens -- and the tokens
    are the ones the loop decodes without the log.  (The GPU parity tests compare this log with the oracle's hidden states.)"""
    from lina_speech_amd.decode import DecodeEngine
    g = load_golden("lina_d64.npz")
    model = build_lina()
    model.load_state_dict(golden_state_dict(g), strict=True)
    model = model.to(dev).eval()
    B = 3
    x = torch.randint(3, 256, (B, 9), generator=torch.Generator().manual_seed(5)).to(dev)
    with torch.inference_mode():
        x_enc = model.txt_encoder(model.txt_embed(x))
        eng = DecodeEngine(model, x_enc, batch_size=B)
        plain = eng.run_greedy(n)
        eng.reset()
        eng.begin_greedy(n, log_hidden=True)
        eng.greedy_steps(n)
        toks, hid = eng.greedy_tokens(), eng.logged_hidden(n)
        assert torch.equal(toks, plain) and hid.shape == (n, B, eng.d)
        eng2 = DecodeEngine(model, x_enc, batch_size=B)
        y = model.rvq_embed.embed_sum(torch.ones(eng.Q, B, 1, dtype=torch.long, device=dev))
        for t in range(n):
            logits, _ = eng2(y, t)
            close(hid[t], eng2.parts[0].x.cpu(), f"hidden-state log, step {t}", 2e-5)
            close(torch.einsum("bd,qld->bql", hid[t].float(), model.logits_head.weight.float()), logits[:, 0].float().cpu(),
                  f"logits implied by the hidden-state log, step {t}", 2e-5)
            y = model.rvq_embed.embed_sum(toks[:, :, t:t + 1])


def check_engine_bf16_state(dev, n=6):
    """Opt-in ``DecodeEngine(state_dtype=torch.bfloat16)`` (the reference's state dtype for a bf16 model, model/gla.py:229-240 +
    Cache.update): the device loop at window 1 (rounded after every step) and at window 4, and the generic step API on a bf16
    state, against the oracle that rounds its state to bf16 after every step (fp32 arithmetic on the same bf16-rounded weights),
    teacher-forced on the loop's tokens."""
    from lina_speech_amd.decode import DecodeEngine
    from oracle.lina_decode_oracle import OracleLina
    g = load_golden("lina_d64.npz")
    model = build_lina()
    model.load_state_dict(golden_state_dict(g), strict=True)
    model = model.to(torch.bfloat16).to(dev).eval()
    sd = {k: v.float().cpu() for k, v in model.state_dict().items()}
    B = 3
    x = torch.randint(3, 256, (B, 9), generator=torch.Generator().manual_seed(5))
    orc = OracleLina(sd, n_layer=1, heads=1, state_dtype=torch.bfloat16)
    with torch.inference_mode():
        x_enc = model.txt_encoder(model.txt_embed(x.to(dev)))
        for window in (1, 4):
            eng = DecodeEngine(model, x_enc, batch_size=B, state_dtype=torch.bfloat16, window=window)
            assert all(P.S.dtype == torch.bfloat16 and P.lazy for P in eng.packs)
            eng.begin_greedy(n, log_hidden=True)
            eng.greedy_steps(n)
            toks, hid = eng.greedy_tokens().cpu(), eng.logged_hidden(n).float().cpu()
            assert eng.state.states[0][3].dtype == torch.bfloat16
            ref_toks, ref_logits, _, margins = orc.generate_greedy(x, n, teacher=toks)
            lg = torch.einsum("nbd,ld->bnl", hid, sd["logits_head.weight"][0])
            close(lg, ref_logits[:, :, 0], f"bf16-state engine (window {window}): logits vs the bf16-state oracle", 4e-2)
            close(eng.state.states[0][3], orc.final_state[0][3], f"bf16-state engine (window {window}): first block's state", 4e-2)
            eng.close()
        # generic step API (immediate update of the bf16 state through K1w at window 1)
        eng = DecodeEngine(model, x_enc, batch_size=B, state_dtype=torch.bfloat16)
        y = model.rvq_embed.embed_sum(torch.ones(eng.Q, B, 1, dtype=torch.long, device=dev))
        for t in range(n):
            logits, _ = eng(y, t)
            close(logits[:, 0, 0], ref_logits[:, t, 0], f"bf16-state engine, generic step {t}", 4e-2)
            y = model.rvq_embed.embed_sum(toks[:, :, t:t + 1].to(dev))
        # fp32 models refuse the option
        try:
            DecodeEngine(model.float(), x_enc.float(), batch_size=B, state_dtype=torch.bfloat16)
            raise AssertionError("a bf16 state on an fp32 model must be refused")
        except ValueError:
            pass


def check_generate_batch_early_stop(dev, dtype=torch.float32, d=256, B=8, max_seqlen=160, need_late_stop=True):
    """a-10, the early-stop path of the device loop over MANY stop checks (reference model/modeling_lina.py:168-173): a small
    vocabulary (13 codes + 3 specials) in the reference's default SAMPLED mode makes every row emit the stop token (id 2) at a
    random step.  For a seed, the run with ``force_max_seqlen=True`` is the ground truth (same seed, same kernels -> the same
    token stream): the early-stopping run must return exactly its first n steps, n = the first step at which every row has
    emitted the stop token at some step so far, + 1 -- tokens and attention rows bit-exact, stop flags and cuts as the
    reference's post-processing derives them -- for several ``stop_check_every`` (the loop runs past the stop by up to two
    groups of steps and trims)."""
    from lina_speech_amd.configs import tiny
    torch.manual_seed(2)
    model = tiny(d=d, heads=2, n_layer=2, n_codebook=13).eval().to(dev).to(dtype)
    x = torch.randint(3, 256, (B, 12), generator=torch.Generator().manual_seed(3)).to(dev)
    kw = dict(batch_size=B, max_seqlen=max_seqlen, k=16, temp=1.0, first_greedy_quant=1, device=dev)
    late, checked = 0, 0
    for seed in range(6):
        full = model.generate_batch(x, force_max_seqlen=True, seed=seed, **kw)
        assert full[0].shape == (1, B, max_seqlen) and full[1].shape[2] == max_seqlen
        all_seen = (full[0] == 2).all(dim=0).cummax(dim=1).values.all(dim=0)        # [n]
        if not bool(all_seen.any()):
            continue
        n = int(all_seen.int().argmax()) + 1
        late += n >= 40
        for every in (8, 16, 40):
            got = model.generate_batch(x, seed=seed, stop_check_every=every, **kw)
            assert got[0].shape[-1] == n, (seed, every, got[0].shape, n)
            assert torch.equal(got[0], full[0][:, :, :n]) and torch.equal(got[1], full[1][:, :, :n])
            st = torch.cat([(full[0][:, :, :n] == 2).all(dim=0).float(), torch.ones(B, 1, device=full[0].device)], dim=1)
            assert torch.equal(got[2], st)
            idx = (st * torch.arange(n + 1, device=st.device)[None]).long()
            for i in range(B):
                cut = int(torch.unique(idx[i])[1])                                  # the reference's formula, row by row
                assert got[3][i][0].shape[-1] == len(range(max(n - 2, 0))[:cut - 1]) and got[3][i][1].shape[1] == min(cut, n)
            checked += 1
    assert checked >= 6, "too few seeds stopped early: the check did not bite"
    if need_late_stop:
        assert late >= 1, "no seed stopped after step 40: the stop check never ran more than a few times"


def check_config5_structured_golden(dev, dtype=torch.float32, rel_loss=2e-4, rel_grad=5e-3, min_cos=0.97, max_norm_err=0.10):
    """The config-5 slice (L169 width, 3 GLA blocks, b = 1, T = 4096) on a WELL-CONDITIONED problem -- targets that follow a
    bigram chain (structured_targets), so that every block's gradient is a sum of coherent terms instead of 4096 cancelling
    ones -- against the REFERENCE's fp32 autograd (tests/golden/l169_slice_T4096_structured.npz: loss, and per parameter the
    gradient's norm, max and 4096 strided entries).
      fp32: every sampled entry within ``rel_grad`` of max|golden| (as check_config5_slice_golden);
      bf16 autocast (the way training runs: K2 / K2b full-head kernels, bf16 MFMA): per tensor COSINE of the sampled entries
      with the fp32 reference >= ``min_cos`` and norm error <= ``max_norm_err``.  On this problem the reference's own modules
      under torch.autocast(bfloat16) reach cosine 0.975-0.99 (median 0.984) and norm errors <= 1.5 % against their fp32
      gradients (stored in the golden); this path measured 0.977 and, on ONE tensor (the pos_net block's rank-16 gate
      projection, a small gradient that only the positional attention path feeds), a 7.2 % norm error -- hence 0.10, with the
      achieved figure recorded; a kernel that loses 25 % of a gradient component lands below either bound.  This replaces the
      round-4 rule "1.6 x the reference's bf16 deviation + 0.02" (0.3-0.6 relative on single entries of the random-target run)."""
    from lina_speech_amd.configs import l169
    from kernel_cases import record_parity
    g = load_golden("l169_slice_T4096_structured.npz")
    torch.manual_seed(0)
    model = reseed_parameters(l169(n_layer=1, txt_layers=1), seed=int(g["seed"])).to(dev).train()
    x, y = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["y"]).to(dev)
    assert torch.equal(y.cpu(), structured_targets(1, y.shape[1], 4099, seed=29)), "the golden's targets are the documented chain"
    em = torch.ones(x.shape[0], x.shape[1], x.shape[1], dtype=torch.bool, device=dev)
    cm = torch.ones(x.shape[0], y.shape[1], x.shape[1], dtype=torch.bool, device=dev)
    lm = torch.ones(x.shape[0], y.shape[1], dtype=torch.bool, device=dev)
    model.zero_grad()
    if dtype == torch.float32:
        _, loss, _, _, _ = model(x, y, em, cm, logits_mask=lm)
    else:
        with torch.autocast("cuda", dtype=dtype):
            _, loss, _, _, _ = model(x, y, em, cm, logits_mask=lm)
    loss.backward()
    tag = f"config-5 slice T=4096, bigram targets ({str(dtype)[6:]})"
    e = abs(float(loss.detach()) - float(g["loss"])) / abs(float(g["loss"]))
    record_parity(f"{tag}: loss vs reference autograd", e, rel_loss)
    assert e <= rel_loss, (float(loss), float(g["loss"]))
    names = [k[6:] for k in g.files if k.startswith("gsam::")]
    largest = max(float(g["gmax::" + n]) for n in names)
    ref_cos = dict(zip([str(n) for n in g["bf16ref_names"]], g["bf16ref_cos"].tolist()))
    grads = dict(model.named_parameters())
    worst = {"cos": (2.0, None), "norm": (0.0, None), "max": (0.0, None)}
    n_checked = 0
    for name in names:
        grad = grads[name].grad
        assert grad is not None, name
        nrm, mx, sam = grad_digest(grad, n=4096)
        ref_sam, ref_mx, ref_nrm = torch.from_numpy(g["gsam::" + name]), float(g["gmax::" + name]), float(g["gnorm::" + name])
        if ref_mx < 1e-5 * largest:                         # analytically zero: noise on both sides
            assert mx < 1e-3 * largest, (name, mx)
            continue
        n_checked += 1
        cos = float((sam @ ref_sam) / (sam.norm() * ref_sam.norm()).clamp_min(1e-30))
        en = abs(nrm - ref_nrm) / ref_nrm
        es = float((sam - ref_sam).abs().max()) / ref_mx
        if cos < worst["cos"][0]:
            worst["cos"] = (cos, name)
        if en > worst["norm"][0]:
            worst["norm"] = (en, name)
        if es > worst["max"][0]:
            worst["max"] = (es, name)
    assert n_checked > 80
    if dtype == torch.float32:
        record_parity(f"{tag}: worst |sampled gradient - golden| / max|golden| over {n_checked} tensors", worst["max"][0], rel_grad,
                      tensor=worst["max"][1], lowest_cosine=worst["cos"][0])
        assert worst["max"][0] <= rel_grad, worst
    else:
        record_parity(f"{tag}: 1 - lowest per-tensor cosine with the fp32 reference gradients over {n_checked} tensors",
                      1.0 - worst["cos"][0], 1.0 - min_cos, tensor=worst["cos"][1],
                      reference_bf16_autocast_cosine_of_that_tensor=ref_cos.get(worst["cos"][1]),
                      reference_bf16_autocast_lowest_cosine=min(v for k, v in ref_cos.items() if abs(v) > 0.5))
        record_parity(f"{tag}: worst per-tensor gradient-norm error vs the fp32 reference", worst["norm"][0], max_norm_err,
                      tensor=worst["norm"][1])
        assert worst["cos"][0] >= min_cos and worst["norm"][0] <= max_norm_err, worst


def check_init_state_tuning_golden(dev, rel=5e-4):
    """f-4: loss and the gradients of the rank-1 start-state parameters (dh0 out of K2b, through
    get_state_from_params) equal the reference's (golden from the reference modules, mode 'fused_recurrent',
    train()), and the speaker-state file round-trips."""
    import tempfile
    from lina_speech_amd.initial_state import parse_speaker_state, save_speaker_state, tuning_loss
    from lina_speech_amd.train import Batch
    g = load_golden("lina_d64.npz")
    model = build_lina()
    model.load_state_dict(golden_state_dict(g), strict=True)
    model = model.to(dev).train()
    model.attentive_rnn.to_mode("fused_recurrent")
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    n = sum(1 for k in g.files if k.startswith("ist_k_"))
    params = [(t(f"ist_k_{i}").requires_grad_(True), t(f"ist_v_{i}").requires_grad_(True)) for i in range(n)]
    batch = Batch(t("x"), t("y"), t("encoder_mask"), t("crossatt_mask"), t("logits_mask"))
    loss = tuning_loss(model, batch, params, scale=0.02)
    close(loss, g["ist_loss"], "init-state-tuning loss", 1e-5)
    loss.backward()
    for i, (pk, pv) in enumerate(params):
        close(pk.grad, g[f"ist_gk_{i}"], f"grad state k[{i}]", rel)
        close(pv.grad, g[f"ist_gv_{i}"], f"grad state v[{i}]", rel)
    with tempfile.TemporaryDirectory() as d:
        path = d + "/spk.safetensors"
        save_speaker_state(params, path)
        back = parse_speaker_state(path)
        assert len(back) == n and all(torch.equal(a.cpu(), b[0].detach().cpu()) and torch.equal(c.cpu(), b[1].detach().cpu())
                                      for (a, c), b in zip(back, params))


def _vocoder_from_golden(g, dev):
    from lina_speech_amd.vocoder import WavTokenizerDecoder
    C_in, dim, inter, layers, n_fft, hop = (int(v) for v in g["cfg"])
    voc = WavTokenizerDecoder(n_codes=50, n_codebooks=1, codebook_dim=C_in, dim=dim, intermediate_dim=inter,
                              num_layers=layers, adanorm_num_embeddings=4, n_fft=n_fft, hop_length=hop)
    sd = {k: v for k, v in golden_state_dict(g).items()}
    missing, unexpected = voc.load_state_dict(sd, strict=False)
    assert missing == ["codebook"] and not unexpected, (missing, unexpected)
    return voc.to(dev).eval()


def check_vocoder_golden(dev):
    """f-3: backbone output and waveform equal the reference's VocosBackbone + ISTFTHead (golden from the reference
    modules, tests/golden/make_golden.py::golden_vocoder) -- 2e-4 of max|golden| (fp32) -- and the fp64 oracle
    restatement equals the same golden at 1e-5 (this is what pins the oracle)."""
    from oracle.vocoder_oracle import OracleVocoder
    g = load_golden("vocoder_small.npz")
    voc = _vocoder_from_golden(g, dev)
    feats, bw = torch.from_numpy(g["feats"]).to(dev), torch.from_numpy(g["bw"]).to(dev)
    with torch.no_grad():
        hid = voc.backbone(feats, bandwidth_id=bw)
        audio = voc.head(hid)
    close(hid, g["hidden"], "vocoder backbone output")
    close(audio, g["audio"], "vocoder waveform")
    C_in, dim, inter, layers, n_fft, hop = (int(v) for v in g["cfg"])
    orc = OracleVocoder(golden_state_dict(g), layers, n_fft, hop)
    close(orc.decode(torch.from_numpy(g["feats"]), torch.from_numpy(g["bw"])), g["audio"], "oracle waveform", 1e-5)
    # codes -> features -> audio end to end: the gather is K6a
    codes = torch.randint(0, 50, (1, 3, 23), generator=torch.Generator().manual_seed(4)).to(dev)
    with torch.no_grad():
        a2 = voc(codes, bandwidth_id=bw)
        f2 = voc.codebook[0][codes[0]].transpose(1, 2)
        close(a2, voc.decode(f2, bandwidth_id=bw).cpu().numpy(), "codes -> waveform", 1e-6)
    assert a2.shape == (3, 23 * hop)


def check_simple_gla_golden(dev, full=True):
    """a-12 / BASELINE configs[0]: the product's AttentiveSimpleGLA (scalar-gate GLA on K2) reproduces the output and
    attention weights the REFERENCE's AttentiveSimpleGLA.forward (model/simple_gla.py:152-165) produced on the CPU
    with the pure-PyTorch recurrent layer (golden_simple_gla in tests/golden/make_golden.py): d=256, 2 GLA blocks +
    pos_net, B=4, T=256 (``full``) and a ragged short case T=37, Ttxt=11.  fp32; 2e-4 of max|golden|."""
    from lina_speech_amd.simple_gla import AttentiveSimpleGLA
    g = load_golden("simple_gla_d256.npz")
    rnn = AttentiveSimpleGLA(d_model=256, n_layer=1, heads=4, blind=True, use_short_conv=True)
    sd = {k[4:]: torch.from_numpy(g[k].astype(np.float32)) for k in g.files if k.startswith("sd::")}
    rnn.load_state_dict(sd, strict=True)                     # the reference module's own keys
    rnn = rnn.to(dev).eval()
    x = torch.from_numpy(g["x"].astype(np.float32)).to(dev)
    ctx = torch.from_numpy(g["ctx"].astype(np.float32)).to(dev)
    with torch.no_grad():
        y, att = rnn(x[:, :37], ctx[:, :11])
        close(y, g["y_short"], "simple-GLA short y")
        close(att, g["att_short"], "simple-GLA short att")
        if full:
            y, att = rnn(x, ctx)
            close(y, g["y"], "simple-GLA y (B=4, T=256, d=256)")
            close(att, g["att"].astype(np.float32), "simple-GLA att", 1e-3)      # stored as fp16


def peak_logits(model, alpha: float = 16.0, seed: int = 11, weak: float = 0.07):
    """Random-init weights give nearly flat logits: a third of the positions of a greedy decode are near-ties, where "same
    arg-max as the oracle" says nothing.  This adds a seeded successor structure to the codec head -- every input token i gets
    a successor p(i) (a code token, never a special) and the head row of p(i) receives ``strength_i * E_i / d`` (E = the input
    embedding table, strength_i = alpha * U[0.5, 1.5)) on top of its random initialisation -- so that the logits are PEAKED the
    way a trained model's are: the residual stream carries the current token's embedding, the successor's logit stands out by a
    margin that varies with the token and with what the 13 GLA blocks and the cross-attention add to the stream.  A fraction
    ``weak`` of the tokens gets NO successor: there the flat initialisation decides, so the rows of a batch (which all start
    from BOS) leave the common chain at different places and decode different sequences.  Calibrated with the fp32 CPU oracle
    (L169, B=16, 32 free-running steps): alpha = 16, weak = 0.07 -> median top-2 margin 0.45 of max|logit|, 2.7 % of the
    positions below 0.008 (3.9 % below 0.016), 8 distinct rows, 163 distinct tokens; weak = 0 -> no position below 0.1 but every
    row decodes the same chain; alpha = 0 (plain init) -> 25 % of the positions below 0.016.  The model's other parameters are
    untouched.  In place."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        E = model.rvq_embed.weight[0]                       # [n_in, d]
        W = model.logits_head.weight[0]                     # [n_out, d]
        n, d = E.shape
        assert W.shape[0] == n, "peak_logits: input and output vocabularies must have the same size"
        succ = torch.randperm(n - 3, generator=g) + 3
        succ = torch.cat([succ[:3], succ])                  # the three specials get successors too (BOS starts the chain)
        strength = alpha * (0.5 + torch.rand(n, generator=g))
        strength[3:][torch.rand(n - 3, generator=g) < weak] = 0.0
        W.index_add_(0, succ, (strength[:, None] * E.float() / d).to(W.dtype))
    return model


def reseed_parameters(model, seed: int = 0):
    """Overwrite EVERY parameter of ``model`` with values drawn from a generator keyed by (seed, parameter name) -- the same
    numbers whichever library built the module and in whatever order its initialisers consumed the global RNG.  This is how a
    golden made from the REFERENCE's modules (tests/golden/make_golden.py) and the product's modules get identical weights
    without storing 140 MB of them: same state-dict keys (the drop-in contract) -> same parameters.  Scales by shape: matrices
    0.5 / sqrt(fan_in), embedding tables N(0, 1), depthwise conv taps 0.3, norm weights 1 + 0.1 N, other vectors 0.02 N."""
    import zlib
    with torch.no_grad():
        for name, p in sorted(model.named_parameters()):
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
            r = torch.randn(p.shape, generator=g, dtype=torch.float32)
            if "embed" in name and p.dim() >= 2:
                v = r
            elif p.dim() >= 2 and p.shape[-1] <= 8 and p.shape[-2] == 1:          # depthwise conv [D, 1, W]
                v = 0.3 * r
            elif p.dim() >= 2:
                v = r * (0.5 / (p.shape[-1] ** 0.5))
            elif name.endswith("weight"):                                          # norm gains
                v = 1.0 + 0.1 * r
            else:
                v = 0.02 * r
            p.copy_(v.to(p.dtype))
    return model


def grad_group(name: str) -> str:
    """Parameter groups of the config-5 slice check, by what differentiates them."""
    if name.startswith("txt_encoder") or (".cross_att." in name and ".pos_net." not in name):
        return "softmax attention side (text encoder, cross-attention q/k/v + norms: torch SDPA)"
    if ".tmix." in name:
        return "GLA mixers (K2/K2b, K3/K3b, K5/K5b, gate)"
    if ".cmix." in name:
        return "channel mixers (K11)"
    if "embed" in name or "logits_head" in name:
        return "embeddings + head (K14)"
    return "block norms (K10)"


def grad_errors_by_group(named_grads, g, zero_tol):
    """Worst per-group errors of gradients against the golden digests ``g``: {group: {max, l2, norm, tensor, n}}, and the number
    of analytically-zero tensors (golden max < 1e-5 of the largest: both sides must be noise, ``zero_tol`` x largest)."""
    names = [k[6:] for k in g.files if k.startswith("gsam::")]
    largest = max(float(g["gmax::" + n]) for n in names)
    worst, n_zero = {}, 0
    for name in names:
        grad = named_grads[name]
        assert grad is not None, f"no gradient for {name}"
        nrm, mx, sam = grad_digest(grad)
        ref_sam, ref_mx, ref_nrm = torch.from_numpy(g["gsam::" + name]), float(g["gmax::" + name]), float(g["gnorm::" + name])
        if ref_mx < 1e-5 * largest:
            # analytically zero (e.g. the bias of the keys' LayerNorm: a shift of every key moves all scores of a row
            # alike and the softmax does not see it): the golden holds rounding noise -- ours must be noise too
            assert mx < zero_tol * largest, (name, mx, largest)
            n_zero += 1
            continue
        es = float((sam - ref_sam).abs().max()) / ref_mx
        el = float((sam - ref_sam).norm() / ref_sam.norm())
        en = abs(nrm - ref_nrm) / ref_nrm
        w = worst.setdefault(grad_group(name), {"max": 0.0, "l2": 0.0, "norm": 0.0, "tensor": None, "n": 0})
        w["n"] += 1
        if es > w["max"]:
            w["max"], w["tensor"] = es, name
        w["l2"], w["norm"] = max(w["l2"], el), max(w["norm"], en)
    return worst, n_zero


def structured_targets(B, n, vocab, seed=29, follow=0.9):
    """Codec-token targets with a bigram structure: y[0] = BOS (1), then y[t+1] = succ[y[t]] with probability ``follow``
    (succ = a seeded permutation of the code tokens 3..vocab-1), a random code token otherwise.  [B, n, 1] int64."""
    g = torch.Generator().manual_seed(seed)
    succ = torch.randperm(vocab - 3, generator=g) + 3
    succ = torch.cat([succ[:3], succ])
    y = torch.empty(B, n, dtype=torch.long)
    y[:, 0] = 1
    rnd = torch.randint(3, vocab, (B, n), generator=g)
    keep = torch.rand(B, n, generator=g) < follow
    for t in range(1, n):
        y[:, t] = torch.where(keep[:, t], succ[y[:, t - 1]], rnd[:, t])
    return y.unsqueeze(-1)


def grad_digest(t, n: int = 256):
    """What a golden keeps of one gradient tensor: its L2 norm, its max |.| and ``n`` entries at a fixed stride."""
    f = t.detach().float().flatten().cpu()
    step = max(1, f.numel() // n)
    return f.norm().item(), f.abs().max().item(), f[::step][:n].clone()


def check_config5_slice_golden(dev, dtype=torch.float32, rel_loss=2e-4, rel_grad=5e-3):
    """BASELINE configs[4] at its named sequence length against the REFERENCE's autograd: a slice of L169 (d = 1024, H = 4,
    1 + 1 GLA blocks + the pos_net block, one text-encoder layer, 4099-way head), b = 1, T = 4096: teacher-forced forward in
    train() mode, CE loss, backward.  The golden (tests/golden/l169_slice_T4096.npz, made by make_golden.py from the reference's
    model/*.py with the CPU oracle behind the fla names) holds the loss and a digest (norm, max, 256 strided entries) of EVERY
    parameter gradient in fp32; weights come from reseed_parameters on both sides.  Per tensor: max |sampled - golden| /
    max|golden|, relative L2 error of the samples, relative norm error; worst case per parameter group (recorded).
    fp32: every group within ``rel_grad``.
    bf16 autocast: on this random-init model with random targets the gradients are sums of 4096 cancelling terms, and the
    REFERENCE's own modules under ``torch.autocast(bfloat16)`` deviate from their fp32 gradients by 19-35 % of max|gradient| on
    single entries (11 % on norms) -- the golden holds those figures per group (``bf16ref::*``): ours must be as close to the fp32
    reference as the reference's bf16 run is (x 1.6 + 0.02: two independent draws of the same rounding noise)."""
    from lina_speech_amd.configs import l169
    g = load_golden("l169_slice_T4096.npz")
    torch.manual_seed(0)
    model = reseed_parameters(l169(n_layer=1, txt_layers=1), seed=int(g["seed"])).to(dev).train()
    x, y = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["y"]).to(dev)
    em = torch.ones(x.shape[0], x.shape[1], x.shape[1], dtype=torch.bool, device=dev)
    cm = torch.ones(x.shape[0], y.shape[1], x.shape[1], dtype=torch.bool, device=dev)
    lm = torch.ones(x.shape[0], y.shape[1], dtype=torch.bool, device=dev)
    model.zero_grad()
    if dtype == torch.float32:
        _, loss, _, _, _ = model(x, y, em, cm, logits_mask=lm)
    else:
        with torch.autocast("cuda", dtype=dtype):
            _, loss, _, _, _ = model(x, y, em, cm, logits_mask=lm)
    loss.backward()
    from kernel_cases import record_parity
    tag = f"config-5 slice T=4096 ({str(dtype)[6:]})"
    ref_loss = float(g["loss"])
    e = abs(float(loss.detach()) - ref_loss) / abs(ref_loss)
    record_parity(f"{tag}: loss vs reference autograd", e, rel_loss)
    assert e <= rel_loss, (float(loss), ref_loss)
    worst, n_zero = grad_errors_by_group({n_: p.grad for n_, p in model.named_parameters()}, g,
                                         1e-5 if dtype == torch.float32 else 1e-3)
    assert sum(w["n"] for w in worst.values()) > 40
    groups = [str(s_) for s_ in g["bf16ref_groups"]]
    bad = []
    for grp, w in worst.items():
        if dtype == torch.float32:
            tol_max = tol_norm = rel_grad
        else:
            i = groups.index(grp)
            # (single entries: the group's own figure; norms: the largest group figure -- a norm error is one draw per tensor)
            tol_max, tol_norm = 1.6 * float(g["bf16ref_max"][i]) + 0.02, 1.6 * float(g["bf16ref_norm"].max()) + 0.02
        record_parity(f"{tag}: {grp}: worst |sampled gradient - golden| / max|golden| over {w['n']} tensors", w["max"], tol_max,
                      tensor=w["tensor"], worst_relative_l2=w["l2"], worst_norm_error=w["norm"], norm_tolerance=tol_norm)
        if w["max"] > tol_max or w["norm"] > tol_norm:
            bad.append((grp, w, tol_max, tol_norm))
    record_parity(f"{tag}: analytically zero gradient tensors (noise on both sides)", n_zero, None)
    assert not bad, bad
