#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ -- runs ONLY in the build container.

It imports the reference's in-tree host modules from /root/reference (model/gla.py,
model/modeling_lina.py, model/crossatt.py, model/encoder.py, model/base_blocks.py,
model/multiembed.py, model/tools.py) with the CPU oracle bound to the absent
``fla.*`` names (oracle/fla_standin.py), runs them on seeded synthetic inputs and
stores inputs + weights + outputs as .npz.  Only DATA is stored; no reference source
travels.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import fla_standin  # noqa: E402

fla_standin.install()

from model.gla import AttentiveGLA, GatedLinearAttention  # noqa: E402  (reference)
from model.encoder import TextEncoder  # noqa: E402
from model.modeling_lina import LinaModel  # noqa: E402
from model import tools as ref_tools  # noqa: E402


def npz(path, **kw):
    out = {}
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez(os.path.join(HERE, path), **out)
    print("wrote", path, {k: tuple(v.shape) for k, v in out.items() if v.ndim} and
          sum(v.nbytes for v in out.values()) // 1024, "KiB")


POS_ROWS = 32  # ConvPos holds a 2000-row table (crossatt.py:22-24); only the rows a test can touch are stored


def sd_arrays(model, prefix="sd::"):
    out = {}
    for k, v in model.state_dict().items():
        if k.endswith("pos_embed.embed.weight"):
            out[prefix + k + "::first_rows"] = v[:POS_ROWS]
            out[prefix + k + "::full_rows"] = torch.tensor(v.shape[0])
        else:
            out[prefix + k] = v
    return out


def golden_mixer():
    """One GatedLinearAttention (model/gla.py:44-247): prefill through every mode, then
    cached prefill + 3 single-token steps (cache layout, model/gla.py:229-240)."""
    torch.manual_seed(0)
    cfg = dict(hidden_size=128, num_heads=2, expand_k=1.0, expand_v=2.0, use_short_conv=True, layer_idx=0)
    m = GatedLinearAttention(mode="fused_chunk", **cfg).eval()
    B, T = 3, 21
    x = torch.randn(B, T + 3, 128)
    out = {"x": x}
    with torch.no_grad():
        for mode in ("fused_chunk", "chunk", "fused_recurrent"):
            m.mode = mode
            out["o_" + mode] = m(x[:, :T])
        # reset mask (packing): gate -20 at two positions (model/gla.py:182-183)
        reset = torch.zeros(B, T, dtype=torch.bool)
        reset[0, 7] = True
        reset[2, 13] = True
        m.mode = "fused_chunk"
        out["reset_mask"] = reset
        out["o_reset"] = m(x[:, :T], reset_mask=reset)
        # cached prefill + steps
        cache = fla_standin.Cache()
        cache.update(m.init_state(B), 0, offset=0)
        m.mode = "fused_recurrent"
        out["o_prefill_cached"] = m(x[:, :T], past_key_values=cache, use_cache=True)
        for i in range(3):
            out[f"o_step{i}"] = m(x[:, T + i:T + i + 1], past_key_values=cache, use_cache=True)
        for j, s in enumerate(cache.states[0]):
            out[f"cache_after_{j}"] = s.clone()
    out.update(sd_arrays(m))
    npz("mixer_d128.npz", **out)


def build_lina(d=64, n_layer=1, heads=1, ev=1.0, n_codebook=253, txt_layers=1):
    rnn = AttentiveGLA(d_model=d, n_layer=n_layer, heads=heads, blind=True, use_short_conv=True,
                       expand_k=1.0, expand_v=ev, pos_type="convolutional")
    txt = TextEncoder(d, heads, n_layers=txt_layers, dropout=0.0, rotary=False)
    return LinaModel(rnn, d_model=d, n_quant=1, n_codebook=n_codebook, n_special_token_in=3,
                     n_special_token_out=3, n_txt_vocab=256, txt_encoder=txt)


def golden_lina():
    """LinaModel.forward (modeling_lina.py:61-108) and generate_batch (:111-192), greedy."""
    torch.manual_seed(0)
    model = build_lina().eval()
    B, Ttxt, n = 3, 9, 14
    x = torch.randint(3, 256, (B, Ttxt))
    y = torch.randint(3, 256, (B, n, 1))
    y[:, 0] = 1
    txt_len = torch.tensor([9, 6, 8])
    enc_mask = (torch.arange(Ttxt)[None, :] < txt_len[:, None])
    encoder_mask = enc_mask[:, None, :] & enc_mask[:, :, None]
    crossatt_mask = enc_mask[:, None, :].expand(B, n, Ttxt).contiguous()
    logits_mask = torch.ones(B, n, dtype=torch.bool)
    logits_mask[1, 10:] = False
    out = dict(x=x, y=y, encoder_mask=encoder_mask, crossatt_mask=crossatt_mask, logits_mask=logits_mask)
    with torch.no_grad():
        logits, loss, att, _, _ = model(x, y, encoder_mask, crossatt_mask, logits_mask=logits_mask)
        out.update(fwd_logits=logits, fwd_loss=loss, fwd_att=att)
        # decode: one text repeated over the batch (modeling_lina.py:125)
        xg = x[0]
        qs, atts, stop_tokens, cuts = model.generate_batch(xg, batch_size=B, max_seqlen=12, k=1,
                                                           first_greedy_quant=0, force_max_seqlen=True)
        out.update(gen_x=xg, gen_qs=qs, gen_atts=atts, gen_stop_tokens=stop_tokens)
        out["gen_cut_lens"] = torch.tensor([c[0].shape[-1] for c in cuts])
        # with a codec prompt (prompt-forcing branch, modeling_lina.py:134-142,175-176)
        prompt = torch.randint(0, 253, (1, 1, 4))
        qs2, atts2, st2, _ = model.generate_batch(xg, batch_size=B, prompt=prompt, max_seqlen=9, k=1,
                                                  first_greedy_quant=0, force_max_seqlen=True)
        out.update(gen_prompt=prompt, gen_prompt_qs=qs2, gen_prompt_atts=atts2, gen_prompt_stop=st2)
        # per-step logits of a teacher-forced step loop + final cache (AttentiveGLA.step, gla.py:358-365)
        x_enc = model.txt_encoder(model.txt_embed(x))
        state = model.attentive_rnn.init_state(batch_size=B)
        from einops import reduce, rearrange
        y_embd = reduce(model.rvq_embed(rearrange(y, "b n q -> q b n")), "q b n d -> b n d", "sum")
        step_logits = []
        for t in range(n - 1):
            h, a, state = model.attentive_rnn.step(y_embd[:, t:t + 1], x_enc, t, state)
            step_logits.append(model.logits_head(h))
        out["step_logits"] = torch.cat(step_logits, 1)
        for li, st in enumerate(state.states):
            for j, s in enumerate(st):
                out[f"cache_{li}_{j}"] = s
    # training step: loss and parameter gradients of the teacher-forced forward in train() mode
    # (train_lina.py:72-94; all dropouts are 0 so the step is deterministic)
    model.train()
    model.zero_grad()
    _, tloss, _, _, _ = model(x, y, encoder_mask, crossatt_mask, logits_mask=logits_mask)
    tloss.backward()
    out["train_loss"] = tloss.detach()
    for name, p in model.named_parameters():
        if p.grad is not None:
            out["grad::" + name] = p.grad.detach().clone()
    # initial-state tuning (initial_state.py:85-160): rank-1 state parameters, mode 'fused_recurrent', loss and the
    # gradients w.r.t. the state parameters only
    torch.manual_seed(5)
    model.attentive_rnn.to_mode("fused_recurrent")
    params = model.attentive_rnn.get_init_state_tuning_params(lora=1, device="cpu")
    model.zero_grad()
    init_state = model.attentive_rnn.get_state_from_params(params, B, scale=0.02)
    _, iloss, _, _, _ = model(x, y, encoder_mask, crossatt_mask, logits_mask=logits_mask, init_state=init_state)
    iloss.backward()
    out["ist_loss"] = iloss.detach()
    for i, (pk, pv) in enumerate(params):
        out[f"ist_k_{i}"], out[f"ist_v_{i}"] = pk.detach().clone(), pv.detach().clone()
        out[f"ist_gk_{i}"], out[f"ist_gv_{i}"] = pk.grad.detach().clone(), pv.grad.detach().clone()
    model.attentive_rnn.to_mode("fused_chunk")
    model.eval()
    out.update(sd_arrays(model))
    npz("lina_d64.npz", **out)


def golden_config5_slice():
    """BASELINE configs[4] at its named length: a slice of L169 (d=1024, H=4, n_layer=1 -> 3 GLA blocks, 1 text layer) through
    the REFERENCE's modules (train() mode, default chunk mode, the CPU oracle behind fla.*), b=1, T=4096: loss + a digest of
    every parameter gradient.  Weights: tests/model_cases.reseed_parameters (keyed by parameter name), not stored."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from model_cases import reseed_parameters, grad_digest
    seed = 3
    torch.manual_seed(0)
    model = reseed_parameters(build_lina(d=1024, n_layer=1, heads=4, n_codebook=4096, txt_layers=1), seed=seed).train()
    B, Ttxt, n = 1, 64, 4097
    g = torch.Generator().manual_seed(17)
    x = torch.randint(3, 256, (B, Ttxt), generator=g)
    y = torch.randint(3, 4099, (B, n, 1), generator=g)
    y[:, 0] = 1
    em = torch.ones(B, Ttxt, Ttxt, dtype=torch.bool)
    cm = torch.ones(B, n, Ttxt, dtype=torch.bool)
    lm = torch.ones(B, n, dtype=torch.bool)
    model.zero_grad()
    _, loss, _, _, _ = model(x, y, em, cm, logits_mask=lm)
    loss.backward()
    out = dict(x=x, y=y, loss=loss.detach(), seed=torch.tensor(seed))
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        nrm, mx, sam = grad_digest(p.grad)
        out["gnorm::" + name], out["gmax::" + name], out["gsam::" + name] = torch.tensor(nrm), torch.tensor(mx), sam
    # the yardstick of the bf16 test: how far the REFERENCE's own bf16-autocast gradients are from its fp32 gradients
    from model_cases import grad_errors_by_group

    class _G(dict):
        files = property(lambda self: list(self.keys()))
    gg = _G({k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in out.items()})
    model.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, loss16, _, _, _ = model(x, y, em, cm, logits_mask=lm)
    loss16.backward()
    worst, _ = grad_errors_by_group({n_: p.grad for n_, p in model.named_parameters()}, gg, 1e-3)
    groups = sorted(worst)
    out["bf16ref_groups"] = np.array(groups)
    out["bf16ref_max"] = np.array([worst[k]["max"] for k in groups], dtype=np.float32)
    out["bf16ref_l2"] = np.array([worst[k]["l2"] for k in groups], dtype=np.float32)
    out["bf16ref_norm"] = np.array([worst[k]["norm"] for k in groups], dtype=np.float32)
    out["bf16ref_loss"] = loss16.detach().float()
    print({k: (round(worst[k]["max"], 3), round(worst[k]["norm"], 3)) for k in groups})
    npz("l169_slice_T4096.npz", **out)


def golden_config5_structured():
    """The config-5 slice again (d=1024, H=4, 3 GLA blocks, T=4096, b=1) on a WELL-CONDITIONED problem: the targets follow a
    seeded successor chain (y[t+1] = succ[y[t]] for 90 % of the positions, random otherwise) -- a bigram structure every
    block's gradient sees coherently, instead of 4096 random targets whose contributions cancel (on those the reference's own
    bf16-autocast gradients are 19-35 % off its fp32 ones on single entries, so a bf16 check against them cannot bite).
    Stored: loss, and per parameter gradient its norm, max and 4096 strided entries (fp32 reference autograd); plus, as the
    yardstick, the cosine / norm error of the REFERENCE's own bf16-autocast gradients against them."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from model_cases import reseed_parameters, grad_digest, structured_targets
    seed = 5
    torch.manual_seed(0)
    model = reseed_parameters(build_lina(d=1024, n_layer=1, heads=4, n_codebook=4096, txt_layers=1), seed=seed).train()
    B, Ttxt, n = 1, 64, 4097
    g = torch.Generator().manual_seed(23)
    x = torch.randint(3, 256, (B, Ttxt), generator=g)
    y = structured_targets(B, n, 4099, seed=29)
    em = torch.ones(B, Ttxt, Ttxt, dtype=torch.bool)
    cm = torch.ones(B, n, Ttxt, dtype=torch.bool)
    lm = torch.ones(B, n, dtype=torch.bool)
    model.zero_grad()
    _, loss, _, _, _ = model(x, y, em, cm, logits_mask=lm)
    loss.backward()
    out = dict(x=x, y=y, loss=loss.detach(), seed=torch.tensor(seed))
    ref = {}
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        nrm, mx, sam = grad_digest(p.grad, n=4096)
        out["gnorm::" + name], out["gmax::" + name], out["gsam::" + name] = torch.tensor(nrm), torch.tensor(mx), sam
        ref[name] = p.grad.detach().clone()
    model.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, loss16, _, _, _ = model(x, y, em, cm, logits_mask=lm)
    loss16.backward()
    names, cos, nerr = [], [], []
    for name, p in model.named_parameters():
        if name not in ref:
            continue
        a, b = p.grad.detach().float().flatten(), ref[name].float().flatten()
        names.append(name)
        cos.append(float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30)))
        nerr.append(float((a.norm() - b.norm()).abs() / b.norm().clamp_min(1e-30)))
    out["bf16ref_names"] = np.array(names)
    out["bf16ref_cos"] = np.array(cos, dtype=np.float32)
    out["bf16ref_norm_err"] = np.array(nerr, dtype=np.float32)
    out["bf16ref_loss"] = loss16.detach().float()
    order = np.argsort(cos)
    print("loss", float(loss), "bf16", float(loss16))
    print("lowest cosines:", [(names[i], round(cos[i], 4), round(nerr[i], 4)) for i in order[:12]])
    print("median cosine", float(np.median(cos)), "max norm err", max(nerr))
    npz("l169_slice_T4096_structured.npz", **out)


def golden_tools():
    """Known answers of model/tools.py helpers."""
    torch.manual_seed(0)
    code = torch.tensor([[10, 11, 12, 13]]) + 3
    d = ref_tools.delay_rvq(code, head_token=1, tail_token=2)
    code2 = torch.randint(3, 100, (3, 7))
    d2 = ref_tools.delay_rvq(code2, head_token=1, tail_token=2)
    und = ref_tools.undelay_rvq(d2.unsqueeze(1))
    logits = torch.randn(5, 4099)
    k1 = ref_tools.topk_sampling(logits.clone(), k=1)
    pm = ref_tools.packmask_2d([2, 1], [3, 2])
    sm = ref_tools.sequence_mask(torch.tensor([3, 1, 4]), device="cpu")
    npz("tools.npz", delay_in=code, delay_out=d, delay2_in=code2, delay2_out=d2, undelay2=und,
        topk_logits=logits, topk_k1=k1, packmask=pm, seqmask=sm)


def golden_vocoder():
    """WavTokenizer decode: VocosBackbone (3rdparty/decoder/models.py:152-235) + ISTFTHead (heads.py:24-67) of the
    reference with small seeded weights; torchaudio is only an import-time name of heads.py (unused by ISTFTHead)."""
    import types
    ta = types.ModuleType("torchaudio"); taf = types.ModuleType("torchaudio.functional")
    taff = types.ModuleType("torchaudio.functional.functional")
    taff._hz_to_mel = taff._mel_to_hz = None
    sys.modules.update({"torchaudio": ta, "torchaudio.functional": taf, "torchaudio.functional.functional": taff})
    sys.path.insert(0, os.path.join(REF, "3rdparty"))
    from decoder.models import VocosBackbone
    from decoder.heads import ISTFTHead
    torch.manual_seed(11)
    C_in, dim, inter, layers, n_fft, hop = 32, 64, 128, 2, 64, 16
    backbone = VocosBackbone(input_channels=C_in, dim=dim, intermediate_dim=inter, num_layers=layers,
                             adanorm_num_embeddings=4).eval()
    head = ISTFTHead(dim=dim, n_fft=n_fft, hop_length=hop, padding="same").eval()
    with torch.no_grad():                       # the reference initialises most of these to constants: randomise
        for name, p in list(backbone.named_parameters()) + list(head.named_parameters()):
            if p.dim() == 1 or "scale" in name or "shift" in name:
                p.add_(torch.randn_like(p) * 0.2)
            else:
                p.mul_(8.0)
        head.out.weight.mul_(0.3)
        B, L = 3, 23
        feats = torch.randn(B, C_in, L)
        bw = torch.tensor([2])                  # one shared bandwidth id (the only form the reference's AdaLayerNorm broadcasts)
        hid = backbone(feats, bandwidth_id=bw)
        audio = head(hid)
    out = dict(feats=feats, bw=bw, hidden=hid, audio=audio,
               cfg=torch.tensor([C_in, dim, inter, layers, n_fft, hop]))
    for k, v in backbone.state_dict().items():
        out["sd::backbone." + k] = v
    for k, v in head.state_dict().items():
        out["sd::head." + k] = v
    npz("vocoder_small.npz", **out)


def golden_simple_gla():
    """BASELINE.json configs[0] / SURVEY 8(a) a-12: the reference's AttentiveSimpleGLA.forward (model/simple_gla.py:
    152-165) at d=256, 2 GLA blocks (+ the pos_net block), B=4, T=256 on the CPU -- through the reference's own wrapper,
    MixingBlock and BlindCrossAttention, with the oracle's scalar-gate layer bound to fla.layers.simple_gla.  Weights
    are rounded to fp16-representable values BEFORE the run and stored as fp16 (halves the fixture)."""
    from model.simple_gla import AttentiveSimpleGLA  # noqa: E402  (reference)
    torch.manual_seed(3)
    rnn = AttentiveSimpleGLA(d_model=256, n_layer=1, heads=4, blind=True, use_short_conv=True).eval()
    with torch.no_grad():
        for p_ in rnn.parameters():
            p_.copy_(p_.half().float())
    B, T, Ttxt = 4, 256, 24
    x = torch.randn(B, T, 256).half().float()
    ctx = torch.randn(B, Ttxt, 256).half().float()
    with torch.no_grad():
        y, att = rnn(x, ctx)
        y_short, att_short = rnn(x[:, :37], ctx[:, :11])
    out = {"x": x.half(), "ctx": ctx.half(), "y": y, "att": att.half(), "y_short": y_short, "att_short": att_short}
    for k, v in rnn.state_dict().items():
        out["sd::" + k] = v.half()
    npz("simple_gla_d256.npz", **out)


if __name__ == "__main__":
    import argparse
    only = sys.argv[1:]
    todo = {"vocoder": golden_vocoder, "tools": golden_tools, "mixer": golden_mixer, "lina": golden_lina,
            "simple_gla": golden_simple_gla, "config5_slice": golden_config5_slice,
            "config5_structured": golden_config5_structured}
    for name, fn in todo.items():
        if not only or name in only:
            fn()
