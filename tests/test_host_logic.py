"""Host-side behaviours that mirror reference glue (CPU, emulator backend where a kernel is involved)."""
import math
import os

import pytest
import torch

from model_cases import build_lina


def test_rotary_matches_closed_form_rotation():
    """RotaryEmbedding / apply_rotary_emb (reference model/base_blocks.py:15,29-35 via rotary_embedding_torch):
    channel pair (2i, 2i+1) of the first `dim` channels is rotated by angle pos * theta^(-2i/dim); the rest of the
    head is untouched; the parameter is `freqs` (state-dict key `rotary.freqs`)."""
    from lina_speech_amd.blocks import RotaryEmbedding, SelfAttention, TextEncoder
    torch.manual_seed(0)
    dim, n = 8, 5
    rot = RotaryEmbedding(dim)
    assert list(rot.state_dict()) == ["freqs"] and not rot.freqs.requires_grad
    t = torch.randn(2, 3, n, 16, dtype=torch.float64)
    got = rot.rotate_queries_or_keys(t, offset=2)
    ref = t.clone()
    for p in range(n):
        for i in range(dim // 2):
            a = (p + 2) * 10000.0 ** (-2 * i / dim)
            x0, x1 = t[..., p, 2 * i], t[..., p, 2 * i + 1]
            ref[..., p, 2 * i] = x0 * math.cos(a) - x1 * math.sin(a)
            ref[..., p, 2 * i + 1] = x1 * math.cos(a) + x0 * math.sin(a)
    assert (got - ref).abs().max() < 1e-6
    assert torch.equal(got[..., dim:], t[..., dim:])
    # relative-position property: <rot(q,p1), rot(k,p2)> depends on p1 - p2 only
    q, k = torch.randn(1, 1, 1, 16, dtype=torch.float64), torch.randn(1, 1, 1, 16, dtype=torch.float64)
    d1 = (rot.rotate_queries_or_keys(q, 7) * rot.rotate_queries_or_keys(k, 3)).sum()
    d2 = (rot.rotate_queries_or_keys(q, 14) * rot.rotate_queries_or_keys(k, 10)).sum()
    assert abs(d1 - d2) < 1e-6          # freqs are fp32
    # default construction = the reference's (rotary on), checkpoint key present, forward runs with and without pos
    enc = TextEncoder(32, 2, n_layers=1, dropout=0.0)
    assert "sa.0.tmix.rotary.freqs" in enc.state_dict()
    x = torch.randn(2, 6, 32)
    y0 = enc(x)
    y1 = enc(x, pos=torch.arange(6)[None, :].expand(2, -1))
    assert (y0 - y1).abs().max() < 1e-5                     # explicit positions 0..n-1 == implicit ones
    assert isinstance(enc.sa[0].tmix, SelfAttention)


def test_multiembedding_padding_row_gets_no_gradient(emu):
    """reference model/multiembed.py:21-23: F.embedding(padding_idx=0) under vmap -> no gradient for row 0, in both
    the per-level forward (training) and the fused gather+sum."""
    from lina_speech_amd.codec import MultiEmbedding
    torch.manual_seed(0)
    emb = MultiEmbedding(2, 7, 8, padding_idx=0)
    idx = torch.tensor([[[0, 3, 0, 5]], [[2, 0, 0, 1]]])
    emb(idx).sum().backward()
    g1 = emb.weight.grad.clone()
    emb.weight.grad = None
    emb.embed_sum(idx).sum().backward()
    g2 = emb.weight.grad.clone()
    for g in (g1, g2):
        assert float(g[:, 0].abs().max()) == 0.0
        assert float(g[0, 3].abs().min()) == 1.0 and float(g[1, 2].abs().min()) == 1.0
    assert torch.equal(g1, g2)
    # the forward still gathers the (non-zero) padding row
    assert float(emb.embed_sum(idx)[0, 0].abs().max()) > 0


def test_short_conv_accepts_a_cache_of_another_dtype(emu):
    """An fp32 cache (GatedLinearAttention.init_state with fp32 master weights) with bf16 activations (autocast):
    the reference's cache.copy_(...) casts; so does ops.short_conv."""
    from lina_speech_amd import ops
    from oracle import gla_oracle as O
    torch.manual_seed(0)
    B, T, D, W = 2, 6, 16, 4
    x = torch.randn(B, T, D).to(torch.bfloat16)
    w = torch.randn(D, 1, W) * 0.5
    cache = torch.zeros(B, D, W)
    y = ops.short_conv(x, w, cache=cache)
    ref_cache = torch.zeros(B, D, W, dtype=torch.float64)
    ref = O.short_conv(x.double(), w.to(torch.bfloat16).double(), cache=ref_cache)
    assert (y.double() - ref).abs().max() < 3e-2
    assert cache.dtype == torch.float32 and (cache.double() - ref_cache).abs().max() < 1e-6
    x1 = torch.randn(B, 1, D).to(torch.bfloat16)
    y1 = ops.short_conv(x1, w, cache=cache)                      # step form reads AND writes the cache
    ref1 = O.short_conv(x1.double(), w.to(torch.bfloat16).double(), cache=ref_cache)
    assert (y1.double() - ref1).abs().max() < 3e-2
    assert (cache.double() - ref_cache).abs().max() < 1e-6


def test_grad_ckpt_switch_gives_identical_loss_and_gradients(emu, monkeypatch):
    """GRAD_CKPT (reference model/gla.py:26-33,290-291,297-298): blocks recomputed in backward, same numbers."""
    from lina_speech_amd.train import synthetic_batch
    torch.manual_seed(0)
    model = build_lina(d=64, n_layer=1).train()
    batch = synthetic_batch(b=2, n=20, t_txt=9, n_codebook=253, seed=5)

    def run():
        model.zero_grad(set_to_none=True)
        loss = model(batch.x, batch.y, batch.encoder_mask, batch.crossatt_mask, logits_mask=batch.logits_mask)[1]
        loss.backward()
        return loss.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    monkeypatch.delenv("GRAD_CKPT", raising=False)
    l0, g0 = run()
    monkeypatch.setenv("GRAD_CKPT", "1")
    calls = []
    import torch.utils.checkpoint as ck
    orig = ck.checkpoint
    monkeypatch.setattr(ck, "checkpoint", lambda *a, **kw: (calls.append(1), orig(*a, **kw))[1])
    l1, g1 = run()
    assert len(calls) == 2                                     # one encoder + one decoder block
    assert torch.equal(l0, l1)
    assert set(g0) == set(g1)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n


def test_training_forward_takes_the_fused_qkv_convolution(emu):
    """mixer.py: with gradients, no cache and the stacked projection, the three depthwise convolutions run as ONE
    short_conv3 node per block (autograd._ShortConv3Function) and the projection's gradient slab needs no copy for them."""
    from lina_speech_amd import autograd as AG, ops
    from lina_speech_amd.train import synthetic_batch
    torch.manual_seed(0)
    model = build_lina(d=64, n_layer=1).train()
    batch = synthetic_batch(b=2, n=20, t_txt=9, n_codebook=253, seed=5)
    calls, orig = [], AG._ShortConv3Function.apply
    AG._ShortConv3Function.apply = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        loss = model(batch.x, batch.y, batch.encoder_mask, batch.crossatt_mask, logits_mask=batch.logits_mask)[1]
        loss.backward()
    finally:
        AG._ShortConv3Function.apply = orig
    assert len(calls) == 3                                     # encoder block, decoder block, pos_net block
    assert all(p.grad is not None for n, p in model.named_parameters() if "conv1d" in n)


def test_train_step_defaults_follow_the_reference(emu):
    """train_lina.py:25-29,104-118: AdamW 5e-4, betas (0.9, 0.999), wd 0.1, cosine schedule with 500 warm-up steps."""
    from lina_speech_amd.train import TrainStep
    ts = TrainStep(build_lina(), autocast_dtype=None, ddp=False)
    g = ts.opt.param_groups[0]
    assert g["betas"] == (0.9, 0.999) and g["weight_decay"] == 0.1 and g["initial_lr"] == 5e-4
    assert g["lr"] == 0.0 and ts.grad_clip is None
    lam = ts.sched.lr_lambdas[0]
    assert lam(250) == 0.5 and lam(500) == 1.0 and abs(lam(150250) - 0.5) < 1e-9 and lam(300000) == 0.0


def test_train_initial_state_restores_requires_grad(emu):
    from lina_speech_amd.initial_state import train_initial_state
    from lina_speech_amd.train import synthetic_batch
    model = build_lina()
    model.txt_embed.weight.requires_grad_(False)
    before = {n: p.requires_grad for n, p in model.named_parameters()}
    batch = synthetic_batch(b=2, n=12, t_txt=7, n_codebook=253, seed=1)
    train_initial_state(model, iter([batch] * 2), n_steps=2, grad_acc=1, device="cpu")
    assert {n: p.requires_grad for n, p in model.named_parameters()} == before


def test_build_refuses_a_kernel_that_spills_where_the_dma_wait_counts_operations():
    from lina_speech_amd import build
    ok = ("a.hip:9:1: remark: Function Name: k1 [-Rpass-analysis=kernel-resource-usage]\n"
          "a.hip:9:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]\nreal warning\n")
    assert build._check_no_scratch("a.hip", ok) == "real warning"
    bad = ok.replace("ScratchSize [bytes/lane]: 0", "ScratchSize [bytes/lane]: 20")
    with pytest.raises(RuntimeError, match="k1 spills"):
        build._check_no_scratch("a.hip", bad)


def test_isa_audit_flags_an_mfma_under_an_unskipped_exec_mask():
    """tools/isa_mfma_exec.py (the audit behind the round-3 finding: an MFMA issued under EXEC = 0 still executes on
    unwritten operand registers): the scanner must flag an MFMA inside an s_and_saveexec region without s_cbranch_execz,
    accept the skipped form and a scalar branch, and follow masks restored out of order."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "isa_mfma_exec", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "isa_mfma_exec.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = """_Z3badv:
        s_and_saveexec_b64 s[4:5], vcc
        v_mov_b32_e32 v18, v61
        v_mfma_f32_16x16x32_bf16 v[2:5], v[18:21], v[6:9], v[2:5]
        s_or_b64 exec, exec, s[4:5]
        s_endpgm""".splitlines()
    assert mod.scan(bad) == {"_Z3badv": 1}
    good = """_Z4goodv:
        s_and_saveexec_b64 s[4:5], vcc
        s_cbranch_execz .LBB0_2
        v_mfma_f32_16x16x32_bf16 v[2:5], v[18:21], v[6:9], v[2:5]
    .LBB0_2:
        s_or_b64 exec, exec, s[4:5]
        s_cbranch_scc1 .LBB0_4
        v_mfma_f32_16x16x32_bf16 v[2:5], v[18:21], v[6:9], v[2:5]
    .LBB0_4:
        s_and_saveexec_b64 s[4:5], vcc
        s_and_saveexec_b64 s[6:7], vcc
        s_or_b64 exec, exec, s[4:5]
        v_mfma_f32_16x16x32_bf16 v[2:5], v[18:21], v[6:9], v[2:5]
        s_endpgm""".splitlines()
    assert mod.scan(good) == {}
    # control flow: behind an unconditional branch the fall-through is dead (the masked block above jumped back to its loop
    # header, which restores EXEC): the block at the next label is entered by jumps only and starts clean; an MFMA that
    # follows the saveexec in the SAME block is still flagged, and scanning goes on behind an s_endpgm in mid-kernel
    flow = """_Z4flowv:
        s_and_saveexec_b64 s[10:11], s[8:9]
        v_mfma_f32_16x16x32_bf16 v[2:5], v[18:21], v[6:9], v[2:5]
        s_branch .LBB0_9
    .LBB0_5:
        v_mfma_f32_16x16x32_bf16 v[2:5], v[18:21], v[6:9], v[2:5]
        s_endpgm
    .LBB0_9:
        s_and_saveexec_b64 s[10:11], s[8:9]
        v_mfma_f32_16x16x32_bf16 v[2:5], v[18:21], v[6:9], v[2:5]
        s_endpgm""".splitlines()
    assert mod.scan(flow) == {"_Z4flowv": 2}


@pytest.mark.parametrize("rows,n_out,n_in,split", [(4096, 24, 16, None), (16384, 40, 21, None), (16384, 40, 21, 4),
                                                   (300, 8, 8, None)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_matches_f_linear_and_splits_the_weight_gradient(rows, n_out, n_in, split, dtype):
    """ops.linear (the train path's projection): same y, dx, dW, db as F.linear under autograd; the token-split weight
    gradient (CPU form of the batched GEMM) equals the one-GEMM form."""
    from lina_speech_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, rows // 2, n_in, generator=g).to(dtype)
    w = torch.randn(n_out, n_in, generator=g) * 0.2
    b = torch.randn(n_out, generator=g)
    dy = torch.randn(2, rows // 2, n_out, generator=g).to(dtype)
    outs = []
    for fn in (ops.linear, torch.nn.functional.linear):
        xx = x.clone().requires_grad_()
        ww, bb = w.clone().to(dtype).requires_grad_(), b.clone().to(dtype).requires_grad_()
        y = fn(xx, ww, bb)
        (y.float() * dy.float()).sum().backward()
        outs.append((y, xx.grad, ww.grad, bb.grad))
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    for a, r, what in zip(outs[0], outs[1], ("y", "dx", "dW", "db")):
        assert a.dtype == r.dtype and a.shape == r.shape, what
        assert (a.float() - r.float()).abs().max() <= tol * max(1.0, r.float().abs().max().item()), what
    s_auto = ops._linear_split(rows, n_out, n_in)
    assert s_auto == (1 if rows < 4096 else 8 if rows >= 16384 else 2)
    d2, x2 = dy.reshape(-1, n_out), x.reshape(-1, n_in)
    one = ops.linear_weight_grad(d2, x2, split=1)
    many = ops.linear_weight_grad(d2, x2, split=split)
    assert one.dtype == many.dtype == torch.float32 and many.shape == (n_out, n_in)
    assert torch.allclose(one, many, rtol=1e-4, atol=1e-3)
    # no gradient wanted: the plain library call, no autograd node of ours
    assert ops.linear(x, w.to(dtype), None).grad_fn is None


def test_host_paths_of_the_late_round6_train_ops_fall_back_to_torch():
    """Off the fused-op device (CPU tensors, HIP backend): ``FusedAdamW`` takes torch's own AdamW step, ``ops.stacked_linear`` is
    ``F.linear`` on the concatenated weight, ``train_attention`` equals ``F.scaled_dot_product_attention`` -- same numbers as
    the plain torch expressions, no kernel call."""
    from lina_speech_amd import ops
    from lina_speech_amd.blind_attention import train_attention
    from lina_speech_amd.train import FusedAdamW
    g = torch.Generator().manual_seed(5)
    ref = [torch.randn(7, 5, generator=g).requires_grad_(), torch.randn(11, generator=g).requires_grad_()]
    mine = [r.detach().clone().requires_grad_() for r in ref]
    kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.1)
    o_ref, o_mine = torch.optim.AdamW(ref, **kw), FusedAdamW(mine, **kw)
    for _ in range(3):
        for r, m in zip(ref, mine):
            r.grad = torch.randn(r.shape, generator=g)
            m.grad = r.grad.clone()
        o_ref.step()
        o_mine.step()
    for r, m in zip(ref, mine):
        assert torch.allclose(r, m, rtol=1e-6, atol=1e-7)
    parts = [torch.randn(4, 8, generator=g).requires_grad_(), torch.randn(2, 8, generator=g).requires_grad_()]
    x = torch.randn(3, 5, 8, generator=g)
    y = ops.stacked_linear(x, parts, pad=2)
    yr = torch.nn.functional.linear(x, torch.cat(parts + [torch.zeros(2, 8)], 0))
    assert torch.equal(y, yr)
    q, k, v = torch.randn(2, 1, 9, 16, generator=g), torch.randn(1, 1, 4, 16, generator=g), torch.randn(2, 1, 4, 16, generator=g)
    mask = torch.rand(2, 1, 9, 4, generator=g) > 0.3
    mask[..., 0] = True
    a = train_attention(q, k, v, mask)
    r = torch.nn.functional.scaled_dot_product_attention(q, k.expand(2, -1, -1, -1), v, attn_mask=mask)
    assert torch.allclose(a, r, rtol=1e-5, atol=1e-6)
