"""Shared kernel-vs-oracle checks.  `dev` = "cpu" (ops bound to the wave64 emulator) or
"cuda" (ops bound to the HIP library on a real MI355X).  The oracle always runs on the CPU
in fp64 on exactly the inputs the kernel saw (bf16 inputs are rounded first).

Tolerances, relative to max|reference| of the compared tensor:
    fp32 I/O : K1 / K3 / K4 / K5 / prologue   <= 1e-5
               K2 (different summation order, fast exp)  <= 1e-4
    bf16 I/O : outputs <= 2e-2, fp32 state <= 1e-2  (operands are rounded to bf16 for the MFMA)
    token / index outputs: exact.
"""
import torch
import torch.nn.functional as F

from lina_speech_amd import ops
from oracle import gla_oracle as O

F64 = torch.float64


def tol_out(dtype, chunk=False):
    if dtype == torch.bfloat16:
        return 2e-2
    return 1e-4 if chunk else 1e-5


PARITY_LOG = []          # (label, achieved max rel err, tolerance) of every assert_close / record_parity of this session;
                         # tests/conftest.py tags the entries with the running test and writes them out after a GPU session


def record_parity(what, achieved, tol=None, **extra):
    PARITY_LOG.append({"what": what, "achieved": float(achieved), "tolerance": None if tol is None else float(tol), **extra})


def assert_close(got, ref, rel, what):
    got, ref = got.detach().cpu().to(F64), ref.detach().cpu().to(F64)
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    scale = ref.abs().max().clamp_min(1e-30)
    err = (got - ref).abs().max() / scale
    record_parity(what, err, rel)
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    assert err <= rel, f"{what}: max rel err {err:.3e} > {rel:.1e}"


def make_gla_inputs(B, H, T, Dk, Dv, dtype, dev, seed=0, resets=False):
    """``resets``: False | True (reset gates, a cut-forcing run of them, one gate beyond the clamp) | "saturated" (see below)."""
    g = torch.Generator().manual_seed(seed)
    # projections arrive as [B,T,H*D]; the ops see the head-first VIEW (reference gla.py:173)
    def heads(x):
        return x.view(B, T, H, -1).transpose(1, 2)
    q = torch.randn(B, T, H * Dk, generator=g).to(dtype)
    k = torch.randn(B, T, H * Dk, generator=g).to(dtype)
    v = torch.randn(B, T, H * Dv, generator=g).to(dtype)
    gk = (F.logsigmoid(torch.randn(B, T, H * Dk, generator=g) * 2.0) / 4.0)
    if resets:
        gk[:, 5:9] = -20.0          # 4 consecutive resets: 80 > 60 forces a chunk cut
        gk[:, 17, ::3] = -20.0
        gk[:, 23, 1::2] = -70.0     # single gate beyond the clamp
        gk[:, 30:33, :7] = -25.0
    if resets == "saturated":       # ONE gate below the -60 clamp with every other gate of its chunk at 0, in a full chunk
        gk[:, 32:64] = 0.0          # (tokens 32..63) and in the last, partial one -- the optimistic scan flags it and rescans
        gk[:, 40, 3::5] = -70.0
        if T % 32 and T > 64:
            t0 = T - T % 32
            gk[:, t0:] = 0.0
            gk[:, t0 + 1, 1::4] = -75.0
    gk = gk.to(dtype)
    h0 = torch.randn(B, H, Dk, Dv, generator=g) * 0.5
    return [heads(x.to(dev)) for x in (q, k, v, gk)] + [h0.to(dev)]


def oracle_gla(q, k, v, gk, h0):
    n_thr = torch.get_num_threads()
    torch.set_num_threads(min(n_thr, 8))       # a long python loop of tiny ops: a 256-thread pool only adds wake-up cost
    try:
        return _oracle_gla(q, k, v, gk, h0)
    finally:
        torch.set_num_threads(n_thr)


def _oracle_gla(q, k, v, gk, h0):
    o, S = O.naive_recurrent_gla(q.cpu().to(F64), k.cpu().to(F64), v.cpu().to(F64), gk.cpu().to(F64),
                                 initial_state=None if h0 is None else h0.cpu().to(F64),
                                 output_final_state=True, compute_dtype=F64)
    return o, S


def check_recurrent(dev, B, H, T, Dk, Dv, dtype):
    q, k, v, gk, h0 = make_gla_inputs(B, H, T, Dk, Dv, dtype, dev)
    ro, rS = oracle_gla(q, k, v, gk, h0)
    o, S = ops.fused_recurrent_gla(q, k, v, gk, initial_state=h0, output_final_state=True)
    assert o.dtype == dtype and S.dtype == torch.float32 and S.data_ptr() != h0.data_ptr()
    assert_close(o, ro, tol_out(dtype), "K1 o")
    assert_close(S, rS, 1e-5 if dtype == torch.float32 else 1e-2, "K1 state")
    # no initial state, no final state
    o2, S2 = ops.fused_recurrent_gla(q, k, v, gk)
    assert S2 is None
    assert_close(o2, oracle_gla(q, k, v, gk, None)[0], tol_out(dtype), "K1 o (h0=None)")
    # in-place decode form
    h_in = h0.clone()
    o3, S3 = ops.fused_recurrent_gla(q, k, v, gk, initial_state=h_in, output_final_state=True, inplace_state=True)
    assert S3.data_ptr() == h_in.data_ptr()
    assert torch.equal(S3, S) and torch.equal(o3, o)
    # fp32 gates next to bf16 activations (decode path)
    if dtype == torch.bfloat16:
        o4, _ = ops.fused_recurrent_gla(q, k, v, gk.float(), initial_state=h0, output_final_state=True)
        assert_close(o4, ro, tol_out(dtype), "K1 o (f32 gates)")


def check_chunk(dev, B, H, T, Dk, Dv, dtype, resets=False):
    q, k, v, gk, h0 = make_gla_inputs(B, H, T, Dk, Dv, dtype, dev, seed=1, resets=resets)
    ro, rS = oracle_gla(q, k, v, gk, h0)
    for fn in (ops.chunk_gla, ops.fused_chunk_gla):
        o, S = fn(q, k, v, gk, initial_state=h0, output_final_state=True)
        assert o.dtype == dtype and S.dtype == torch.float32
        assert_close(o, ro, tol_out(dtype, chunk=True), f"K2 o ({fn.__name__})")
        assert_close(S, rS, 1e-4 if dtype == torch.float32 else 1e-2, "K2 state")
    o2, S2 = ops.chunk_gla(q, k, v, gk)
    assert S2 is None
    assert_close(o2, oracle_gla(q, k, v, gk, None)[0], tol_out(dtype, chunk=True), "K2 o (h0=None)")
    # chunk == recurrent kernel (self-consistency law)
    o3, S3 = ops.fused_recurrent_gla(q, k, v, gk, initial_state=h0, output_final_state=True)
    assert_close(o, o3.float(), 2 * tol_out(dtype, chunk=True), "K2 vs K1")


def check_chunk_dv512_one_launch(dev, monkeypatch, B, H, T, oracle=True):
    """256 x 512 heads (the reference's default expand_v = 2, model/gla.py:52,267): ONE launch with two workgroups per head
    (gla_chunk_full.hip NCB = 2) gives bit for bit what two launches on the value column blocks give (the recurrence is
    independent per value column) -- outputs and final state, with and without an initial state -- and both are the oracle's."""
    dtype = torch.bfloat16
    q, k, v, gk, h0 = make_gla_inputs(B, H, T, 256, 512, dtype, dev, seed=33, resets=True)
    res = {}
    for one in (True, False):
        monkeypatch.setattr(ops.POLICY, "dv512_one_launch", one)
        res[one] = (ops.chunk_gla(q, k, v, gk, initial_state=h0, output_final_state=True, nseg=1),
                    ops.chunk_gla(q, k, v, gk, nseg=1))
    (o1, S1), (o1n, _) = res[True]
    (o2, S2), (o2n, _) = res[False]
    assert torch.equal(o1, o2) and torch.equal(S1, S2) and torch.equal(o1n, o2n), "one launch != two launches"
    assert o1.shape == (B, H, T, 512) and S1.shape == (B, H, 256, 512)
    if oracle:
        ro, rS = oracle_gla(q, k, v, gk, h0)
        assert_close(o1, ro, tol_out(dtype, chunk=True), "K2 (256 x 512, one launch) o")
        assert_close(S1, rS, 1e-2, "K2 (256 x 512, one launch) state")


def check_chunk_segmented(dev, B, H, T, nseg, resets=False, D=256):
    """Segment-parallel K2 (state-only pass + combine + full pass) == the fp64 recurrent oracle and == the plain
    one-workgroup-per-head(-group) kernel, with and without an initial state; bf16, Dk = Dv = D (256, or 128 / 64 with
    2 / 4 heads per workgroup)."""
    dtype = torch.bfloat16
    q, k, v, gk, h0 = make_gla_inputs(B, H, T, D, D, dtype, dev, seed=21, resets=resets)
    ro, rS = oracle_gla(q, k, v, gk, h0)
    o, S = ops.chunk_gla(q, k, v, gk, initial_state=h0, output_final_state=True, nseg=nseg)
    assert_close(o, ro, tol_out(dtype, chunk=True), f"K2 segmented o (nseg={nseg})")
    assert_close(S, rS, 1e-2, "K2 segmented state")
    o1, S1 = ops.chunk_gla(q, k, v, gk, initial_state=h0, output_final_state=True, nseg=1)
    assert_close(o.float(), o1.float(), 2e-2, "K2 segmented vs plain")
    o2, S2 = ops.chunk_gla(q, k, v, gk, nseg=nseg)
    assert S2 is None
    assert_close(o2, oracle_gla(q, k, v, gk, None)[0], tol_out(dtype, chunk=True), "K2 segmented o (h0=None)")


def oracle_gla_grads_long(q, k, v, gk, h0, d_o, d_ht, seg=256):
    """Gradients of  sum(o * d_o) + sum(S_T * d_ht)  through the fp64 recurrent oracle at LONG T: exact torch autograd
    through oracle.naive_recurrent_gla, run segment by segment (forward once keeping only the segment-boundary states,
    then each segment again under autograd from the last to the first, the state gradient handed down) -- the same
    numbers as one autograd pass over all T steps at 1/(T/seg) of its memory (a 4096-step pass would keep ~20 GB of
    fp64 states alive)."""
    qd, kd, vd, gd, dod = (x.detach().cpu().to(F64) for x in (q, k, v, gk, d_o))
    n_thr = torch.get_num_threads()
    torch.set_num_threads(min(n_thr, 8))       # thousands of tiny ops: a 256-thread pool only adds wake-up cost
    try:
        return _oracle_gla_grads_long(qd, kd, vd, gd, dod, h0, d_ht, seg)
    finally:
        torch.set_num_threads(n_thr)


def _oracle_gla_grads_long(qd, kd, vd, gd, dod, h0, d_ht, seg):
    B, H, T, Dk = qd.shape
    Dv = vd.shape[-1]
    S = torch.zeros(B, H, Dk, Dv, dtype=F64) if h0 is None else h0.detach().cpu().to(F64)
    starts = []
    with torch.no_grad():
        for t0 in range(0, T, seg):
            starts.append(S)
            sl = slice(t0, min(T, t0 + seg))
            _, S = O.naive_recurrent_gla(qd[:, :, sl], kd[:, :, sl], vd[:, :, sl], gd[:, :, sl], initial_state=S,
                                         output_final_state=True, compute_dtype=F64)
    dS = torch.zeros_like(S) if d_ht is None else d_ht.detach().cpu().to(F64)
    grads = [torch.empty_like(x) for x in (qd, kd, vd, gd)]
    for i in reversed(range(len(starts))):
        t0 = i * seg
        sl = slice(t0, min(T, t0 + seg))
        leaves = [x[:, :, sl].clone().requires_grad_(True) for x in (qd, kd, vd, gd)]
        s_in = starts[i].clone().requires_grad_(True)
        o, s_out = O.naive_recurrent_gla(*leaves, initial_state=s_in, output_final_state=True, compute_dtype=F64)
        ((o * dod[:, :, sl]).sum() + (s_out * dS).sum()).backward()
        for gsum, leaf in zip(grads, leaves):
            gsum[:, :, sl] = leaf.grad
        dS = s_in.grad
    return grads, dS, S


def check_chunk_bwd_long(dev, B, H, T, Dk, Dv, dtype, reset_every=512, with_h0=True, with_dht=True):
    """K2b at the config-5 sequence length (SURVEY 8(d) adversarial set): model-like gates logsigmoid(.)/16 plus reset
    gates (-20, reference reset_val model/gla.py:136,183) on every channel at every ``reset_every``-th token and on a
    third of the channels half-way between -- many state renormalisations, segment boundaries of the sweeps, bf16
    error accumulated over 4096 tokens.  Tolerances relative to max|ref| per tensor: fp32 I/O 5e-4, bf16 I/O 2e-2."""
    g = torch.Generator().manual_seed(41)
    heads = lambda x: x.view(B, T, H, -1).transpose(1, 2)
    q = torch.randn(B, T, H * Dk, generator=g).to(dtype)
    k = torch.randn(B, T, H * Dk, generator=g).to(dtype)
    v = torch.randn(B, T, H * Dv, generator=g).to(dtype)
    gk = F.logsigmoid(torch.randn(B, T, H * Dk, generator=g)) / 16
    if reset_every:
        gk[:, reset_every - 1::reset_every] = -20.0
        gk[:, reset_every // 2::reset_every, ::3] = -20.0
    gk = gk.to(dtype)
    d_o = heads(torch.randn(B, T, H * Dv, generator=g).to(dtype).to(dev))
    h0 = (torch.randn(B, H, Dk, Dv, generator=g) * 0.5).to(dev) if with_h0 else None
    d_ht = (torch.randn(B, H, Dk, Dv, generator=g) * 0.3).to(dev) if with_dht else None
    q, k, v, gk = (heads(x.to(dev)) for x in (q, k, v, gk))
    leaves = [x.detach().clone().requires_grad_(True) for x in (q, k, v, gk)]
    lh0 = None if h0 is None else h0.detach().clone().requires_grad_(True)
    o, S = ops.chunk_gla(*leaves, initial_state=lh0, output_final_state=with_dht)
    loss = (o.float() * d_o.float()).sum()
    if with_dht:
        loss = loss + (S * d_ht).sum()
    loss.backward()
    (rq, rk, rv, rg), rdh0, rS = oracle_gla_grads_long(q, k, v, gk, h0, d_o, d_ht)
    tol = 2e-2 if dtype == torch.bfloat16 else 5e-4
    # dg_t = sum_{s >= t} (q_s (.) dq_s - k_s (.) dk_s): a suffix sum over up to T tokens of DIFFERENCES of products whose
    # factors went through bf16 MFMA operands (2^-9 each) -- the rounding noise of thousands of cancelling terms adds up
    # to a few per cent of max|dg| at T = 4096 (measured 2.4e-2; 1.5e-2 at T = 150); fp32 I/O (exact fp32 MFMA) stays 5e-4
    tol_g = 4e-2 if dtype == torch.bfloat16 else 5e-4
    if with_dht:
        assert_close(S, rS, 1e-2 if dtype == torch.bfloat16 else 2e-4, "K2 final state at long T")
    for name, a, r in zip(("dq", "dk", "dv", "dg"), leaves, (rq, rk, rv, rg)):
        assert_close(a.grad, r, tol_g if name == "dg" else tol, f"K2b {name} (T={T})")
    if h0 is not None:
        assert_close(lh0.grad, rdh0, 1e-2 if dtype == torch.bfloat16 else 5e-4, f"K2b dh0 (T={T})")


def check_chunk_bwd_full(dev, B, H, T, D, nseg, resets=False, with_h0=True, with_dht=True, seed=5):
    """K2b on the full-head kernel (lina_gla_chunk_bwd_full: reverse sweep -> dv, value-gated sweeps -> dq / dk + dg, ``nseg``
    sequence segments from boundary states) called directly, against torch autograd through the fp64 recurrent oracle.
    bf16 I/O: 2e-2 of max|ref| per tensor (4e-2 for dg at T >= 2048, see check_chunk_bwd_long)."""
    dtype = torch.bfloat16
    q, k, v, gk, h0 = make_gla_inputs(B, H, T, D, D, dtype, dev, seed=seed, resets=resets)
    if not with_h0:
        h0 = None
    g = torch.Generator().manual_seed(seed + 1)
    d_o = torch.randn(B, T, H * D, generator=g).to(dtype).to(dev).view(B, T, H, D).transpose(1, 2)
    d_ht = (torch.randn(B, H, D, D, generator=g) * 0.3).to(dev) if with_dht else None
    scale = D ** -0.5
    ht = None
    if with_dht:
        _, ht = ops.chunk_gla(q, k, v, gk, scale=scale, initial_state=h0, output_final_state=True)
    dq, dk, dv, dg, dh0 = ops.gla_chunk_bwd(q, k, v, gk, d_o, scale, h0, ht, d_ht, need_dh0=with_h0, nseg=nseg,
                                            path="full")
    if T <= 512:
        rl = [x.detach().cpu().to(F64).requires_grad_(True) for x in (q, k, v, gk)]
        rh0 = None if h0 is None else h0.detach().cpu().to(F64).requires_grad_(True)
        ro, rS = O.naive_recurrent_gla(*rl, initial_state=rh0, output_final_state=True, compute_dtype=F64)
        rloss = (ro * d_o.cpu().to(F64)).sum()
        if with_dht:
            rloss = rloss + (rS * d_ht.cpu().to(F64)).sum()
        rloss.backward()
        refs, rdh0 = [x.grad for x in rl], None if rh0 is None else rh0.grad
    else:
        refs, rdh0, _ = oracle_gla_grads_long(q, k, v, gk, h0, d_o, d_ht)
    tol_g = 4e-2 if T >= 2048 else 2e-2
    for name, a, r in zip(("dq", "dk", "dv", "dg"), (dq, dk, dv, dg), refs):
        assert_close(a, r, tol_g if name == "dg" else 2e-2, f"K2b(full, nseg={nseg}) {name}")
    if h0 is not None:
        assert_close(dh0, rdh0, 1e-2, f"K2b(full, nseg={nseg}) dh0")


def check_chunk_simple(dev, B, H, T, Dk, Dv, dtype, with_h0=True):
    """ops.chunk_simple_gla (fla.ops.simple_gla.chunk_simple_gla, reference model/gla.py:22, simple_gla.py:135 via the
    fla layer): scalar log-gate per head g [B,H,T] -> K2 with the gate broadcast over Dk; vs the fp64 scalar-gate
    recurrence of the oracle."""
    q, k, v, gk, h0 = make_gla_inputs(B, H, T, Dk, Dv, dtype, dev, seed=31)
    g = gk[..., 0].float().contiguous()                         # [B,H,T] fp32 (the layer computes it in fp32)
    if not with_h0:
        h0 = None
    ro, rS = O.simple_gla_recurrent(q.cpu().to(F64), k.cpu().to(F64), v.cpu().to(F64), g.cpu().to(F64),
                                    initial_state=None if h0 is None else h0.cpu().to(F64),
                                    output_final_state=True, compute_dtype=F64)
    o, S = ops.chunk_simple_gla(q, k, v, g, initial_state=h0, output_final_state=True)
    assert o.dtype == dtype and S.dtype == torch.float32
    assert_close(o, ro, tol_out(dtype, chunk=True), "simple-GLA o")
    assert_close(S, rS, 1e-4 if dtype == torch.float32 else 1e-2, "simple-GLA state")
    # gradients flow through the broadcast (dg = sum over Dk of the vector-gate gradient)
    if dtype == torch.float32 and T <= 80:
        leaves = [x.detach().clone().requires_grad_(True) for x in (q, k, v, g)]
        o2, _ = ops.chunk_simple_gla(*leaves)
        o2.square().sum().backward()
        rl = [x.detach().cpu().to(F64).requires_grad_(True) for x in (q, k, v, g)]
        ro2, _ = O.simple_gla_recurrent(*rl, compute_dtype=F64)
        ro2.square().sum().backward()
        for name, a, r in zip(("dq", "dk", "dv", "dg"), leaves, rl):
            assert_close(a.grad, r.grad, 5e-4, f"simple-GLA {name}")


def check_chunk_bwd(dev, B, H, T, Dk, Dv, dtype, resets=False, with_h0=True, with_dht=True, via="chunk_gla"):
    """K2b: gradients of (o, final_state) w.r.t. q, k, v, g, h0 against torch autograd through the fp64
    recurrent oracle.  Tolerances relative to max|ref|: fp32 I/O 2e-4 (different summation order, fast exp),
    bf16 I/O 2e-2 (gradients are rounded to bf16 once; the arithmetic is fp32)."""
    q, k, v, gk, h0 = make_gla_inputs(B, H, T, Dk, Dv, dtype, dev, seed=5, resets=resets)
    if not with_h0:
        h0 = None
    g = torch.Generator().manual_seed(6)
    d_o = torch.randn(B, T, H * Dv, generator=g).to(dtype).to(dev).view(B, T, H, Dv).transpose(1, 2)
    d_ht = (torch.randn(B, H, Dk, Dv, generator=g) * 0.3).to(dev) if with_dht else None
    leaves = [x.detach().clone().requires_grad_(True) for x in (q, k, v, gk)]
    lh0 = None if h0 is None else h0.detach().clone().requires_grad_(True)
    o, S = getattr(ops, via)(*leaves, initial_state=lh0, output_final_state=with_dht)
    loss = (o.float() * d_o.float()).sum()
    if with_dht:
        loss = loss + (S * d_ht).sum()
    loss.backward()
    # oracle
    rl = [x.detach().cpu().to(F64).requires_grad_(True) for x in (q, k, v, gk)]
    rh0 = None if h0 is None else h0.detach().cpu().to(F64).requires_grad_(True)
    ro, rS = O.naive_recurrent_gla(*rl, initial_state=rh0, output_final_state=True, compute_dtype=F64)
    rloss = (ro * d_o.cpu().to(F64)).sum()
    if with_dht:
        rloss = rloss + (rS * d_ht.cpu().to(F64)).sum()
    rloss.backward()
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-4
    for name, a, r in zip(("dq", "dk", "dv", "dg"), leaves, rl):
        assert a.grad is not None and a.grad.dtype == a.dtype, name
        assert_close(a.grad, r.grad, tol, f"K2b {name}")
    if h0 is not None:
        assert_close(lh0.grad, rh0.grad, 2e-4 if dtype == torch.float32 else 1e-2, "K2b dh0")


def check_conv(dev, B, T, D, W, dtype):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, T, D, generator=g).to(dtype).to(dev)
    w = (torch.randn(D, 1, W, generator=g) * 0.5).to(dtype).to(dev)
    mask = (torch.rand(B, T, generator=g) > 0.2).float().to(dev) if T > 1 else None
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    # no cache
    y = ops.short_conv(x, w, None, mask, None, "silu")
    ry = O.short_conv(x.cpu().to(F64), w.cpu().to(F64), None if mask is None else mask.cpu().to(F64), None)
    assert_close(y, ry, tol, "K3 y")
    # prefill into a cache, then two steps
    cache = torch.full((B, D, W), 7.0, dtype=dtype, device=dev)
    rcache = torch.full((B, D, W), 7.0, dtype=F64)
    if T > 1:
        y = ops.short_conv(x, w, None, mask, cache, "silu")
        ry = O.short_conv(x.cpu().to(F64), w.cpu().to(F64), mask.cpu().to(F64), rcache)
        assert_close(y, ry, tol, "K3 y (cache)")
        assert_close(cache, rcache, 1e-6 if dtype == torch.float32 else 1e-2, "K3 cache")
    for i in range(2):
        xs = torch.randn(B, 1, D, generator=g).to(dtype).to(dev)
        ys = ops.short_conv(xs, w, None, None, cache, "silu")
        rys = O.short_conv(xs.cpu().to(F64), w.cpu().to(F64), None, rcache)
        assert_close(ys, rys, tol, f"K4 y step {i}")
        assert_close(cache, rcache, 1e-6 if dtype == torch.float32 else 1e-2, f"K4 cache step {i}")
    # no activation + bias
    bias = torch.randn(D, generator=g).to(dtype).to(dev)
    y = ops.short_conv(x, w, bias, None, None, None)
    ry = O.short_conv(x.cpu().to(F64), w.cpu().to(F64), None, None, activation=None, bias=bias.cpu().to(F64))
    assert_close(y, ry, tol, "K3 y (bias, no act)")


def check_rmsnorm(dev, rows, D, dtype):
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(rows, 3, D, generator=g) * 3).to(dtype).to(dev)
    gate = torch.randn(rows, 3, D, generator=g).to(dtype).to(dev)
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(dtype).to(dev)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    y = ops.rmsnorm_swish_gate(x, gate, w, 1e-5)
    assert y.dtype == dtype and y.shape == x.shape
    assert_close(y, O.rmsnorm_swish_gate(x.cpu().to(F64), gate.cpu().to(F64), w.cpu().to(F64), 1e-5), tol, "K5 gate")
    y = ops.rmsnorm(x, w, 1e-5)
    assert_close(y, O.rmsnorm(x.cpu().to(F64), w.cpu().to(F64), 1e-5), tol, "K5 plain")
    y = ops.rmsnorm(x, None, 1e-5)
    assert_close(y, O.rmsnorm(x.cpu().to(F64), None, 1e-5), tol, "K5 no affine")
    # fp32 partial sums in, model dtype out (decode path)
    parts = torch.randn(2, rows, D, generator=g).to(dev)
    gate2 = gate[:, 0].contiguous()
    y = ops.rmsnorm_swish_gate(parts, gate2, w, 1e-5, n_partial=2, out_dtype=dtype)
    ry = O.rmsnorm_swish_gate(parts.cpu().to(F64).sum(0), gate2.cpu().to(F64), w.cpu().to(F64), 1e-5)
    assert y.dtype == dtype
    assert_close(y, ry, tol, "K5 partials")


def _grad_pair(dev, dtype, *tensors):
    mine = [None if t is None else t.to(dtype).to(dev).requires_grad_(True) for t in tensors]
    ref = [None if t is None else t.to(dtype).float().requires_grad_(True) for t in tensors]
    return mine, ref


def check_conv_bwd(dev, B, T, D, W, dtype, use_bias=False, activation="silu"):
    """K3b vs torch autograd through the oracle conv (fp32).  fp32: 2e-5; bf16: 2e-2 (dx rounded to bf16;
    dw/dbias accumulated in fp32 from bf16 inputs, compared at 2e-2 as well)."""
    g = torch.Generator().manual_seed(12)
    x = torch.randn(B, T, D, generator=g)
    w = torch.randn(D, 1, W, generator=g) * 0.5
    bias = torch.randn(D, generator=g) if use_bias else None
    mask = (torch.rand(B, T, generator=g) > 0.2).float()
    dy = torch.randn(B, T, D, generator=g).to(dtype)
    (mx, mw, mb), (rx, rw, rb) = _grad_pair(dev, dtype, x, w, bias)
    y = ops.short_conv(mx, mw, mb, mask.to(dev), None, activation)
    (y.float() * dy.to(dev).float()).sum().backward()
    ry = O.short_conv(rx, rw, mask, None, activation=activation, bias=rb)
    (ry * dy.float()).sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert_close(y, ry, tol, "K3 y (grad mode)")
    assert_close(mx.grad, rx.grad, tol, "K3b dx")
    assert mw.grad.shape == mw.shape
    assert_close(mw.grad, rw.grad, tol, "K3b dw")
    if use_bias:
        assert_close(mb.grad, rb.grad, tol, "K3b dbias")


def check_rmsnorm_bwd(dev, rows, D, dtype, gate=True, affine=True):
    """K5b vs torch autograd through the oracle norm (fp32).  fp32: 2e-5; bf16: 2e-2."""
    g = torch.Generator().manual_seed(13)
    x = torch.randn(rows, 2, D, generator=g) * 3
    gt = torch.randn(rows, 2, D, generator=g) if gate else None
    w = (1 + 0.1 * torch.randn(D, generator=g)) if affine else None
    dy = torch.randn(rows, 2, D, generator=g).to(dtype)
    (mx, mg, mw), (rx, rg, rw) = _grad_pair(dev, dtype, x, gt, w)
    y = ops.rmsnorm_swish_gate(mx, mg, mw, 1e-5) if gate else ops.rmsnorm(mx, mw, 1e-5)
    (y.float() * dy.to(dev).float()).sum().backward()
    ry = O.rmsnorm_swish_gate(rx, rg, rw, 1e-5) if gate else O.rmsnorm(rx, rw, 1e-5)
    (ry * dy.float()).sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert_close(y, ry, tol, "K5 y (grad mode)")
    assert_close(mx.grad, rx.grad, tol, "K5b dx")
    if gate:
        assert_close(mg.grad, rg.grad, tol, "K5b dg")
    if affine:
        assert_close(mw.grad, rw.grad, tol, "K5b dw")


def check_split_slab(dev, B, T, H, D, dtype, in_place=True):
    """split_slab: the consumers of a stacked projection's column slices (K3b conv backward, K5b norm-gate backward, one
    plain torch op) against torch autograd over the plain split: same dZ, same outputs; and the slab is REALLY written in
    place by the two kernels (one torch copy -- the plain consumer's slice -- in the backward, not three)."""
    g = torch.Generator().manual_seed(29)
    Kd, W, L = H * D, 4, 16
    z = torch.randn(B, T, 2 * Kd + L + 4, generator=g)
    o = torch.randn(B, T, H, D, generator=g) * 2
    cw = torch.randn(Kd, W, generator=g) * 0.5
    nw = 1 + 0.1 * torch.randn(D, generator=g)
    dq = torch.randn(B, T, Kd, generator=g).to(dtype)
    do = torch.randn(B, T, H, D, generator=g).to(dtype)
    (mz, mo, mcw, mnw), (rz, ro, rcw, rnw) = _grad_pair(dev, dtype, z, o, cw, nw)
    sizes = [Kd, L, Kd, 4]
    (q, lr, gt, rest), slab = ops.split_slab(mz, sizes)
    assert slab is not None
    qc = ops.short_conv(q, mcw, None, None, None, "silu", grad_slab=(slab, 0) if in_place else None)
    on = ops.rmsnorm_swish_gate(mo, gt.view(B, T, H, D), mnw, 1e-5, grad_slab=(slab, 2) if in_place else None)
    loss = ((qc.float() * dq.to(dev).float()).sum() + (on.float() * do.to(dev).float()).sum()
            + (lr.float() ** 2).sum() * 0.5)                  # `rest` gets no gradient: its columns must come out zero
    loss.backward()
    rq, rlr, rgt, _ = rz.split(sizes, dim=-1)
    rqc = O.short_conv(rq, rcw, None, None, "silu")
    ron = O.rmsnorm_swish_gate(ro, rgt.view(B, T, H, D), rnw, 1e-5)
    ((rqc * dq.float()).sum() + (ron * do.float()).sum() + (rlr ** 2).sum() * 0.5).backward()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert_close(qc, rqc, tol, "slab conv y")
    assert_close(on, ron, tol, "slab norm y")
    assert mz.grad.shape == z.shape and mz.grad.is_contiguous()
    assert_close(mz.grad, rz.grad, tol, "slab dZ")
    assert torch.count_nonzero(mz.grad[..., -4:]) == 0
    assert_close(mo.grad, ro.grad, tol, "slab do")
    assert_close(mcw.grad, rcw.grad, tol, "slab dw conv")
    assert_close(mnw.grad, rnw.grad, tol, "slab dw norm")
    assert slab.buf is None                                   # handed over to autograd, not kept alive by the holder
    assert slab.copied == ([1] if in_place else [0, 1, 2]), slab.copied


def check_short_conv3(dev, B, T, H, D, dtype, use_bias=False, through_gla=True):
    """short_conv3: the q | k | v convolutions of a stacked projection in one launch each way, feeding K2 / K2b through
    STRIDED views of one output buffer and taking K2b's dq | dk | dv as one operand: outputs, dZ and the three filter (and
    bias) gradients against torch autograd through the oracle conv + the oracle recurrence; the slab must be written in place
    by the one launch (no copies) and the fused form must really have been taken."""
    g = torch.Generator().manual_seed(31)
    Kd, W = H * D, 4
    z = torch.randn(B, T, 3 * Kd + 8, generator=g)
    ws = [torch.randn(Kd, 1, W, generator=g) * 0.5 for _ in range(3)]
    bs = [torch.randn(Kd, generator=g) * 0.1 if use_bias else None for _ in range(3)]
    mask = (torch.rand(B, T, generator=g) > 0.1).float()
    gk = (torch.nn.functional.logsigmoid(torch.randn(B, T, H, D, generator=g)) / 16)
    do = torch.randn(B, T, H, D, generator=g).to(dtype)
    dys = [torch.randn(B, T, Kd, generator=g).to(dtype) for _ in range(3)]
    (mz, mgk, *mwb), (rz, rgk, *rwb) = _grad_pair(dev, dtype, z, gk, *ws, *bs)
    sizes = [Kd, Kd, Kd, 8]
    (q, k, v, rest), slab = ops.split_slab(mz, sizes)
    out = ops.short_conv3((q, k, v), mwb[:3], mwb[3:], mask.to(dev), "silu", grad_slab=(slab, 0))
    assert out is not None, "the fused form was refused"
    heads = lambda t: t.view(B, T, H, D).transpose(1, 2)
    rq, rk, rv, _ = rz.split(sizes, dim=-1)
    rout = [O.short_conv(x, w, mask, None, activation="silu", bias=b) for x, w, b in zip((rq, rk, rv), rwb[:3], rwb[3:])]
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for a, b, n in zip(out, rout, "qkv"):
        assert_close(a, b, tol, f"conv3 {n}")
    if through_gla:
        o, _ = ops.chunk_gla(heads(out[0]), heads(out[1]), heads(out[2]), heads(mgk))
        (o.float() * heads(do.to(dev)).float()).sum().backward()
        ro, _ = O.naive_recurrent_gla(heads(rout[0]), heads(rout[1]), heads(rout[2]), heads(rgk))
        (ro * heads(do.float())).sum().backward()
        tol = 5e-4 if dtype == torch.float32 else 4e-2
    else:
        sum((a.float() * d.to(dev).float()).sum() for a, d in zip(out, dys)).backward()
        sum((a * d.float()).sum() for a, d in zip(rout, dys)).backward()
    assert_close(mz.grad, rz.grad, tol, "conv3 dZ")
    assert torch.count_nonzero(mz.grad[..., -8:]) == 0
    for i in range(3):
        assert mwb[i].grad.shape == ws[i].shape
        assert_close(mwb[i].grad, rwb[i].grad, tol, f"conv3 dw{i}")
        if use_bias:
            assert_close(mwb[3 + i].grad, rwb[3 + i].grad, tol, f"conv3 db{i}")
    assert slab.copied == [], slab.copied                      # q | k | v columns written in place by the ONE launch


def check_embed_bwd(dev, Q, B, n, n_emb, d, dtype):
    g = torch.Generator().manual_seed(14)
    table = torch.randn(Q, n_emb, d, generator=g)
    idx = torch.randint(0, n_emb, (Q, B, n), generator=g)
    dy = torch.randn(B, n, d, generator=g).to(dtype)
    (mt,), (rt,) = _grad_pair(dev, dtype, table)
    y = ops.embed_sum(mt, idx.to(dev))
    (y.float() * dy.to(dev).float()).sum().backward()
    ry = O.embed_sum(rt, idx)
    (ry * dy.float()).sum().backward()
    assert_close(mt.grad, rt.grad, 1e-5 if dtype == torch.float32 else 1e-2, "K6 dtable")


def check_embed(dev, Q, B, n, n_emb, d, dtype):
    g = torch.Generator().manual_seed(4)
    table = torch.randn(Q, n_emb, d, generator=g).to(dtype).to(dev)
    idx = torch.randint(0, n_emb, (Q, B, n), generator=g).to(dev)
    out = ops.embed_sum(table, idx)
    ref = O.embed_sum(table.cpu().to(F64), idx.cpu())
    assert out.shape == (B, n, d)
    assert_close(out, ref, 1e-6 if dtype == torch.float32 else 1e-2, "K6a")


def check_greedy_pick_embed(dev, B, Q, L, d, dtype, steps=3):
    """K6d: arg-max per quantizer (lowest index on exact ties), token log at the device step, next-input embedding
    sum and the step increment -- against the oracle helpers (reference tools.py:38-44 at k = 1, multiembed.py:21-23)."""
    g = torch.Generator().manual_seed(17)
    n_emb = L + 3
    table = torch.randn(Q, n_emb, d, generator=g).to(dtype).to(dev)
    tok_log = torch.full((steps + 1, Q, B), -1, dtype=torch.int64, device=dev)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    for t in range(steps + 2):                                   # two steps past the end of the log: not written
        logits = torch.randn(B, Q, L, generator=g).to(dtype)
        logits[0, 0, 5] = logits[0, 0, 9] = 50.0                 # an exact tie: the lowest index wins
        logits = logits.to(dev)
        x = torch.full((B, d), float("nan"), dtype=dtype, device=dev)
        ops.greedy_pick_embed(logits, table, x, tok_log, step, counter)
        ref = O.argmax_lowest(logits.cpu().float().reshape(B * Q, L)).view(B, Q).t().contiguous()   # [Q,B]
        assert int(step) == t + 1 and int(counter) == 0
        if t <= steps:
            assert torch.equal(tok_log[t].cpu(), ref), f"picks differ at step {t}"
        assert int(ref[0, 0]) == 5
        rx = O.embed_sum(table.cpu().to(F64), ref.unsqueeze(-1)).squeeze(1)
        assert_close(x, rx, 1e-2 if dtype == torch.bfloat16 else 1e-6, "K6d next-input embedding")


def check_pick_loop_ctl(dev, B, Q, L, d, dtype, sampled=False):
    """The loop-control block of K6d / K6e (include/lina_gla.h): rows that picked the stop token (id 2) on EVERY quantizer are
    counted once, word [1] = the first step at which all rows have -- the reference's is_stop_token / all_stop_token / break
    of model/modeling_lina.py:168-173 replayed on the host from the same logits; and (K6e) the per-call seed word: a block
    carrying word w with seed s draws what a block-less call draws with seed s ^ w."""
    g = torch.Generator().manual_seed(53)
    n_emb = L + 3
    table = torch.randn(Q, n_emb, d, generator=g).to(dtype).to(dev)
    steps = 6
    tok_log = torch.zeros(steps, Q, B, dtype=torch.int64, device=dev)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    word = 0x9E3779B97F4A7C15
    ctl = ops.new_loop_ctl(B, dev, word if sampled else 0)
    R = ops.LOOP_CTL_ROWS
    assert ctl.tolist()[:2] == [0, -1]
    stop_plan = {0: [0], 1: [], 2: [0, 1], 3: list(range(1, B - 1)), 4: [B - 1], 5: [0]}     # rows forced to stop per step
    seen = torch.zeros(B, dtype=torch.bool)
    first_all = -1
    for t in range(steps):
        logits = torch.randn(B, Q, L, generator=g).to(dtype)
        logits[:, :, 2] = -40.0                                   # nobody picks the stop token by chance ...
        for b in stop_plan[t]:
            logits[b, :, 2] = 40.0                                # ... these rows pick it on every quantizer
        if B > 1 and Q > 1:
            logits[B - 1, 0, 2] = 40.0 if t == 1 else logits[B - 1, 0, 2]     # one quantizer only: NOT a stop
        logits = logits.to(dev)
        x = torch.empty(B, d, dtype=dtype, device=dev)
        if sampled:
            ref = ops.topk_sample_rows(logits, 3, 0.7, seed=5 ^ word, step=step.clone()).t().contiguous()      # [Q,B]
            ops.sample_pick_embed(logits, table, x, tok_log, step, counter, Q, 3, 0.7, seed=5, loop_ctl=ctl)
            assert torch.equal(tok_log[t], ref), "seed word: the draws differ from those of seed ^ word"
        else:
            ops.greedy_pick_embed(logits, table, x, tok_log, step, counter, loop_ctl=ctl)
        picks = tok_log[t].cpu()                                                   # [Q,B]
        is_stop = (picks == 2).all(dim=0)
        for b in stop_plan[t]:
            assert bool(is_stop[b])
        seen |= is_stop
        if first_all < 0 and bool(seen.all()):
            first_all = t
        got = ctl.cpu().tolist()
        assert got[0] == int(seen.sum()) and got[1] == first_all, (t, got[:2], int(seen.sum()), first_all)
        assert got[R:R + B] == seen.int().tolist()
    assert first_all == 4


def check_sample_pick_embed(dev, B, Q, L, d, dtype, n_sampled, k=7, temp=0.8, seed=11, steps=4):
    """K6e: the one-launch token epilogue with the first ``n_sampled`` quantizers sampled.  Picks must EQUAL the separate
    launches it replaces at the same (seed, device step): K6c over the [B*Q] rows for the sampled quantizers (itself checked
    against the fp64 inverse-CDF oracle in check_topk_sample), K6b for the others; then token log, embedding (row-major and
    fragment-major) and the step increment as K6d."""
    g = torch.Generator().manual_seed(29)
    n_emb = L + 3
    table = torch.randn(Q, n_emb, d, generator=g).to(dtype).to(dev)
    tok_log = torch.full((steps, Q, B), -1, dtype=torch.int64, device=dev)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    kq = 32 if dtype == torch.bfloat16 else 16
    use_packed = d % kq == 0
    for t in range(steps):
        logits = (torch.randn(B, Q, L, generator=g) * 2).to(dtype).to(dev)
        step_before = step.clone()
        samp = ops.topk_sample_rows(logits, k, temp, seed=seed, step=step_before)          # [B,Q]
        amax = ops.argmax_rows(logits)
        ref = torch.where(torch.arange(Q, device=dev).unsqueeze(0) < n_sampled, samp, amax).t().contiguous()   # [Q,B]
        x = torch.full((B, d), float("nan"), dtype=dtype, device=dev)
        x_p = torch.zeros(ops.packed_numel(B, d), dtype=dtype, device=dev) if use_packed else None
        ops.sample_pick_embed(logits, table, x, tok_log, step, counter, n_sampled, k, temp, seed=seed, x_packed=x_p)
        assert int(step) == t + 1 and int(counter) == 0
        assert torch.equal(tok_log[t], ref), f"picks differ at step {t}"
        rx = ops.embed_sum(table, ref)
        assert torch.equal(x, rx), "K6e next-input embedding differs from K6a on the same picks"
        if use_packed:
            assert torch.equal(ops.unpack_rows(x_p, B, d), x), "K6e packed copy differs"
        if n_sampled:       # a draw depends on the step: the same logits at the next step give other uniforms
            assert bool((samp >= 0).all()) and bool((samp < L).all())
    if n_sampled == 0:      # degenerates to K6d
        x2 = torch.empty(B, d, dtype=dtype, device=dev)
        log2 = torch.full((1, Q, B), -1, dtype=torch.int64, device=dev)
        ops.greedy_pick_embed(logits, table, x2, log2, torch.zeros(1, dtype=torch.int64, device=dev),
                              torch.zeros(1, dtype=torch.int32, device=dev))
        assert torch.equal(log2[0], tok_log[steps - 1]) and torch.equal(x2, x)


def check_layer_norm(dev, N, D, x_dtype, r_dtype, y_dtype):
    """K10 forward + backward against fp64 autograd through  y = LayerNorm(x + r) * gamma + beta,  loss = <y, wy> + <x + r, ws>
    (the second term sends a pass-through gradient into the residual stream, as the next block does)."""
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(N, D, generator=g) * 2 + 0.5).to(x_dtype)
    r = None if r_dtype is None else torch.randn(N, D, generator=g).to(r_dtype)
    gamma = 1 + 0.3 * torch.randn(D, generator=g)
    beta = 0.2 * torch.randn(D, generator=g)
    wy = torch.randn(N, D, generator=g).to(y_dtype)
    ws = torch.randn(N, D, generator=g).to(x_dtype)
    # reference in fp64 on the SAME (dtype-rounded) inputs
    x64, r64 = x.to(F64).requires_grad_(), (None if r is None else r.to(F64).requires_grad_())
    g64, b64 = gamma.to(F64).requires_grad_(), beta.to(F64).requires_grad_()
    xs64 = x64 if r is None else x64 + r64
    y64 = F.layer_norm(xs64, (D,), g64, b64, 1e-5)
    loss = (y64 * wy.to(F64)).sum() + ((xs64 * ws.to(F64)).sum() if r is not None else 0.0)
    loss.backward()
    xd, rd = x.to(dev).requires_grad_(), (None if r is None else r.to(dev).requires_grad_())
    gd, bd = gamma.to(dev).requires_grad_(), beta.to(dev).requires_grad_()
    out = ops.layer_norm(xd, gd, bd, 1e-5, residual=rd, out_dtype=y_dtype)
    y, xs = (out, None) if r is None else out
    assert y.dtype == y_dtype and (xs is None or xs.dtype == x_dtype)
    lo = y_dtype == torch.bfloat16 or x_dtype == torch.bfloat16
    assert_close(y, y64.detach(), 1e-2 if lo else 2e-6, "K10 y")
    if xs is not None:
        assert_close(xs, xs64.detach(), 8e-3 if x_dtype == torch.bfloat16 else 1e-6, "K10 x + r")
    l2 = (y.float() * wy.to(dev).float()).sum() + ((xs.float() * ws.to(dev).float()).sum() if xs is not None else 0.0)
    l2.backward()
    tol = 2e-2 if lo else 2e-5
    assert_close(xd.grad, x64.grad, tol, "K10 dx")
    if r is not None:
        assert rd.grad.dtype == r_dtype
        assert_close(rd.grad, r64.grad, tol, "K10 dr")
    assert_close(gd.grad, g64.grad, tol, "K10 dgamma")
    assert_close(bd.grad, b64.grad, tol, "K10 dbeta")


def check_swiglu_gate(dev, N, H, dtype):
    """K11 / K11b: silu(a) * b and its gradient against fp64 autograd (odd widths take the scalar kernels)."""
    g = torch.Generator().manual_seed(37)
    u = (torch.randn(N, 2 * H, generator=g) * 1.5).to(dtype)
    w = torch.randn(N, H, generator=g).to(dtype)
    u64 = u.to(F64).requires_grad_()
    a, b = u64.chunk(2, -1)
    y64 = F.silu(a) * b
    (y64 * w.to(F64)).sum().backward()
    ud = u.to(dev).requires_grad_()
    y = ops.swiglu_gate(ud)
    lo = dtype == torch.bfloat16
    assert_close(y, y64.detach(), 1e-2 if lo else 2e-6, "K11 silu(a) b")
    (y.float() * w.to(dev).float()).sum().backward()
    assert_close(ud.grad, u64.grad, 1.5e-2 if lo else 2e-6, "K11b du")


def check_swiglu_unit_column(dev, dtype):
    """The padded channel mixer (`_SwiGLUMLPFunction`) carries the down-projection's bias in column H of its weight and relies on
    the gate producing EXACTLY 1 there from the bias pair (32, 1/32): silu(32) * (1/32) == 1 in the gate kernel's arithmetic.  Any
    change to the device silu (a faster exp / reciprocal) that breaks this would silently bias the output and its gradient."""
    from lina_speech_amd.autograd import _mlp_one
    one = _mlp_one(dtype, dev)
    for width in (8, 7):                                     # vector and scalar kernels
        u = torch.zeros(5, 2 * width, dtype=dtype, device=dev)
        u[:, 3] = one[0]
        u[:, width + 3] = one[1]
        h = ops.swiglu_gate(u)
        assert torch.equal(h[:, 3].float().cpu(), torch.ones(5)), (dtype, width, h[:, 3].tolist())
        assert float(h.float().abs().sum()) == 5.0           # silu(0) * 0 == 0 everywhere else


def check_gate_logsigmoid(dev, n, dtype, clamp):
    """K12: logsigmoid(x) / normalizer (+ clamp) and its gradient against fp64 autograd, over the range the fast
    log1p / exp forms switch in."""
    g = torch.Generator().manual_seed(41)
    x = (torch.randn(n, generator=g) * 6).to(dtype)
    x[:8] = torch.tensor([-90.0, -20.0, -4.2, -1e-3, 0.0, 4.2, 20.0, 90.0]).to(dtype)
    w = torch.randn(n, generator=g).to(dtype)
    x64 = x.to(F64).requires_grad_()
    y64 = F.logsigmoid(x64) / 16.0
    if clamp is not None:
        y64 = torch.clamp_min(y64, clamp)
    (y64 * w.to(F64)).sum().backward()
    xd = x.to(dev).requires_grad_()
    y = ops.gate_logsigmoid(xd, 16.0, clamp)
    assert y.dtype == dtype and y.shape == x.shape
    lo = dtype == torch.bfloat16
    assert_close(y, y64.detach(), 8e-3 if lo else 2e-6, "K12 gate")
    (y.float() * w.to(dev).float()).sum().backward()
    gd, g64 = xd.grad.double().cpu(), x64.grad
    if clamp is not None:                      # elements within rounding of the clamp edge may take either side
        edge = (F.logsigmoid(x.to(F64)) / 16.0 - clamp).abs() < (2e-2 if lo else 1e-6)
        gd, g64 = gd[~edge], g64[~edge]
    assert_close(gd, g64, 8e-3 if lo else 2e-6, "K12 dx")


def check_gate_lowrank(dev, B, T, C, L, dtype, clamp, bias=True, strided=False):
    """K12b: logsigmoid(lr W^T + b) / normalizer (+ clamp) and the gradients of lr, W, b against fp64 autograd of the
    unfused chain on the operands the kernel sees (weights rounded to the GEMM dtype, as autocast does)."""
    g = torch.Generator().manual_seed(43)
    # strided: True = a column slice at an 8-byte offset (the K12b kernel), "aligned" = a 16-byte aligned slice (bf16, L = 16,
    # C % 64 == 0: the matrix-core form K12c -- the layout of the mixer's stacked projection)
    off = 8 if strided == "aligned" else 4
    z = torch.randn(B, T, L + 16, generator=g).to(dtype)
    lr0 = z[..., off:off + L] if strided else z[..., :L].contiguous()
    w = torch.randn(C, L, generator=g) * 1.5
    b = torch.randn(C, generator=g) if bias else None
    dy = torch.randn(B, T, C, generator=g).to(dtype)
    lr64 = lr0.to(F64).requires_grad_()
    w64 = w.to(dtype).to(F64).requires_grad_()
    b64 = None if b is None else b.to(dtype).to(F64).requires_grad_()
    pre64 = F.linear(lr64, w64, b64)
    pre64 = pre64 + (pre64.to(dtype).to(F64) - pre64).detach()        # the GEMM's output rounding (straight-through)
    y64 = F.logsigmoid(pre64) / 16.0
    if clamp is not None:
        y64 = torch.clamp_min(y64, clamp)
    (y64 * dy.to(F64)).sum().backward()
    zd = z.to(dev)
    lrd = (zd[..., off:off + L] if strided else zd[..., :L].contiguous()).requires_grad_()
    wd = w.to(dev).requires_grad_()
    bd = None if b is None else b.to(dev).requires_grad_()
    y = ops.gate_lowrank(lrd, wd, bd, 16.0, clamp)
    assert y.dtype == dtype and y.shape == (B, T, C)
    lo = dtype == torch.bfloat16
    assert_close(y, y64.detach(), 2e-2 if lo else 1e-5, "K12b gate")         # bf16: pre is rounded to bf16 first
    (y.float() * dy.to(dev).float()).sum().backward()
    assert wd.grad.dtype == torch.float32 and wd.grad.shape == w.shape
    if clamp is None:
        assert_close(lrd.grad, lr64.grad, 3e-2 if lo else 2e-5, "K12b dlr")
        assert_close(wd.grad, w64.grad, 2e-2 if lo else 2e-5, "K12b dW")
        if bias:
            assert_close(bd.grad, b64.grad, 2e-2 if lo else 2e-5, "K12b db")
    else:                                      # positions at the clamp edge may fall on either side: compare the bulk
        assert_close(wd.grad, w64.grad, 5e-2, "K12b dW (clamped)")


def check_swiglu_mlp(dev, B, T, d, H, dtype, bias=True):
    """ops.swiglu_mlp (padded one-node channel mixer: K11, K11c, bias column) against fp64 autograd of the plain chain on
    the same (dtype-rounded) operands: y and every gradient."""
    g = torch.Generator().manual_seed(47)
    x = torch.randn(B, T, d, generator=g).to(dtype)
    w_in = (torch.randn(2 * H, d, generator=g) / d ** 0.5).to(dtype)
    b_in = torch.randn(2 * H, generator=g).to(dtype) if bias else None
    w_out = (torch.randn(d, H, generator=g) / H ** 0.5).to(dtype)
    b_out = torch.randn(d, generator=g).to(dtype) if bias else None
    dy = torch.randn(B, T, d, generator=g).to(dtype)
    ts = [x, w_in, b_in, w_out, b_out]
    r = [None if t is None else t.to(F64).requires_grad_() for t in ts]
    a, b = F.linear(r[0], r[1], r[2]).chunk(2, -1)
    y64 = F.linear(F.silu(a) * b, r[3], r[4])
    (y64 * dy.to(F64)).sum().backward()
    m = [None if t is None else t.to(dev).requires_grad_() for t in ts]
    y = ops.swiglu_mlp(*m)
    assert y.dtype == dtype and y.shape == x.shape[:-1] + (d,)
    assert type(y.grad_fn).__name__ == "_SwiGLUMLPFunctionBackward", type(y.grad_fn).__name__
    lo = dtype == torch.bfloat16
    tol = 3e-2 if lo else 2e-5
    assert_close(y, y64.detach(), tol, "MLP y")
    (y.float() * dy.to(dev).float()).sum().backward()
    for mine, ref, what in zip(m, r, ("dx", "dW_in", "db_in", "dW_out", "db_out")):
        if mine is not None:
            assert mine.grad.shape == ref.grad.shape and mine.grad.dtype == dtype, what
            assert_close(mine.grad, ref.grad, tol, "MLP " + what)
    # the padded weights are kept per parameter version: a second forward on the same weights reuses them (same result), an
    # in-place update of ANY of the four parameters invalidates them
    from lina_speech_amd.autograd import _MLP_PACK
    pack0 = _MLP_PACK[m[1]][1]
    y2 = ops.swiglu_mlp(*m)
    assert _MLP_PACK[m[1]][1] is pack0 and torch.equal(y2, y)
    with torch.no_grad():
        m[3].mul_(2.0)
        if bias:
            m[4].mul_(2.0)
    y3 = ops.swiglu_mlp(*m)
    assert _MLP_PACK[m[1]][1] is not pack0
    assert_close(y3, 2.0 * y64.detach(), tol, "MLP y after an in-place weight update")
    # a write through ``.data`` does NOT bump the version (EMA swaps, weight clipping): the documented way out is
    # ops.clear_mlp_pack() (also part of ops.clear_workspaces and of TrainStep.step)
    pack3 = _MLP_PACK[m[1]][1]
    m[3].data.mul_(0.5)                                      # back to the original weights, behind the version counter's back
    if bias:
        m[4].data.mul_(0.5)
    assert ops.swiglu_mlp(*m) is not None and _MLP_PACK[m[1]][1] is pack3, "expected: the stale pack is still served"
    ops.clear_mlp_pack()
    y4 = ops.swiglu_mlp(*m)
    assert_close(y4, y64.detach(), tol, "MLP y after a .data write + clear_mlp_pack()")


def check_mlp_pack(dev, H=45, d_in=24, d_out=20, out_dtype=torch.bfloat16, bias=True):
    """K15 ``lina_mlp_pack`` (the channel mixer's padded GEMM operands in one pass over the fp32 master weights) against the
    torch construction it replaces -- bit for bit: zero pads, the (32, 1/32) bias pair, b_out in column H."""
    from lina_speech_amd.autograd import _mlp_padded_weights, _MLP_PAD
    g = torch.Generator().manual_seed(61)
    w_in = torch.randn(2 * H, d_in, generator=g).to(dev)
    b_in = torch.randn(2 * H, generator=g).to(dev) if bias else None
    w_out = torch.randn(d_out, H, generator=g).to(dev)
    b_out = torch.randn(d_out, generator=g).to(dev) if bias else None
    Hp = (H + _MLP_PAD) // _MLP_PAD * _MLP_PAD
    ops.clear_mlp_pack()
    Wi, bi, Wo, Wo_wide = _mlp_padded_weights(w_in, b_in, w_out, b_out, out_dtype, H, Hp)
    assert Wo_wide.shape[1] >= Hp and Wo.data_ptr() == Wo_wide.data_ptr() and float(Wo_wide[:, Hp:].abs().sum()) == 0.0
    rWi = torch.zeros(2, Hp, d_in, dtype=out_dtype, device=dev)
    rWi[:, :H] = w_in.view(2, H, d_in).to(out_dtype)
    rbi = torch.zeros(2, Hp, dtype=out_dtype, device=dev)
    rWo = torch.zeros(d_out, Hp, dtype=out_dtype, device=dev)
    rWo[:, :H] = w_out.to(out_dtype)
    if bias:
        rbi[:, :H] = b_in.view(2, H).to(out_dtype)
        rbi[0, H], rbi[1, H] = 32.0, 1.0 / 32.0
        rWo[:, H] = b_out.to(out_dtype)
    assert Wi.dtype == out_dtype and torch.equal(Wi, rWi) and torch.equal(bi, rbi) and torch.equal(Wo, rWo)
    ops.clear_mlp_pack()


def check_stacked_linear(dev, rows=(8, 8, 16, 16, 4), n_in=24, pad=12, B=3, T=7, autocast=False, expect_split=None):
    """``ops.stacked_linear`` (K16: the stacked operand in one pass, block gradients as row ranges of dW) against
    ``F.linear(x, cat(parts + zero rows))`` through fp64 autograd; the blocks' gradients are contiguous views of ONE tensor."""
    g = torch.Generator().manual_seed(62)
    parts64 = [torch.randn(r, n_in, generator=g).to(F64).requires_grad_() for r in rows]
    x64 = torch.randn(B, T, n_in, generator=g).to(F64).requires_grad_()
    dy = torch.randn(B, T, sum(rows) + pad, generator=g)
    y64 = F.linear(x64, torch.cat(parts64 + [torch.zeros(pad, n_in, dtype=F64)], 0))
    (y64 * dy.to(F64)).sum().backward()
    parts = [p.detach().float().to(dev).requires_grad_() for p in parts64]
    x = x64.detach().float().to(dev).requires_grad_()
    if autocast:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = ops.stacked_linear(x, parts, pad)
        assert y.dtype == torch.bfloat16
        tol = 3e-2
    else:
        y = ops.stacked_linear(x, parts, pad)
        tol = 2e-5
    assert type(y.grad_fn).__name__ == "_StackedLinearFunctionBackward", type(y.grad_fn).__name__
    assert y.shape == (B, T, sum(rows) + pad)
    if expect_split is not None:                # the 256-aligned main product + narrow tail form of the GEMMs
        assert (y.grad_fn.main < sum(rows) + pad) == expect_split, y.grad_fn.main
    assert_close(y, y64.detach(), tol, "stacked linear y")
    assert pad == 0 or float(y.detach()[..., sum(rows):].abs().max()) == 0.0, "pad columns must be exactly zero"
    (y.float() * dy.to(dev)).sum().backward()
    assert_close(x.grad, x64.grad, tol, "stacked linear dx")
    for i, (p, r) in enumerate(zip(parts, parts64)):
        assert p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.shape == r.grad.shape
        assert_close(p.grad, r.grad, tol, f"stacked linear dW[{i}]")


def check_fused_adamw(dev, n_small=60, steps=3):
    """``train.FusedAdamW`` (K17 lina_adamw_multi) against ``torch.optim.AdamW`` on the same parameters / gradients for a few
    steps: parameters and both moments to fp32 round-off; more tensors than one launch takes, sizes that are not multiples of
    4 or of a block, a parameter without a gradient, a changing lr; the state dict of one loads into the other."""
    from lina_speech_amd.train import FusedAdamW
    g = torch.Generator().manual_seed(71)
    shapes = [(4099, 64), (1024, 1365), (7,), (1,), (4096,), (4097,), (3, 5, 11)] + [(17 + i,) for i in range(n_small)]
    ref = [torch.randn(*s, generator=g).requires_grad_() for s in shapes]
    mine = [r.detach().clone().to(dev).requires_grad_() for r in ref]
    kw = dict(lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    o_ref, o_mine = torch.optim.AdamW(ref, foreach=False, **kw), FusedAdamW(mine, **kw)
    for it in range(steps):
        for r, m in zip(ref, mine):
            gr = torch.randn(r.shape, generator=g)
            r.grad, m.grad = gr.clone(), gr.clone().to(dev)
        if it == 1:                                  # a parameter that skips a step (its own step count)
            ref[3].grad = mine[3].grad = None
        for o in (o_ref, o_mine):
            o.param_groups[0]["lr"] = 5e-4 * (it + 1)
        o_ref.step()
        o_mine.step()
    for i, (r, m) in enumerate(zip(ref, mine)):
        assert_close(m.detach(), r.detach(), 2e-6, f"AdamW param {i} {tuple(r.shape)}")
        assert_close(o_mine.state[m]["exp_avg"], o_ref.state[r]["exp_avg"], 2e-6, f"AdamW exp_avg {i}")
        assert_close(o_mine.state[m]["exp_avg_sq"], o_ref.state[r]["exp_avg_sq"], 2e-6, f"AdamW exp_avg_sq {i}")
        assert float(o_mine.state[m]["step"]) == float(o_ref.state[r]["step"])
    # state dicts are interchangeable: torch's AdamW continues from ours and the other way round
    o_ref2 = torch.optim.AdamW([r.detach().clone().requires_grad_() for r in ref], foreach=False, **kw)
    sd = o_mine.state_dict()
    sd["state"] = {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in sd["state"].items()}
    o_ref2.load_state_dict(sd)
    o_mine2 = FusedAdamW([m.detach().clone().requires_grad_() for m in mine], **kw)
    o_mine2.load_state_dict(o_ref.state_dict())
    for o, ps in ((o_ref2, o_ref2.param_groups[0]["params"]), (o_mine2, o_mine2.param_groups[0]["params"])):
        for p_ in ps:
            p_.grad = torch.ones_like(p_)
        o.step()
    for a, b in zip(o_ref2.param_groups[0]["params"], o_mine2.param_groups[0]["params"]):
        assert_close(b.detach(), a.detach(), 2e-6, "AdamW after exchanging state dicts")


def check_block_chain(dev, dtype, B=2, T=70, d=64):
    """MixingBlock chained on (stream, pending branch) -- the last residual add of a block inside the next block's norm1
    pass (K10 with a residual), the way AttentiveGLA.forward runs a stack -- against the plain loop over the same blocks:
    same output, same input / parameter gradients; and the chain really defers (a pending branch comes back)."""
    import copy
    import torch.nn as nn
    from lina_speech_amd.blocks import MixingBlock, SwiGLU
    torch.manual_seed(5)
    blocks = nn.ModuleList([MixingBlock(lambda: nn.Linear(d, d), lambda: SwiGLU(d), lambda: nn.LayerNorm(d))
                            for _ in range(3)]).to(dtype).to(dev)
    plain = copy.deepcopy(blocks)
    g = torch.Generator().manual_seed(6)
    x0 = torch.randn(B, T, d, generator=g).to(dtype)
    v0 = torch.randn(B, T, d, generator=g).to(dtype)
    dy = torch.randn(B, T, d, generator=g).to(dtype).to(dev)
    xa, va = x0.to(dev).requires_grad_(), v0.to(dev).requires_grad_()
    x, pend = xa, va                                    # a branch handed in from outside (the cross-attention output)
    for blk in blocks:
        x, pend = blk(x, _pending=pend, _defer=True)
        assert pend is not None and blk.can_defer(x)
    ya = x + pend
    (ya.float() * dy.float()).sum().backward()
    xb, vb = x0.to(dev).requires_grad_(), v0.to(dev).requires_grad_()
    x = xb + vb
    for blk in plain:
        x = blk(x)
    (x.float() * dy.float()).sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert_close(ya, x, tol, "chain y")
    assert_close(xa.grad, xb.grad, tol, "chain dx")
    assert_close(va.grad, vb.grad, tol, "chain dpending")
    for (n, pa), (_, pb) in zip(blocks.named_parameters(), plain.named_parameters()):
        assert_close(pa.grad, pb.grad, tol, "chain d" + n)


def check_cross_entropy(dev, N, V, dtype, ld=None):
    """K14 vs F.cross_entropy in fp64 on the same (dtype-rounded) logits: the mean loss over the rows that count and the
    gradient of the logits, with ignored rows, an odd row width (rows start on any 2-byte boundary) and a row stride."""
    g = torch.Generator().manual_seed(53)
    ld = ld or V
    buf = (torch.randn(N, ld, generator=g) * 4).to(dtype)
    tgt = torch.randint(0, V, (N,), generator=g)
    if N >= 5:
        tgt[::5] = 1                                          # the ignored class
        tgt[3] = V - 1
        tgt[4] = 0
    else:                                                     # tiny cases: every row counts
        tgt[tgt == 1] = 0
        tgt[-1] = V - 1
    l64 = buf[:, :V].to(F64).requires_grad_()
    ref = F.cross_entropy(l64, tgt, ignore_index=1)
    ref.backward()
    bd = buf.to(dev)
    lg = bd[:, :V].requires_grad_() if ld != V else bd.requires_grad_()
    loss = ops.cross_entropy(lg, tgt.to(dev), ignore_index=1)
    assert type(loss.grad_fn).__name__ == "_CrossEntropyFunctionBackward", type(loss.grad_fn).__name__
    lo = dtype == torch.bfloat16
    assert_close(loss, ref.detach(), 1e-5, "K14 loss")        # fp32 arithmetic on the stored values either way
    (loss * 3.0).backward()
    assert lg.grad.dtype == dtype
    assert_close(lg.grad, l64.grad * 3.0, 1e-2 if lo else 1e-5, "K14 dlogits")
    if N >= 5:
        assert torch.count_nonzero(lg.grad[::5]) == 0
    # all rows ignored: nan, like torch
    n4 = min(4, N)
    none = ops.cross_entropy(bd[:n4, :V].contiguous(), torch.ones(n4, dtype=torch.int64, device=dev), ignore_index=1)
    assert torch.isnan(none)


def check_sum_partials(dev):
    """K13 against torch's sum over the partial-row axis: odd numbers of partial rows (more and fewer than the 16 waves x 16
    rows of one round), a width that is not a multiple of 256, the outer axis, a bf16 result; widths the kernel does not
    take (N % 4 != 0) fall back to torch."""
    g = torch.Generator().manual_seed(61)
    # (the launcher narrows the workgroup to 32 / 16 / 8 lanes along the columns when there are few of them, and picks 4 / 8 /
    # 16 rows per round from the row count: the shapes below reach every combination incl. the train step's own)
    for P, shape in ((1, (8,)), (7, (40, 5)), (37, (256,)), (300, (1024, 5)), (513, (260,)), (16, (3,)), (1024, (256,)),
                     (130, (2816,)), (64, (3072, 5)), (256, (1024, 17)), (1025, (1024,))):
        part = torch.randn(P, *shape, generator=g).to(dev)
        got = ops._sum_partials(part)
        assert got.shape == shape and got.dtype == torch.float32
        assert_close(got, part.double().sum(0), 1e-6, f"K13 P={P} {shape}")
    part = torch.randn(2, 77, 512, generator=g).to(dev)
    assert_close(ops._sum_partials2(part), part.double().sum(1), 1e-6, "K13 outer")
    gb = ops._sum_partials(part[0], torch.bfloat16)
    assert gb.dtype == torch.bfloat16
    assert_close(gb, part[0].double().sum(0), 1e-2, "K13 bf16 out")


def check_column_sum(dev):
    """K13a + K13 (ops.column_sum, the bias gradient of ops.linear) against the fp64 column sum: row counts around the
    128-row slabs, widths that are not multiples of 256, a row stride, both dtypes; and the long-vector sum of K14's rows."""
    g = torch.Generator().manual_seed(67)
    for M, N, ld, dtype in ((1, 4, 4, torch.float32), (127, 40, 40, torch.bfloat16), (129, 260, 264, torch.float32),
                            (1000, 1024, 1024, torch.bfloat16), (300, 16, 4112, torch.bfloat16)):
        buf = torch.randn(M, ld, generator=g).to(dtype).to(dev)
        x = buf[:, :N]
        got = ops.column_sum(x)
        assert got.dtype == torch.float32 and got.shape == (N,)
        assert_close(got, x.double().sum(0), 1e-5, f"K13a {M}x{N}")
    for n in (4096, 1028, 7):
        v = torch.randn(n, generator=g).to(dev)
        assert_close(ops._sum_vector(v), v.double().sum(), 1e-5, f"vector sum n={n}")


def check_argmax(dev, rows, n, dtype):
    g = torch.Generator().manual_seed(5)
    lg = torch.randn(rows, n, generator=g).to(dtype)
    lg[1, 100] = lg[1, 3000] = 50.0        # exact tie -> lowest index
    lg[2, n - 1] = 60.0                    # max in the last column
    lg[3, 0] = 60.0
    got = ops.argmax_rows(lg.to(dev))
    ref = O.argmax_lowest(lg.float())
    assert got.dtype == torch.int64
    assert torch.equal(got.cpu(), ref), (got, ref)
    assert got[1].item() == 100 and got[2].item() == n - 1 and got[3].item() == 0
    # 3-D input keeps its leading shape
    got3 = ops.argmax_rows(lg.view(rows, 1, n).to(dev))
    assert got3.shape == (rows, 1) and torch.equal(got3.cpu().view(-1), ref)


def check_topk_sample(dev, rows, n, k, temp, dtype, seed=3, draws=4000):
    """K6c vs the fp64 inverse-CDF oracle: token ids EXACT wherever the uniform number sits further than 1e-5 from
    a CDF edge (the kernel's running sums are fp32); both the caller-supplied-uniform and the hashed-uniform forms;
    the picked token is always one of the top-k; and the empirical distribution over many draws follows p."""
    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(rows, n, generator=g) * 3).to(dtype)
    u = torch.rand(rows, generator=g)
    tok = ops.topk_sample_rows(logits.to(dev), k, temp, u=u.to(dev)).cpu()
    rt, margin, p = O.topk_sample_inverse_cdf(logits.float(), k, temp, u)
    ok = margin > 1e-5
    assert ok.float().mean() > 0.8
    assert torch.equal(tok[ok], rt[ok]), f"{int((tok[ok] != rt[ok]).sum())} sampled tokens differ"
    assert bool((p.gather(-1, tok.unsqueeze(-1)) > 0).all()), "picked a token outside the top-k set"
    # hashed uniforms: (seed, step, row)
    step = torch.tensor([5], dtype=torch.int64)
    tok2 = ops.topk_sample_rows(logits.to(dev), k, temp, seed=1234, step=step.to(dev)).cpu()
    u2 = torch.tensor([O.hash_uniform(1234, 5, r, rows) for r in range(rows)], dtype=torch.float64)
    rt2, margin2, _ = O.topk_sample_inverse_cdf(logits.float(), k, temp, u2)
    ok2 = margin2 > 1e-5
    assert torch.equal(tok2[ok2], rt2[ok2])
    assert 0.0 <= float(u2.min()) and float(u2.max()) < 1.0
    # distribution: one logits row, many uniforms -> chi-square-like bound on the top entries
    row = logits[:1].expand(draws, n).contiguous()
    uu = (torch.arange(draws, dtype=torch.float32) + 0.5) / draws          # stratified uniforms
    tk = ops.topk_sample_rows(row.to(dev), k, temp, u=uu.to(dev)).cpu()
    freq = torch.bincount(tk, minlength=n).double() / draws
    assert (freq - p[0]).abs().max() < 2.0 / draws + 1e-6, "sampling frequencies do not follow softmax(top-k)"


def check_swiglu(dev, rows, hidden, dtype):
    g = torch.Generator().manual_seed(6)
    u = torch.randn(rows, 2 * hidden, generator=g).to(dtype).to(dev)
    ref = F.silu(u.cpu().to(F64)[:, :hidden]) * u.cpu().to(F64)[:, hidden:]
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    y = ops.swiglu(u, hidden)
    assert_close(y, ref, tol, "swiglu")
    pad = hidden + 3
    y = ops.swiglu(u, hidden, pad_to=pad)
    assert y.shape == (rows, pad)
    assert_close(y[:, :hidden], ref, tol, "swiglu padded")
    assert torch.equal(y[:, hidden].float().cpu(), torch.ones(rows)) and (y[:, hidden + 1:] == 0).all()


def check_prologue(dev, B, Kd, Vd, dtype, R=16, W=4, clamp_min=None):
    g = torch.Generator().manual_seed(7)
    ldz = 2 * Kd + 2 * Vd + R + 5                      # q | k | v | g | low-rank (+ slack)
    off_q, off_k, off_v, off_lr = 0, Kd, 2 * Kd, 2 * Kd + 2 * Vd
    z = torch.randn(B, ldz, generator=g).to(dtype).to(dev)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dtype).to(dev)
    wq, wk, wv = mk(Kd, W), mk(Kd, W), mk(Vd, W)
    cq, ck, cv = mk(B, Kd, W), mk(B, Kd, W), mk(B, Vd, W)
    w2, b2 = mk(Kd, R) * 4, mk(Kd)
    qkv = torch.empty(B, 2 * Kd + Vd, dtype=dtype, device=dev)
    gk = torch.empty(B, Kd, dtype=torch.float32, device=dev)
    rc = [c.cpu().to(F64).clone() for c in (cq, ck, cv)]
    zc = z.cpu().to(F64)
    ry = [O.short_conv(zc[:, None, o:o + D], w.cpu().to(F64), None, c)
          for o, D, w, c in ((off_q, Kd, wq, rc[0]), (off_k, Kd, wk, rc[1]), (off_v, Vd, wv, rc[2]))]
    rgk = O.gate_logsigmoid(zc[:, off_lr:off_lr + R] @ w2.cpu().to(F64).t() + b2.cpu().to(F64), 16.0, clamp_min)
    ops.gla_decode_prologue(z, off_q, off_k, off_v, off_lr, wq, wk, wv, cq, ck, cv, w2, b2, qkv, gk, 16.0, clamp_min)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert_close(qkv, torch.cat([r[:, 0] for r in ry], dim=1), tol, "prologue qkv")
    assert_close(gk, rgk, 1e-5 if dtype == torch.float32 else 2e-2, "prologue gk")
    for c, r, nm in zip((cq, ck, cv), rc, "qkv"):
        assert_close(c, r, 1e-6 if dtype == torch.float32 else 1e-2, f"prologue cache {nm}")


def check_decode_update(dev, B, H, Dk, Dv, dtype):
    """K1d (row-split, partial o) == one oracle step; partials are added by K5's n_partial path."""
    q, k, v, gk, h0 = make_gla_inputs(B, H, 1, Dk, Dv, dtype, dev, seed=8)
    ro, rS = oracle_gla(q, k, v, gk, h0)
    S = h0.clone()
    o_part = torch.empty(Dk // 64, B, H, Dv, dtype=torch.float32, device=dev)
    ops.gla_decode_update(q[:, :, 0], k[:, :, 0], v[:, :, 0], gk[:, :, 0].float(), o_part, S)
    assert_close(S, rS, 1e-5 if dtype == torch.float32 else 1e-2, "K1d state")
    assert_close(o_part.sum(0), ro[:, :, 0], 1e-5 if dtype == torch.float32 else 1e-2, "K1d sum of partials")
    # bit-identical state to the V-split kernel K1 (same fma order per element)
    _, S1 = ops.fused_recurrent_gla(q, k, v, gk.float(), initial_state=h0, output_final_state=True)
    assert torch.equal(S, S1)


def check_linear_skinny(dev, M, N, K, dtype, ln=False, bias=False, resid=False, swiglu=0):
    g = torch.Generator().manual_seed(9)
    kq = 32 if dtype == torch.bfloat16 else 16
    assert K % kq == 0
    a = (torch.randn(M, K, generator=g) * 1.5 + (0.7 if ln else 0.0)).to(dtype).to(dev)
    n_w = 2 * swiglu if swiglu else N
    w = (torch.randn(n_w, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    b = torch.randn(n_w, generator=g).to(dev) if bias else None
    r = torch.randn(M, N, generator=g).to(dtype).to(dev) if resid else None
    a64, w64 = a.cpu().to(F64), w.cpu().to(F64)
    c1 = c2 = None
    if ln:
        gamma = (1 + 0.2 * torch.randn(K, generator=g)).to(dev)
        beta = (0.3 * torch.randn(K, generator=g)).to(dev)
        w_ln = (w.float() * gamma[None, :]).to(dtype).contiguous()
        c1 = w_ln.float().sum(1).contiguous()
        c2 = (w.float() @ beta).contiguous()
        if bias:
            c2 = c2 + b
        # reference: true LayerNorm in fp64, projected with the SAME rounded folded weight
        mu = a64.mean(1, keepdim=True)
        var = a64.var(1, unbiased=False, keepdim=True)
        ref = ((a64 - mu) / torch.sqrt(var + 1e-5)) @ w_ln.cpu().to(F64).t() + c2.cpu().to(F64)
        w_used = w_ln
    else:
        c2 = b
        ref = a64 @ w64.t() + (0 if b is None else b.cpu().to(F64))
        w_used = w
    n_out = N
    if swiglu:
        ga, gb = ref[:, :swiglu], ref[:, swiglu:]
        ref = torch.zeros(M, N, dtype=F64)
        ref[:, :swiglu] = F.silu(ga) * gb
        ref[:, swiglu] = 1.0
    if resid:
        ref = ref + r.cpu().to(F64)
    out = ops.linear_skinny(a, w_used, c1, None if c2 is None else c2.float().contiguous(), resid=r,
                            swiglu_hidden=swiglu, ln_dim=K if ln else 0, n_out=n_out)
    assert out.shape == (M, N) and out.dtype == dtype
    assert_close(out, ref, 2e-5 if dtype == torch.float32 else 2e-2, f"linear_skinny M{M} N{N} K{K}")
    if resid:   # in-place residual form (out aliases resid)
        r2 = r.clone()
        ops.linear_skinny(a, w_used, c1, None if c2 is None else c2.float().contiguous(), resid=r2, out=r2,
                          swiglu_hidden=swiglu, ln_dim=K if ln else 0, n_out=n_out)
        assert torch.equal(r2, out)


def check_linear_skinny_packed(dev, M, N, K, dtype, ln=False, bias=False, resid=False, swiglu=0):
    """Fragment-major operands (lina_linear_skinny_ex): the SAME arithmetic as the row-major call, so the result must be
    BIT-identical to lina_linear_skinny on the same values -- row-major output and the packed copy (which is what the
    next projection consumes)."""
    g = torch.Generator().manual_seed(19)
    kq = 32 if dtype == torch.bfloat16 else 16
    a = (torch.randn(M, K, generator=g) * 1.5 + (0.7 if ln else 0.0)).to(dtype).to(dev)
    n_w = 2 * swiglu if swiglu else N
    w = (torch.randn(n_w, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    c1 = (torch.randn(n_w, generator=g)).to(dev) if ln else None
    c2 = torch.randn(n_w, generator=g).to(dev) if (bias or ln) else None
    r = torch.randn(M, N, generator=g).to(dtype).to(dev) if resid else None
    ref = ops.linear_skinny(a, w, c1, c2, resid=r, swiglu_hidden=swiglu, ln_dim=K if ln else 0, n_out=N)
    a_p = ops.pack_rows(a)
    if swiglu:       # each half zero-padded to whole 64-row blocks covering the N output columns (swiglu % 64 == 0)
        half = (N + 63) // 64 * 64
        pad = lambda h: torch.cat([h, torch.zeros(half - h.shape[0], K, dtype=dtype, device=dev)])
        w_p = torch.cat([ops.pack_rows(pad(w[:swiglu])), ops.pack_rows(pad(w[swiglu:]))])
    else:
        half = None
        w_p = ops.pack_rows(w)
    Np = (N + kq - 1) // kq * kq
    out = torch.full((M, N), float("nan"), dtype=dtype, device=dev)
    out_p = torch.zeros(ops.packed_numel(M, Np), dtype=dtype, device=dev)
    ops.linear_skinny_packed(a_p, w_p, M, N, K, c1, c2, resid=r, out=out, out_packed=out_p, out_packed_width=Np,
                             swiglu_hidden=swiglu, ln_dim=K if ln else 0, w_half_rows=half)
    # same split-K width (4 waves) in both kernels: the SAME arithmetic.  A wider split (LINA_SKINNY_WAVES / the decode
    # shapes' default, packed kernels only) sums the k-steps in another order: equal to fp32 rounding of the partial sums
    same_order = _skinny_waves(K, kq) == 4
    tol = 0.0 if same_order else (1.6e-2 if dtype == torch.bfloat16 else 2e-5)

    def same(a_, b_, what):
        if same_order:
            assert torch.equal(a_, b_), what
        else:
            assert_close(a_, b_.double().cpu(), tol, what)
    same(out, ref, "packed operands changed the result")
    assert torch.equal(ops.unpack_rows(out_p, M, Np)[:, :N], out), "packed output copy differs"
    if resid:   # the residual stream held ONLY in packed form, updated in place (resid is out_packed)
        rp = torch.zeros(M, Np, dtype=dtype, device=dev)
        rp[:, :N] = r
        x_p = ops.pack_rows(rp)
        ops.linear_skinny_packed(a_p, w_p, M, N, K, c1, c2, resid=x_p, out_packed=x_p, out_packed_width=Np,
                                 swiglu_hidden=swiglu, ln_dim=K if ln else 0, w_half_rows=half)
        assert torch.equal(ops.unpack_rows(x_p, M, Np)[:, :N], out), "in-place packed residual update differs"
    # row-major inputs + packed output copy
    out_p2 = torch.zeros_like(out_p)
    ops.linear_skinny(a, w, c1, c2, resid=r, swiglu_hidden=swiglu, ln_dim=K if ln else 0, n_out=N, out_packed=out_p2,
                      out_packed_width=Np)
    same(ops.unpack_rows(out_p2, M, Np)[:, :N], out, "row-major inputs + packed output copy")


def check_linear_tall(dev, M, N, K, dtype, ln=False, bias=False, resid=False, swiglu=0, force=True, variant=None):
    """The tall tiling of the packed projection (linear_tall.h: 128 rows x 64 weight rows per workgroup, weights staged through
    LDS by DMA, no split-K) against fp64 of the same bf16 / fp32 operands -- LayerNorm fold, bias, residual (row-major and
    in-place packed), SwiGLU with the constant-1 bias column, row-major and packed outputs, ragged M / N / K-stage counts --
    and against the skinny kernel (another summation order: fp32 rounding of the partial sums).  ``force``: LINA_TALL=1 (the
    launcher's own rule picks the tall kernel only for M >= 128 and wide outputs)."""
    import os
    g = torch.Generator().manual_seed(23)
    kq = 32 if dtype == torch.bfloat16 else 16
    a = (torch.randn(M, K, generator=g) * 1.5 + (0.7 if ln else 0.0)).to(dtype).to(dev)
    n_w = 2 * swiglu if swiglu else N
    w = (torch.randn(n_w, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    c1 = (torch.randn(n_w, generator=g)).to(dev) if ln else None
    c2 = torch.randn(n_w, generator=g).to(dev) if (bias or ln) else None
    r = torch.randn(M, N, generator=g).to(dtype).to(dev) if resid else None
    a_p = ops.pack_rows(a)
    if swiglu:
        half = (N + 63) // 64 * 64
        pad = lambda h: torch.cat([h, torch.zeros(half - h.shape[0], K, dtype=dtype, device=dev)])
        w_p = torch.cat([ops.pack_rows(pad(w[:swiglu])), ops.pack_rows(pad(w[swiglu:]))])
    else:
        half = None
        w_p = ops.pack_rows(w)
    Np = (N + kq - 1) // kq * kq
    kw = dict(swiglu_hidden=swiglu, ln_dim=K if ln else 0, w_half_rows=half)
    prev, prev_v = os.environ.get("LINA_TALL"), os.environ.get("LINA_TALL_V")
    try:
        if variant is not None:                     # 0 = LDS ring, 1 = register ring (linear_tall.h); None = the launcher's default
            os.environ["LINA_TALL_V"] = str(variant)
        os.environ["LINA_TALL"] = "0"
        ref_sk = torch.empty(M, N, dtype=dtype, device=dev)
        ops.linear_skinny_packed(a_p, w_p, M, N, K, c1, c2, resid=r, out=ref_sk, **kw)
        if force:
            os.environ["LINA_TALL"] = "1"
        else:
            os.environ.pop("LINA_TALL")
        out = torch.full((M, N), float("nan"), dtype=dtype, device=dev)
        out_p = torch.zeros(ops.packed_numel(M, Np), dtype=dtype, device=dev)
        ops.linear_skinny_packed(a_p, w_p, M, N, K, c1, c2, resid=r, out=out, out_packed=out_p, out_packed_width=Np, **kw)
        if resid:
            rp = torch.zeros(M, Np, dtype=dtype, device=dev)
            rp[:, :N] = r
            x_p = ops.pack_rows(rp)
            ops.linear_skinny_packed(a_p, w_p, M, N, K, c1, c2, resid=x_p, out_packed=x_p, out_packed_width=Np, **kw)
            assert torch.equal(ops.unpack_rows(x_p, M, Np)[:, :N], out), "tall: in-place packed residual update differs"
    finally:
        for name, old in (("LINA_TALL", prev), ("LINA_TALL_V", prev_v)):
            if old is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = old
    # fp64 of the same operands
    a64, w64 = a.cpu().to(F64), w.cpu().to(F64)
    if ln:
        mu = a64.mean(-1, keepdim=True)
        var = (a64 * a64).mean(-1, keepdim=True) - mu * mu
        z = (a64 @ w64.t() - mu * c1.cpu().to(F64)[None]) * torch.rsqrt(var.clamp_min(0) + 1e-5) + c2.cpu().to(F64)[None]
    else:
        z = a64 @ w64.t() + (c2.cpu().to(F64)[None] if c2 is not None else 0.0)
    if swiglu:
        y = torch.zeros(M, N, dtype=F64)
        y[:, :swiglu] = torch.nn.functional.silu(z[:, :swiglu]) * z[:, swiglu:]
        if N > swiglu:
            y[:, swiglu] = 1.0
    else:
        y = z
    if resid:
        y = y + r.cpu().to(F64)
    tol = 1.6e-2 if dtype == torch.bfloat16 else 2e-5
    assert_close(out, y, tol, "tall projection vs fp64")
    assert_close(out, ref_sk.double().cpu(), tol, "tall projection vs the skinny kernel")
    assert torch.equal(ops.unpack_rows(out_p, M, Np)[:, :N], out), "tall: packed output copy differs"


def check_inproj_tall(dev, B, K, Kd, Vd, dtype, force=True, variant=None, same_as_variant=None):
    """The tall tiling of the fused input side of a mixer (gla_inproj_tall_kernel: B >= 128) against the 64-row kernel on the
    same operands: q | k | v (conv step + SiLU), g, the gate (rank-16 up-projection + log-sigmoid) and the rolled conv caches
    -- another summation order of the same products (no split-K), so equal to fp32 rounding of the partial sums.  The 64-row
    kernel itself is checked against fp64 in check_inproj."""
    import os
    g = torch.Generator().manual_seed(31)
    R, W = 16, 4
    x = (torch.randn(B, K, generator=g) + 0.3).to(dtype).to(dev)
    w_in = (torch.randn(2 * Kd + 2 * Vd + R, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    c1 = w_in.float().sum(1).contiguous()
    c2 = torch.randn(w_in.shape[0], generator=g).to(dev)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dtype).to(dev)
    wq, wk, wv, w2, b2 = mk(Kd, W), mk(Kd, W), mk(Vd, W), mk(Kd, R), mk(Kd)
    caches = [mk(B, Kd, W), mk(B, Kd, W), mk(B, Vd, W)]
    x_p, w_p = ops.pack_rows(x), ops.pack_rows(w_in)
    outs = []
    prev, prev_v = os.environ.get("LINA_TALL"), os.environ.get("LINA_TALL_V")
    try:
        if variant is not None:
            os.environ["LINA_TALL_V"] = str(variant)
        for tall in (False, True):
            if tall and not force:
                os.environ.pop("LINA_TALL", None)
            else:
                os.environ["LINA_TALL"] = "1" if tall else "0"
            cq, ck, cv = (c.clone() for c in caches)
            qkv = torch.full((B, 2 * Kd + Vd), float("nan"), dtype=dtype, device=dev)
            go = torch.full((B, Vd), float("nan"), dtype=dtype, device=dev)
            gk = torch.full((B, Kd), float("nan"), dtype=torch.float32, device=dev)
            ops.gla_decode_inproj_packed(x_p, w_p, B, K, c1, c2, wq, wk, wv, cq, ck, cv, w2, b2, qkv, go, gk,
                                         clamp_min=-0.03 if Kd == 128 else None)
            outs.append((qkv, go, gk, cq, ck, cv))
    finally:
        for name, old in (("LINA_TALL", prev), ("LINA_TALL_V", prev_v)):
            if old is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = old
    for name, a, b in zip(("qkv", "g", "gk", "cq", "ck", "cv"), *outs):
        assert_close(b, a.double().cpu(), 1.6e-2 if a.dtype == torch.bfloat16 else 5e-5, f"tall in-projection: {name}")
    # the cache roll itself is exact: three old taps move up unchanged
    for a, c0 in zip(outs[1][3:], caches):
        assert torch.equal(a[..., :3], c0[..., 1:]), "tall in-projection: the rolled cache taps changed"
    if same_as_variant is not None:
        # two variants of the tall launch that add the same products in the same order (e.g. the 128-row kernel with the gate
        # folded in, variant 3, against the 64-row kernel with gate workgroups, variant 0): bit-identical outputs
        try:
            os.environ["LINA_TALL_V"], os.environ["LINA_TALL"] = str(same_as_variant), "1"
            cq, ck, cv = (c.clone() for c in caches)
            qkv, go = torch.empty(B, 2 * Kd + Vd, dtype=dtype, device=dev), torch.empty(B, Vd, dtype=dtype, device=dev)
            gk = torch.empty(B, Kd, dtype=torch.float32, device=dev)
            ops.gla_decode_inproj_packed(x_p, w_p, B, K, c1, c2, wq, wk, wv, cq, ck, cv, w2, b2, qkv, go, gk,
                                         clamp_min=-0.03 if Kd == 128 else None)
        finally:
            for name, old in (("LINA_TALL", prev), ("LINA_TALL_V", prev_v)):
                if old is None:
                    os.environ.pop(name, None)
                else:
                    os.environ[name] = old
        for name, a, b in zip(("qkv", "g", "gk", "cq", "ck", "cv"), outs[1], (qkv, go, gk, cq, ck, cv)):
            assert torch.equal(a, b), f"tall in-projection: {name} of variant {variant} != variant {same_as_variant}"


def _skinny_waves(K, kq):
    """Split-K width the packed projection kernels pick for this K (mirrors linear_skinny_impl / inproj_impl)."""
    import os
    nw = int(os.environ.get("LINA_SKINNY_WAVES", "16") or 16)
    while nw > 4 and K // kq < 2 * nw:
        nw //= 2
    return nw if nw in (8, 16) else 4


def check_inproj_packed(dev, B, K, Kd, Vd, dtype):
    """lina_gla_decode_inproj_packed == lina_gla_decode_inproj bit for bit (outputs AND the rolled conv caches)."""
    g = torch.Generator().manual_seed(23)
    R, W = 16, 4
    x = (torch.randn(B, K, generator=g) + 0.3).to(dtype).to(dev)
    w_in = (torch.randn(2 * Kd + 2 * Vd + R, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    c1 = w_in.float().sum(1).contiguous()
    c2 = torch.randn(w_in.shape[0], generator=g).to(dev)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dtype).to(dev)
    wq, wk, wv, w2, b2 = mk(Kd, W), mk(Kd, W), mk(Vd, W), mk(Kd, R), mk(Kd)
    caches = [mk(B, Kd, W), mk(B, Kd, W), mk(B, Vd, W)]
    outs = []
    for packed in (False, True):
        cq, ck, cv = (c.clone() for c in caches)
        qkv = torch.empty(B, 2 * Kd + Vd, dtype=dtype, device=dev)
        go = torch.empty(B, Vd, dtype=dtype, device=dev)
        gk = torch.empty(B, Kd, dtype=torch.float32, device=dev)
        if packed:
            ops.gla_decode_inproj_packed(ops.pack_rows(x), ops.pack_rows(w_in), B, K, c1, c2, wq, wk, wv, cq, ck, cv, w2, b2,
                                         qkv, go, gk)
        else:
            ops.gla_decode_inproj(x, w_in, c1, c2, wq, wk, wv, cq, ck, cv, w2, b2, qkv, go, gk)
        outs.append((qkv, go, gk, cq, ck, cv))
    kq = 32 if dtype == torch.bfloat16 else 16
    for name, a, b in zip(("qkv", "g", "gk", "cq", "ck", "cv"), *outs):
        if _skinny_waves(K, kq) == 4:
            assert torch.equal(a, b), f"packed in-projection: {name} differs"
        else:        # wider split-K in the packed kernel: another summation order of the same products
            assert_close(a, b.double().cpu(), 1.6e-2 if a.dtype == torch.bfloat16 else 5e-5, f"packed in-projection: {name}")


def check_inproj(dev, B, K, Kd, Vd, dtype):
    """lina_gla_decode_inproj == lina_linear_skinny(LayerNorm fold) + lina_gla_decode_prologue (both oracle-checked
    above), and == the oracle's LayerNorm -> projections -> conv step -> gate in fp64."""
    g = torch.Generator().manual_seed(11)
    R, W = 16, 4
    x = (torch.randn(B, K, generator=g) * 1.3 + 0.4).to(dtype).to(dev)
    n_w = 2 * Kd + 2 * Vd + R
    w = (torch.randn(n_w, K, generator=g) / K ** 0.5).to(dtype).to(dev)
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).to(dev)
    beta = (0.3 * torch.randn(K, generator=g)).to(dev)
    w_ln = (w.float() * gamma[None, :]).to(dtype).contiguous()
    c1 = w_ln.float().sum(1).contiguous()
    c2 = (w.float() @ beta).contiguous()
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dtype).to(dev)
    wq, wk, wv = mk(Kd, W), mk(Kd, W), mk(Vd, W)
    caches = [mk(B, Kd, W), mk(B, Kd, W), mk(B, Vd, W)]
    w2, b2 = mk(Kd, R) * 4, mk(Kd)
    # unfused product path
    c_a = [c.clone() for c in caches]
    z = ops.linear_skinny(x, w_ln, c1, c2, ln_dim=K)
    qkv_a = torch.empty(B, 2 * Kd + Vd, dtype=dtype, device=dev)
    gk_a = torch.empty(B, Kd, dtype=torch.float32, device=dev)
    ops.gla_decode_prologue(z, 0, Kd, 2 * Kd, 2 * Kd + 2 * Vd, wq, wk, wv, *c_a, w2, b2, qkv_a, gk_a)
    # fused
    c_b = [c.clone() for c in caches]
    qkv_b = torch.empty_like(qkv_a)
    g_b = torch.empty(B, Vd, dtype=dtype, device=dev)
    gk_b = torch.empty_like(gk_a)
    ops.gla_decode_inproj(x, w_ln, c1, c2, wq, wk, wv, *c_b, w2, b2, qkv_b, g_b, gk_b)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert_close(qkv_b, qkv_a, tol, "inproj qkv vs unfused")
    assert_close(g_b, z[:, 2 * Kd + Vd:2 * Kd + 2 * Vd], tol, "inproj g vs unfused")
    assert_close(gk_b, gk_a, 1e-4 if dtype == torch.float32 else 3e-2, "inproj gk vs unfused")
    for a, b_, nm in zip(c_a, c_b, "qkv"):
        assert_close(b_, a, 1e-5 if dtype == torch.float32 else 1e-2, f"inproj cache {nm}")
    # oracle (fp64) for the fp32 case
    if dtype == torch.float32:
        x64 = x.cpu().to(F64)
        ln = (x64 - x64.mean(1, keepdim=True)) / torch.sqrt(x64.var(1, unbiased=False, keepdim=True) + 1e-5)
        ln = ln * gamma.cpu().to(F64) + beta.cpu().to(F64)
        zz = ln @ w.cpu().to(F64).t()
        rc = caches[0].cpu().to(F64).clone()
        rq = O.short_conv(zz[:, None, :Kd], wq.cpu().to(F64), None, rc)
        assert_close(qkv_b[:, :Kd], rq[:, 0], 1e-4, "inproj q vs oracle")
        rgk = O.gate_logsigmoid(zz[:, 2 * Kd + 2 * Vd:] @ w2.cpu().to(F64).t() + b2.cpu().to(F64), 16.0)
        assert_close(gk_b, rgk, 1e-4, "inproj gk vs oracle")


def check_decode_update_norm(dev, B, H, Dk, Dv, dtype, repeats=1):
    """K1d+K5 fused (last-arriver hand-off between the Dk/64 row-block workgroups of a head) must be BIT-identical to
    K1d followed by K5(n_partial) -- repeated so that a stale/early read of a partial would show up."""
    g = torch.Generator().manual_seed(12)
    q, k, v, gk, h0 = make_gla_inputs(B, H, 1, Dk, Dv, dtype, dev, seed=13)
    q, k, v, gk = q[:, :, 0], k[:, :, 0], v[:, :, 0], gk[:, :, 0].float()
    ldz = H * Dv + 8
    zrow = torch.randn(B, ldz, generator=g).to(dtype).to(dev)
    gate = zrow[:, 4:4 + H * Dv].view(B, H, Dv)                      # strided view, like the projection row
    w = (1 + 0.1 * torch.randn(Dv, generator=g)).to(dtype).to(dev)
    NP = Dk // 64
    counters = torch.zeros(B * H, dtype=torch.int32, device=dev)
    S_a, S_b = h0.clone(), h0.clone()
    for it in range(repeats):
        op_a = torch.full((NP, B, H, Dv), float("nan"), device=dev)
        op_b = torch.full((NP, B, H, Dv), float("nan"), device=dev)
        ops.gla_decode_update(q, k, v, gk, op_a, S_a)
        og_a = ops.rmsnorm_swish_gate(op_a, gate, w, 1e-5, n_partial=NP, out_dtype=dtype)
        og_b = torch.full((B, H, Dv), float("nan"), dtype=dtype, device=dev)
        ops.gla_decode_update_norm(q, k, v, gk, op_b, S_b, gate, w, og_b, counters, 1e-5)
        assert torch.equal(S_a, S_b), f"state differs (iteration {it})"
        assert torch.equal(og_a.reshape(B, H, Dv), og_b), f"fused norm output differs (iteration {it})"
        assert int(counters.abs().sum()) == 0, "arrival counters must be left at zero"


def check_decode_window(dev, B, H, Dk, Dv, dtype, window=8, n_steps=19, resets=True, origin0=5, state_dtype=torch.float32):
    """K1w + K5 (windowed, lazily written state) over ``n_steps`` decode steps -- several full windows plus a partial
    one -- against the fp64 recurrence + norm-gate of the oracle at EVERY step, the flushed final state against the
    oracle's, and against the immediate kernel K1d+K5 (same inputs): outputs within the kernel tolerance, state 1e-5.
    Reset gates (-20, reference reset_val) sit inside and at the edge of a window.
    ``state_dtype=torch.bfloat16`` (opt-in, lina_gla_decode_window_s): the state tensor is bf16 -- the oracle's fp64 state is
    then ROUNDED to bf16 wherever the kernel writes it back (every ``window``-th step: window 1 = every step = what the
    reference does to the state of a bf16 model, model/gla.py:229-240 + Cache.update), the outputs still come from the
    unrounded updated state."""
    g = torch.Generator().manual_seed(14)
    bf_state = state_dtype == torch.bfloat16
    rnd = lambda S: S.to(torch.bfloat16).to(F64)
    h0 = (torch.randn(B, H, Dk, Dv, generator=g) * 0.5).to(state_dtype).float().to(dev)
    NP = Dk // 64
    w = (1 + 0.1 * torch.randn(Dv, generator=g)).to(dtype).to(dev)
    counters = torch.zeros(B * H, dtype=torch.int32, device=dev)
    counters_i = torch.zeros(B * H, dtype=torch.int32, device=dev)
    S_w, S_i = h0.clone().to(state_dtype), h0.clone()
    st_tol = 8e-3 if bf_state else 1e-5            # bf16: one ulp (2^-8) where the fp32 and the fp64 value round apart
    S_ref = h0.detach().cpu().to(F64)
    hk = torch.full((window, B * H, Dk), float("nan"), device=dev)
    hc = torch.full((window, B * H, Dk), float("nan"), device=dev)
    hv = torch.full((window, B * H, Dv), float("nan"), device=dev)
    step = torch.full((1,), origin0, dtype=torch.int64, device=dev)
    origin = torch.full((1,), origin0, dtype=torch.int64, device=dev)
    o_x = torch.full((B * H * Dv,), float("nan"), device=dev) if Dv > 256 else None   # Dv = 512: column halves meet here
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    for t in range(n_steps):
        q = torch.randn(B, H, Dk, generator=g).to(dtype).to(dev)
        k = torch.randn(B, H, Dk, generator=g).to(dtype).to(dev)
        v = torch.randn(B, H, Dv, generator=g).to(dtype).to(dev)
        gk = (F.logsigmoid(torch.randn(B, H, Dk, generator=g) * 2.0) / 4.0)
        if resets and t in (3, 7, 8, 12):
            gk[:, :, ::2] = -20.0
        gk = gk.to(dev)
        gate = torch.randn(B, H, Dv, generator=g).to(dtype).to(dev)
        og = torch.full((B, H, Dv), float("nan"), dtype=dtype, device=dev)
        ops.gla_decode_window(q, k, v, gk, S_w, gate, w, og, hk, hc, hv, step, origin, window, 1e-5,
                              o_exchange=o_x, counters=counters)
        assert int(counters.abs().sum()) == 0
        step += 1
        if Dv <= 256 and not bf_state:
            op_i = torch.empty(NP, B, H, Dv, device=dev)
            og_i = torch.empty(B, H, Dv, dtype=dtype, device=dev)
            ops.gla_decode_update_norm(q, k, v, gk, op_i, S_i, gate, w, og_i, counters_i, 1e-5)
        # oracle: one recurrence step + norm-gate in fp64
        qd, kd, vd, gd = (x.cpu().to(F64) for x in (q, k, v, gk))
        S_ref = S_ref * gd.exp().unsqueeze(-1) + kd.unsqueeze(-1) * vd.unsqueeze(-2)
        o_ref = torch.einsum("bhk,bhkv->bhv", qd * Dk ** -0.5, S_ref)
        og_ref = O.rmsnorm_swish_gate(o_ref, gate.cpu().to(F64), w.cpu().to(F64), 1e-5)
        assert_close(og, og_ref, tol, f"K1w og (step {t}, window position {t % window})")
        if Dv <= 256 and not bf_state:
            assert_close(og.float(), og_i.float(), tol, f"K1w vs K1d og (step {t})")
        if (t + 1) % window == 0:                 # a completed window leaves the state fully written back
            if bf_state:
                S_ref = rnd(S_ref)
            assert_close(S_w, S_ref, st_tol, f"K1w state after window (step {t})")
    pending = n_steps % window
    ops.gla_decode_window_flush(S_w, hk, hc, hv, pending)
    if bf_state and pending:
        S_ref = rnd(S_ref)
    assert_close(S_w, S_ref, st_tol, "K1w flushed state")
    if Dv <= 256 and not bf_state:
        assert_close(S_w, S_i, 1e-5, "K1w flushed state vs K1d state")


def check_cross_att(dev, B, Tn, d, dtype):
    """lina_cross_att_step1/2 vs the eager attention of reference crossatt.py:13-19,114,143,149 in fp64."""
    g = torch.Generator().manual_seed(14)
    mk = lambda *s: torch.randn(*s, generator=g).to(dtype).to(dev)
    q_lin, kk, vv, pe = mk(B, d), mk(B, Tn, d), mk(B, Tn, d), mk(Tn, d)
    ln_w, ln_b = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype).to(dev), (0.1 * torch.randn(d, generator=g)).to(dtype).to(dev)
    x = mk(B, d)
    scale = d ** -0.5
    att = torch.zeros(B, 2, 1, Tn, dtype=dtype, device=dev)
    xp = torch.empty(B, d, dtype=dtype, device=dev)
    ops.cross_att_step1(q_lin, ln_w, ln_b, 1e-5, kk, pe, att[:, 0, 0], xp, scale)
    c = lambda t: t.cpu().to(F64)
    q = F.layer_norm(c(q_lin), (d,), c(ln_w), c(ln_b), 1e-5)
    a1 = torch.softmax(torch.einsum("bd,btd->bt", q, c(kk)) * scale, -1)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert_close(att[:, 0, 0], a1, tol, "cross-att att1")
    assert_close(xp, a1 @ c(pe), tol, "cross-att xp")
    xp2 = mk(B, d)
    x0 = x.clone()
    ops.cross_att_step2(xp2, pe, vv, att[:, 1, 0], x, scale)
    a2 = torch.softmax(c(xp2) @ c(pe).t() * scale, -1)
    assert_close(att[:, 1, 0], a2, tol, "cross-att att2")
    assert_close(x, c(x0) + torch.einsum("bt,btd->bd", a2, c(vv)), tol, "cross-att x")


def check_cross_spread(dev, B, Tn, d, dtype):
    """lina_cross_scores / lina_softmax_rows / lina_weighted_rows_add vs fp64 torch (reference crossatt.py:13-19)."""
    g = torch.Generator().manual_seed(15)
    mk = lambda *s: torch.randn(*s, generator=g).to(dtype).to(dev)
    q_lin, kk, vv = mk(B, d), mk(B, Tn, d), mk(B, Tn, d)
    ln_w, ln_b = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype).to(dev), (0.1 * torch.randn(d, generator=g)).to(dtype).to(dev)
    c = lambda t: t.cpu().to(F64)
    scale = d ** -0.5
    scores = torch.empty(B, Tn, dtype=torch.float32, device=dev)
    ops.cross_scores(q_lin, ln_w, ln_b, 1e-5, kk, scores, scale)
    q = F.layer_norm(c(q_lin), (d,), c(ln_w), c(ln_b), 1e-5)
    ref_sc = torch.einsum("bd,btd->bt", q, c(kk)) * scale
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert_close(scores, ref_sc, tol, "cross scores")
    Tp = (Tn + 31) // 32 * 32
    att = torch.zeros(B, 2, 1, Tn, dtype=dtype, device=dev)
    attc = torch.full((B, Tp), 7.0, dtype=dtype, device=dev)
    ops.softmax_rows(scores, 1.0, att[:, 1, 0], attc, Tn)
    ref_att = torch.softmax(c(scores), -1)
    assert_close(att[:, 1, 0], ref_att, tol, "softmax rows (strided)")
    assert_close(attc[:, :Tn], ref_att, tol, "softmax rows (padded copy)")
    assert (attc[:, Tn:] == 0).all() and (att[:, 0] == 0).all()
    x = mk(B, d)
    x0 = x.clone()
    ops.weighted_rows_add(attc, vv, x)
    assert_close(x, c(x0) + torch.einsum("bt,btd->bd", c(attc[:, :Tn]), c(vv)), tol, "weighted rows add")


def check_cross_fused(dev, B, Tn, d, dtype):
    """Round-2 fusions of the cross-attention step vs the launches they replace: lina_cross_scores_softmax vs
    cross_scores + softmax_rows, lina_softmax_weighted_rows_add vs softmax_rows + weighted_rows_add (other reduction
    trees inside the LayerNorm / softmax sums: fp32 1e-5, one ulp of bf16), row-major and packed x."""
    g = torch.Generator().manual_seed(29)
    mk = lambda *s_: torch.randn(*s_, generator=g).to(dtype).to(dev)
    q_lin, kk, vv = mk(B, d), mk(B, Tn, d), mk(B, Tn, d)
    ln_w, ln_b = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype).to(dev), (0.1 * torch.randn(d, generator=g)).to(dtype).to(dev)
    Tp = (Tn + 31) // 32 * 32
    scale = d ** -0.5
    scores = torch.empty(B, Tn, dtype=torch.float32, device=dev)
    att_a = torch.zeros(B, 2, 1, Tn, dtype=dtype, device=dev)
    attc_a = torch.full((B, Tp), float("nan"), dtype=dtype, device=dev)
    ops.cross_scores(q_lin, ln_w, ln_b, 1e-5, kk, scores, scale)
    ops.softmax_rows(scores, 1.0, att_a[:, 0, 0], attc_a, Tn)
    att_b = torch.zeros_like(att_a)
    attc_b = torch.full((B, Tp), float("nan"), dtype=dtype, device=dev)
    ops.cross_scores_softmax(q_lin, ln_w, ln_b, 1e-5, kk, att_b[:, 0, 0], attc_b, scale)
    # the LayerNorm statistics are summed over 16 waves instead of 4: last-bit differences in fp32, none after bf16 rounding
    tol0 = 2e-2 if dtype == torch.bfloat16 else 1e-5
    assert_close(att_b[:, 0, 0], att_a[:, 0, 0], tol0, "fused scores + softmax (att)")
    assert_close(attc_b, attc_a, tol0, "fused scores + softmax (padded copy)")
    assert float(attc_b[:, Tn:].abs().max()) == 0.0 if Tp > Tn else True
    sc2 = mk(B, Tp) * 3
    x0 = mk(B, d)
    x_a, x_b = x0.clone(), x0.clone()
    ops.softmax_rows(sc2, scale, att_a[:, 1, 0], attc_a, Tn)
    ops.weighted_rows_add(attc_a, vv, x_a)
    ops.softmax_weighted_rows_add(sc2, scale, att_b[:, 1, 0], vv, x_b)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    assert_close(att_b[:, 1, 0], att_a[:, 1, 0], tol, "fused softmax weights")
    assert_close(x_b, x_a, tol, "fused softmax + weighted rows")
    kq = 32 if dtype == torch.bfloat16 else 16
    if d % kq == 0:
        x_p = ops.pack_rows(x0)
        ops.softmax_weighted_rows_add(sc2, scale, att_b[:, 1, 0], vv, x0.clone(), x_packed=x_p)
        assert torch.equal(ops.unpack_rows(x_p, B, d), x_b), "packed residual form differs from the row-major one"
    if d % 256 == 0:
        # round 4: x_pos . pe^T folded into the same launch == the projection (model-dtype scores) + the launch above
        pe = mk(Tp, d)
        xp = mk(B, d)
        sc3 = (xp.float() @ pe.float().t()).to(dtype) if dev == "cpu" else ops.linear_skinny(xp, pe, out=torch.empty(B, Tp, dtype=dtype, device=dev))
        x_c, x_d = x0.clone(), x0.clone()
        att_c, att_d = torch.zeros_like(att_a), torch.zeros_like(att_a)
        ops.softmax_weighted_rows_add(sc3, scale, att_c[:, 1, 0], vv, x_c)
        ops.pe_softmax_weighted_rows_add(xp, pe, scale, att_d[:, 1, 0], vv, x_d)
        # (the scores are rounded to the model dtype on both sides; their fp32 sums are accumulated in different orders)
        assert_close(att_d[:, 1, 0], att_c[:, 1, 0], tol if dtype == torch.float32 else 3e-2, "pe-scores fused: softmax weights")
        assert_close(x_d, x_c, tol if dtype == torch.float32 else 3e-2, "pe-scores fused: residual stream")
        ref_sc = (xp.cpu().to(F64) @ pe.cpu().to(F64).t())[:, :Tn]
        ref_att = torch.softmax(ref_sc.to(dtype).to(F64) * scale, -1)
        assert_close(att_d[:, 1, 0], ref_att, 2e-2 if dtype == torch.bfloat16 else 1e-4, "pe-scores fused: weights vs fp64")
        xp_p, x_p = ops.pack_rows(xp), ops.pack_rows(x0)
        ops.pe_softmax_weighted_rows_add(xp_p, pe, scale, att_d[:, 1, 0], vv, x0.clone(), x_packed=x_p, xp_is_packed=True)
        assert torch.equal(ops.unpack_rows(x_p, B, d), x_d), "packed form of the pe-scores fusion differs from the row-major one"
        # att-log form (round 5): second attention's row filed at log[b, 1, step[0], :]
        cap = 4
        log = torch.full((B, 2, cap, Tn), 7.0, dtype=dtype, device=dev)
        step = torch.zeros(1, dtype=torch.int64, device=dev)
        for t in (2, cap):
            step.fill_(t)
            before = log.clone()
            x_e = x0.clone()
            ops.pe_softmax_weighted_rows_add(xp, pe, scale, log[:, 1, 0], vv, x_e, att_step=step,
                                             att_step_stride=log.stride(2), att_steps=cap)
            assert torch.equal(x_e, x_d)
            if t < cap:
                assert torch.equal(log[:, 1, t], att_d[:, 1, 0])
                before[:, 1, t] = att_d[:, 1, 0]
            assert torch.equal(log, before), "att log: something else was written"


def check_softmax_pe_rows(dev, B, Tn, d, dtype):
    """lina_softmax_pe_rows (softmax + att . pe in one launch) vs fp64 torch on the same fp32 scores -- the reference's
    first cross-attention half after the scores (crossatt.py:117-127: softmax in the model dtype, then the bmm with pe)."""
    g = torch.Generator().manual_seed(41)
    scores = (torch.randn(B, Tn, generator=g) * 2).to(dev)
    pe = torch.randn(Tn + 3, d, generator=g).to(dtype).to(dev)            # more rows than Tn: only the first Tn count
    att = torch.zeros(B, 2, 1, Tn, dtype=dtype, device=dev)
    xp = torch.full((B, d), float("nan"), dtype=dtype, device=dev)
    kq = 32 if dtype == torch.bfloat16 else 16
    xp_p = torch.zeros(ops.packed_numel(B, d), dtype=dtype, device=dev) if d % kq == 0 else None
    ops.softmax_pe_rows(scores, att[:, 0, 0], pe, xp, xp_p)
    a64 = torch.softmax(scores.cpu().to(F64), -1)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    assert_close(att[:, 0, 0], a64, tol, "softmax_pe_rows att")
    assert float(att[:, 1].abs().max()) == 0.0
    a_model = att[:, 0, 0].cpu().to(F64)                                    # the bmm sees the model-dtype weights
    assert_close(xp, a_model @ pe[:Tn].cpu().to(F64), tol, "softmax_pe_rows xp")
    if xp_p is not None:
        assert torch.equal(ops.unpack_rows(xp_p, B, d), xp), "packed copy of xp differs"
    # att-log form (round 5): the same row filed at log[b, 0, step[0], :]; a step outside the log is dropped
    cap = 5
    log = torch.full((B, 2, cap, Tn), 7.0, dtype=dtype, device=dev)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    for t in (3, 0, cap, -1):
        step.fill_(t)
        before = log.clone()
        xp2 = torch.empty_like(xp)
        ops.softmax_pe_rows(scores, log[:, 0, 0], pe, xp2, None, att_step=step, att_step_stride=log.stride(2), att_steps=cap)
        assert torch.equal(xp2, xp)
        if 0 <= t < cap:
            assert torch.equal(log[:, 0, t], att[:, 0, 0]), f"att log row at step {t}"
            before[:, 0, t] = att[:, 0, 0]
        assert torch.equal(log, before), "att log: something else was written"


def check_dwconv7_ln(dev, B, L, C, dtype, ada=False):
    """K8 vs the oracle (fp64 torch conv1d + layer_norm).  fp32 1e-5, bf16 2e-2 of max|ref|."""
    from oracle import vocoder_oracle as VO
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, L, C, generator=g).to(dtype)
    w = (torch.randn(C, 1, 7, generator=g) * 0.4).to(dtype)
    bias = torch.randn(C, generator=g).to(dtype)
    scale = (1 + 0.3 * torch.randn((B, C) if ada else (C,), generator=g)).to(dtype)
    shift = (0.3 * torch.randn((B, C) if ada else (C,), generator=g)).to(dtype)
    y = ops.dwconv7_ln(x.to(dev), w.to(dev), bias.to(dev), scale.to(dev), shift.to(dev), 1e-6)
    ry = VO.dwconv7_ln(x.to(F64), w.to(F64), bias.to(F64), scale.to(F64), shift.to(F64), 1e-6)
    assert y.dtype == dtype
    assert_close(y, ry, 1e-5 if dtype == torch.float32 else 2e-2, "K8 dwconv7+LN")
    y2 = ops.dwconv7_ln(x.to(dev), w.to(dev), None, None, None, 1e-6)
    assert_close(y2, VO.dwconv7_ln(x.to(F64), w.to(F64)), 1e-5 if dtype == torch.float32 else 2e-2, "K8 plain")


def check_istft_ola(dev, B, T, win, hop):
    """K9 vs the oracle's fold-based overlap-add (fp64): 1e-5; and the constant-overlap-add identity: frames cut
    from a signal with a Hann window reconstruct it exactly in the interior."""
    from oracle import vocoder_oracle as VO
    g = torch.Generator().manual_seed(32)
    frames = torch.randn(B, T, win, generator=g)
    window = torch.hann_window(win)
    y = ops.istft_ola(frames.to(dev), window.to(dev), hop)
    ry = VO.istft_same(frames.to(F64), window.to(F64), hop)
    assert y.shape == ry.shape == (B, T * hop if (win - hop) % 2 == 0 else y.shape[1])
    assert_close(y, ry, 1e-5, "K9 istft overlap-add")
    sig = torch.randn(B, (T - 1) * hop + win, generator=g)
    cut = sig.unfold(1, win, hop) * window                       # analysis frames (windowed once)
    rec = ops.istft_ola(cut.contiguous().to(dev), window.to(dev), hop).cpu()
    pad = (win - hop) // 2
    inner = slice(win, rec.shape[1] - win)
    if rec.shape[1] <= 2 * win:
        return
    assert (rec[:, inner] - sig[:, pad:pad + rec.shape[1]][:, inner]).abs().max() < 1e-4, "K9 does not invert the STFT framing"
