"""CPU oracle for the Lina-Speech codec-token generation path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32/fp64) restatement of the arithmetic the
reference delegates to the external ``fla`` package (flash-linear-attention @
739ef15f97cff06366c97dfdf346f2ceaadf05ce, pinned by /root/reference/README.md:26-27,
an EMPTY submodule in /root/reference/3rdparty/flash-linear-attention) plus the
small helpers of /root/reference/model/tools.py.

PARITY UNPINNED at the fla boundary: the reference ships no tests, no golden
vectors and no fla sources, so these functions follow the published GLA
recurrence (arXiv 2312.06635) as it is *used* at the reference call sites
(model/gla.py:158-220) and SURVEY.md Appendix A.  What IS pinned: the
reference's in-tree glue (model/*.py) is imported by tests/golden/make_golden.py
with this oracle bound to the ``fla.*`` names, and the resulting vectors are
committed under tests/golden/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing under lina-speech_amd/ does.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
# A.1  GLA recurrence  (fla.ops.gla.naive.naive_recurrent_gla / fused_recurrent_gla;
#      call sites /root/reference/model/gla.py:188,190,197,201)
# --------------------------------------------------------------------------- #
def naive_recurrent_gla(q, k, v, gk, initial_state=None, output_final_state=False,
                        scale: Optional[float] = None, compute_dtype=torch.float32):
    """S_t = diag(exp(g_t)) S_{t-1} + k_t^T v_t ;  o_t = scale * q_t S_t.

    q,k,gk: [B,H,T,Dk]  v: [B,H,T,Dv]  initial_state: [B,H,Dk,Dv] or None.
    Python ``for t`` loop in ``compute_dtype`` -- this is the "pure-PyTorch CPU
    recurrent path" (mode='naive', model/gla.py:196-197).  Returns (o cast to
    q.dtype, final state in compute_dtype or None).
    """
    B, H, T, Dk = q.shape
    Dv = v.shape[-1]
    if scale is None:
        scale = Dk ** -0.5
    cd = compute_dtype
    qf, kf, vf, gf = (x.to(cd) for x in (q, k, v, gk))
    S = torch.zeros(B, H, Dk, Dv, dtype=cd, device=q.device)
    if initial_state is not None:
        S = S + initial_state.to(cd)
    o = torch.empty(B, H, T, Dv, dtype=cd, device=q.device)
    for t in range(T):
        S = S * gf[:, :, t].exp().unsqueeze(-1) + kf[:, :, t].unsqueeze(-1) * vf[:, :, t].unsqueeze(-2)
        o[:, :, t] = torch.einsum("bhk,bhkv->bhv", qf[:, :, t] * scale, S)
    return o.to(q.dtype), (S if output_final_state else None)


fused_recurrent_gla = naive_recurrent_gla  # same contract at the call sites


# --------------------------------------------------------------------------- #
# A.3/A.4  chunkwise form (fla.ops.gla.chunk_gla / fused_chunk_gla;
#          call sites model/gla.py:193,195).  Must equal A.1.
# --------------------------------------------------------------------------- #
def chunk_gla(q, k, v, g, scale: Optional[float] = None, initial_state=None,
              output_final_state=False, chunk: int = 64, compute_dtype=torch.float32):
    """Chunkwise-parallel evaluation; every exponent is evaluated as a difference
    b_t - b_s <= 0, so gates of -20 (reset_val,
    model/gla.py:136,183) are harmless."""
    B, H, T, Dk = q.shape
    Dv = v.shape[-1]
    if scale is None:
        scale = Dk ** -0.5
    cd = compute_dtype
    qf, kf, vf, gf = (x.to(cd) for x in (q, k, v, g))
    S = torch.zeros(B, H, Dk, Dv, dtype=cd, device=q.device)
    if initial_state is not None:
        S = S + initial_state.to(cd)
    o = torch.empty(B, H, T, Dv, dtype=cd, device=q.device)
    for t0 in range(0, T, chunk):
        t1 = min(T, t0 + chunk)
        C = t1 - t0
        qc, kc, vc, gc = qf[:, :, t0:t1], kf[:, :, t0:t1], vf[:, :, t0:t1], gf[:, :, t0:t1]
        b = gc.cumsum(2)                                         # inclusive local cumsum
        # inter-chunk
        oi = torch.einsum("bhtk,bhkv->bhtv", qc * b.exp() * scale, S)
        # intra-chunk, exponent differences only
        diff = b.unsqueeze(3) - b.unsqueeze(2)                   # [B,H,t,s,Dk] = b_t - b_s
        mask = torch.tril(torch.ones(C, C, dtype=torch.bool, device=q.device))
        diff = diff.masked_fill(~mask[None, None, :, :, None], -float("inf"))
        A = torch.einsum("bhtk,bhsk,bhtsk->bhts", qc * scale, kc, diff.exp())
        o[:, :, t0:t1] = oi + torch.einsum("bhts,bhsv->bhtv", A, vc)
        # carry
        bC = b[:, :, -1:]
        S = S * bC.squeeze(2).exp().unsqueeze(-1) + torch.einsum("bhtk,bhtv->bhkv", kc * (bC - b).exp(), vc)
    return o.to(q.dtype), (S if output_final_state else None)


fused_chunk_gla = chunk_gla


# --------------------------------------------------------------------------- #
# A.7  scalar-gate ("simple") GLA -- fla.ops.simple_gla.chunk_simple_gla, used by
#      fla.layers.simple_gla (reference model/simple_gla.py:16,135).  Config 1
#      of BASELINE.json asks only for the pure-PyTorch CPU recurrent of this.
# --------------------------------------------------------------------------- #
def simple_gla_recurrent(q, k, v, g, scale: Optional[float] = None, initial_state=None,
                         output_final_state=False, compute_dtype=torch.float32):
    """g: [B,H,T] scalar log-decay per head.  S_t = e^{g_t} S_{t-1} + k_t^T v_t."""
    B, H, T, Dk = q.shape
    gk = g.unsqueeze(-1).expand(B, H, T, Dk)
    return naive_recurrent_gla(q, k, v, gk, initial_state, output_final_state, scale, compute_dtype)


chunk_simple_gla = simple_gla_recurrent


# --------------------------------------------------------------------------- #
# A.2  short causal depthwise conv + SiLU (fla.modules.ShortConvolution;
#      ctor model/gla.py:106-108, calls model/gla.py:161-163)
# --------------------------------------------------------------------------- #
def short_conv(x, weight, mask=None, cache=None, activation: Optional[str] = "silu", bias=None):
    """x: [B,T,D]; weight: [D,1,W] or [D,W]; mask: [B,T] or None; cache: [B,D,W] or None
    (MUTATED in place, model/gla.py:149).  Returns y [B,T,D].

    T>1 (prefill): y_t[c] = act(sum_j w[c,j] x_{t-(W-1)+j}[c]), zeros left of t=0;
    cache <- last W inputs (left zero padded when T<W).
    T==1 with cache (step): cache <- roll(cache,-1); cache[...,-1] <- x; y = act(sum_j cache_j w_j).
    """
    B, T, D = x.shape
    w = weight.reshape(D, -1)
    W = w.shape[1]
    xd = x.dtype
    xf = x.float()
    if mask is not None:
        xf = xf * mask.unsqueeze(-1).to(xf.dtype)
    if cache is not None and T == 1:
        new = torch.roll(cache.float(), shifts=-1, dims=-1)
        new[:, :, -1] = xf[:, 0]
        cache.copy_(new.to(cache.dtype))
        y = (new * w.float().unsqueeze(0)).sum(-1)
        if bias is not None:
            y = y + bias.float()
        y = y.unsqueeze(1)
    else:
        xt = xf.transpose(1, 2)                                     # [B,D,T]
        if cache is not None:
            keep = xt[:, :, -W:]
            cache.copy_(F.pad(keep, (W - keep.shape[-1], 0)).to(cache.dtype))
        y = F.conv1d(F.pad(xt, (W - 1, 0)), w.float().unsqueeze(1), bias=None if bias is None else bias.float(),
                     groups=D).transpose(1, 2)
    if activation in ("silu", "swish"):
        y = F.silu(y)
    elif activation is not None:
        raise ValueError(activation)
    return y.to(xd)


# --------------------------------------------------------------------------- #
# A.6  RMSNorm (x) swish gate (fla.modules.FusedRMSNormSwishGate, model/gla.py:111,219)
# --------------------------------------------------------------------------- #
def rmsnorm(x, weight=None, eps: float = 1e-5):
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    if weight is not None:
        y = y * weight.float()
    return y.to(x.dtype)


def rmsnorm_swish_gate(x, g, weight=None, eps: float = 1e-5):
    xf, gf = x.float(), g.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    if weight is not None:
        y = y * weight.float()
    return (y * gf * torch.sigmoid(gf)).to(x.dtype)


# --------------------------------------------------------------------------- #
# A.7  gate prologue (model/gla.py:174-183)
# --------------------------------------------------------------------------- #
def gate_logsigmoid(gk_pre, normalizer: float = 16.0, clamp_min: Optional[float] = None):
    gk = F.logsigmoid(gk_pre.float()) / normalizer
    if clamp_min is not None:
        gk = torch.clamp_min(gk, clamp_min)
    return gk.to(gk_pre.dtype)


# --------------------------------------------------------------------------- #
# K6 helpers (model/multiembed.py:21-23, model/tools.py:38-67)
# --------------------------------------------------------------------------- #
def embed_sum(weight, idx):
    """weight [q,n_emb,d]; idx [q,B,n] -> sum_q weight[q, idx[q]]  -> [B,n,d]
    (MultiEmbedding.forward + reduce 'q b n d -> b n d', modeling_lina.py:131,178-179)."""
    out = 0
    for qi in range(weight.shape[0]):
        out = out + weight[qi][idx[qi]]
    return out


def argmax_lowest(logits):
    """argmax over the last dim, lowest index on exact ties.  topk_sampling(k=1)
    (tools.py:38-44) equals this except on exact ties, where the reference
    samples uniformly among the tied entries."""
    m = logits.max(-1, keepdim=True).values
    n = logits.shape[-1]
    idx = torch.arange(n, device=logits.device).expand_as(logits)
    return torch.where(logits == m, idx, torch.full_like(idx, n)).min(-1).values


def topk_sampling(seq, k=1, temp=1.0, generator=None):
    """Restatement of model/tools.py:38-44 (does not mutate its input)."""
    topk = torch.topk(seq, k, dim=-1)
    logits = seq / temp
    logits = logits.masked_fill(logits < topk.values[:, [-1]], -float("inf"))
    probs = torch.softmax(logits, dim=-1)
    return torch.multinomial(probs, num_samples=1, generator=generator)


def undelay_rvq(extended_code):
    """model/tools.py:61-67."""
    q, _, n = extended_code.shape
    out = torch.stack([torch.roll(extended_code[i], -(i + 1), dims=1) for i in range(q)], dim=0)
    return out[:, :, :-(q + 1)]


def delay_rvq(code, head_token: int = -2, tail_token: int = -3):
    """model/tools.py:46-59."""
    q, _ = code.shape
    ext = torch.ones((q, q + 1)).tril() * head_token
    ext += torch.ones((q + 1, q)).tril(diagonal=-1).T * tail_token
    ext = torch.flip(ext, (1,))
    ext = torch.cat((code, ext), axis=1)
    for i in range(q):
        ext[i, :] = torch.roll(ext[i, :], i + 1)
    return ext.long()


# --------------------------------------------------------------------------- #
# f-2  top-k sampling as an inverse CDF of a given uniform number (checker for K6c).
#      Same distribution as topk_sampling above (model/tools.py:38-44), incl. its quirk that the tempered
#      logits are compared with the UNtempered k-th largest value.
# --------------------------------------------------------------------------- #
def hash_uniform(seed: int, step: int, row: int, rows: int) -> float:
    """splitmix64 finaliser over (seed, step, row) -> 24-bit uniform in [0,1) (mirrors sample.hip)."""
    M = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15 * (step * rows + row + 1)) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    z = z ^ (z >> 31)
    return (z >> 40) / 16777216.0


def topk_sample_inverse_cdf(seq, k, temp, u):
    """seq [rows, n], u [rows] in [0,1) -> (token [rows], margin [rows], probs [rows, n]); fp64.
    margin = distance of u from the nearest CDF edge of the picked token (tokens whose margin is below
    the fp32 rounding of the kernel's running sums are not comparable)."""
    x = seq.to(torch.float64)
    kth = torch.topk(x, min(k, x.shape[-1]), dim=-1).values[:, -1:]
    logits = x / temp
    logits = logits.masked_fill(logits < kth, -float("inf"))
    p = torch.softmax(logits, dim=-1)
    cdf = torch.cumsum(p, dim=-1)
    uu = u.to(torch.float64).unsqueeze(-1)
    tok = (cdf <= uu).sum(-1).clamp_max(x.shape[-1] - 1)
    # step over zero-probability entries the search may have landed on at the very end
    kept = p > 0
    last_kept = (kept * torch.arange(x.shape[-1])).max(-1).values
    tok = torch.minimum(tok, last_kept)
    hi = cdf.gather(-1, tok.unsqueeze(-1)).squeeze(-1)
    lo = hi - p.gather(-1, tok.unsqueeze(-1)).squeeze(-1)
    margin = torch.minimum(u.to(torch.float64) - lo, hi - u.to(torch.float64))
    return tok, margin, p
