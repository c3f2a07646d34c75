"""Thin launchers of the C ABI (include/lina_gla.h): torch tensors in, one enqueue on the CURRENT torch HIP stream, no
synchronisation (graph-capturable), no autograd.  The differentiable operators of autograd.py and the decode engine are built
on these."""
from __future__ import annotations

from typing import Optional

import torch

from . import backend as _backend
from .backend import _bht, _check, _dt, _inner_contig, _no_grad, _ptr, _workspace, fused_ops_available
from .policy import POLICY, _value_blocks, chunk_segments
from ._lib import LOOP_CTL_ROWS


# --------------------------------------------------------------------------- GLA (K1 / K2 / K2b)
def _gla_prepare(q, k, v, gk, scale, initial_state):
    """Shape / dtype checks and the layout normalisation shared by forward and backward.  Uses only
    differentiable torch ops, so it may run outside the autograd Function."""
    if q.dim() != 4:
        raise ValueError("q must be [B,H,T,Dk]")
    B, H, T, Dk = q.shape
    Dv = v.shape[-1]
    if k.shape != q.shape or gk.shape != q.shape or v.shape[:3] != q.shape[:3]:
        raise ValueError(f"shape mismatch q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)} gk{tuple(gk.shape)}")
    if k.dtype != q.dtype or v.dtype != q.dtype:
        raise TypeError("q, k, v must share a dtype")
    _backend._BACKEND.require(q, k, v, gk, initial_state)
    if gk.dtype != q.dtype and gk.dtype != torch.float32:
        gk = gk.float()
    if q.dtype == torch.float32 and gk.dtype != torch.float32:
        gk = gk.float()
    q, k, v, gk = (_inner_contig(x) for x in (q, k, v, gk))
    if v.stride(0) % 4 or v.stride(1) % 4 or v.stride(2) % 4:
        v = v.contiguous()
    if scale is None:
        scale = Dk ** -0.5
    if initial_state is not None and tuple(initial_state.shape) != (B, H, Dk, Dv):
        raise ValueError(f"initial_state must be [B,H,Dk,Dv]={B, H, Dk, Dv}, got {tuple(initial_state.shape)}")
    return q, k, v, gk, float(scale)


def _head_first_empty(B, H, T, D, dtype, device):
    # laid out [B,T,H,D] in memory and returned as the head-first view, so the caller's
    # 'b h l d -> b l h d' rearrange (reference model/gla.py:215) is free.
    return torch.empty(B, T, H, D, dtype=dtype, device=device).transpose(1, 2)


def _gla_launch(entry: str, q, k, v, gk, scale, initial_state, output_final_state, inplace_state=False, nseg=None,
                keep_seg_states: Optional[list] = None, out: Optional[torch.Tensor] = None):
    """``keep_seg_states``: a list that receives (workspace, nseg) when the segment-parallel kernel ran -- the workspace is
    then a fresh tensor whose head holds the segment start states (the backward's seg_states), not the shared scratch."""
    B, H, T, Dk = q.shape
    Dv = v.shape[-1]
    be = _backend._BACKEND
    m = _value_blocks(q, v, gk) if entry == "lina_gla_chunk_fwd" else 1
    if m == 2 and POLICY.dv512_one_launch and (chunk_segments(2 * B * H, T) if nseg is None else nseg) <= 1:
        m = 1                                   # the C entry runs both value column blocks of a 256 x 512 head in one launch
    if m > 1:
        o = _head_first_empty(B, H, T, Dv, q.dtype, q.device)
        ht = torch.empty(B, H, Dk, Dv, dtype=torch.float32, device=q.device) if output_final_state else None
        for j in range(m):
            cols = slice(j * Dk, (j + 1) * Dk)
            h0j = None if initial_state is None else initial_state[..., cols].float().contiguous()
            _, htj = _gla_launch(entry, q, k, v[..., cols], gk, scale, h0j, output_final_state, False, nseg,
                                 keep_seg_states, out=o[..., cols])
            if ht is not None:
                ht[..., cols] = htj
        if inplace_state and ht is not None and initial_state is not None and initial_state.dtype == torch.float32:
            initial_state.copy_(ht)
            ht = initial_state
        return o, ht
    o = _head_first_empty(B, H, T, Dv, q.dtype, q.device) if out is None else out
    h0 = None
    if initial_state is not None:
        h0 = initial_state
        if h0.dtype != torch.float32 or not h0.is_contiguous():
            h0 = h0.float().contiguous()
            inplace_state = False
    ht = None
    if output_final_state:
        ht = h0 if (inplace_state and h0 is not None) else torch.empty(B, H, Dk, Dv, dtype=torch.float32,
                                                                      device=q.device)
    if entry == "lina_gla_chunk_fwd":
        full = q.dtype == torch.bfloat16 and gk.dtype == torch.bfloat16 and Dk == Dv and Dk in (64, 128, 256)
        groups = 256 // Dk if full else 1                       # heads per workgroup of the full-head kernel
        nseg = chunk_segments(B * H // groups, T) if nseg is None else nseg
        if nseg > 1 and full and H % groups == 0:
            nbytes = int(be.lib.lina_gla_chunk_fwd_seg_workspace(B, H, Dk, Dv, nseg))
            ws = (_workspace("k2seg", nbytes, q.device) if keep_seg_states is None
                  else torch.empty(nbytes // 4, dtype=torch.float32, device=q.device))
            rc = be.lib.lina_gla_chunk_fwd_seg(_ptr(q), _ptr(k), _ptr(v), _ptr(gk), _ptr(o), _ptr(h0), _ptr(ht), _ptr(ws),
                                               nseg, B, H, T, Dk, Dv, _bht(q), _bht(k), _bht(v), _bht(gk), _bht(o),
                                               _dt(q), _dt(gk), scale, be.stream(q))
            if rc == 0:
                if keep_seg_states is not None:
                    keep_seg_states.append((ws, nseg))
                return o, ht
            if rc != -2:                         # -2 = layout not eligible for the segmented kernel: use the plain one
                _check(rc)
    fn = getattr(be.lib, entry)
    _check(fn(_ptr(q), _ptr(k), _ptr(v), _ptr(gk), _ptr(o), _ptr(h0), _ptr(ht), B, H, T, Dk, Dv,
              _bht(q), _bht(k), _bht(v), _bht(gk), _bht(o), _dt(q), _dt(gk), scale, be.stream(q)))
    return o, ht


def gla_chunk_bwd(q, k, v, gk, d_o, scale, initial_state=None, final_state=None, d_final_state=None,
                  need_dh0=False, nseg=None, path=None, seg_states=None):
    """K2b through the C ABI: returns (dq, dk, dv, dg, dh0).  bf16 tensors with Dk = Dv in {64,128,256} take the
    full-head sweeps (lina_gla_chunk_bwd_full, ``nseg`` sequence segments); everything else -- and ``path="sweeps"`` /
    ``policy.POLICY.k2b_path = "sweeps"`` -- the generic kernel (lina_gla_chunk_bwd).  ``seg_states``: the workspace the segment-parallel forward
    left for the same inputs and ``nseg`` (its head holds the segment start states; skips one pass)."""
    B, H, T, Dk = q.shape
    Dv = v.shape[-1]
    be = _backend._BACKEND
    be.require(q, k, v, gk, d_o, initial_state, final_state, d_final_state)
    m = _value_blocks(q, v, gk) if (path or POLICY.k2b_path) == "full" else 1
    if m > 1:                                               # one 256 x 256 backward per value column block
        dq = dk = dg = None
        dvs, dh0s = [], []
        states = list(seg_states) if isinstance(seg_states, (list, tuple)) else [None] * m
        for j in range(m):
            cols = slice(j * Dk, (j + 1) * Dk)
            part = lambda t: None if t is None else t[..., cols].float().contiguous()
            gq, gk_, gv, gg, gh = gla_chunk_bwd(q, k, v[..., cols], gk, d_o[..., cols], scale, part(initial_state),
                                                part(final_state), part(d_final_state), need_dh0, nseg, path, states[j])
            dq = gq.float() if dq is None else dq + gq.float()
            dk = gk_.float() if dk is None else dk + gk_.float()
            dg = gg.float() if dg is None else dg + gg.float()
            dvs.append(gv)
            dh0s.append(gh)
        return (dq.to(q.dtype), dk.to(q.dtype), torch.cat(dvs, dim=-1), dg.to(gk.dtype),
                torch.cat(dh0s, dim=-1) if need_dh0 else None)
    d_o = _inner_contig(d_o.to(q.dtype))
    if d_o.stride(0) % 4 or d_o.stride(1) % 4 or d_o.stride(2) % 4:
        d_o = d_o.contiguous()
    q, k = (x if not (x.stride(0) % 4 or x.stride(1) % 4 or x.stride(2) % 4) else x.contiguous() for x in (q, k))
    h0 = None if initial_state is None else initial_state.float().contiguous()
    dht = None if d_final_state is None else d_final_state.float().contiguous()
    dg_tail = None
    if dht is not None:
        if final_state is None:
            raise ValueError("a gradient for the final state needs the final state itself")
        dg_tail = (final_state.float() * dht).sum(-1).contiguous()
    # dq | dk | dv as column slabs of ONE [B, T, H (2 Dk + Dv)] buffer: the backward of a fused q | k | v short convolution
    # (autograd.short_conv3) reads them as a single operand; for anybody else they are three head-first tensors as before
    qkv = torch.empty(B, T, H * (2 * Dk + Dv), dtype=q.dtype, device=q.device)
    dq = qkv[..., :H * Dk].view(B, T, H, Dk).transpose(1, 2)
    dk = qkv[..., H * Dk:2 * H * Dk].view(B, T, H, Dk).transpose(1, 2)
    dv = qkv[..., 2 * H * Dk:].view(B, T, H, Dv).transpose(1, 2)
    dg = _head_first_empty(B, H, T, Dk, gk.dtype, q.device)
    dh0 = torch.empty(B, H, Dk, Dv, dtype=torch.float32, device=q.device) if need_dh0 else None
    path = path or POLICY.k2b_path
    full = (path == "full" and q.dtype == torch.bfloat16 and gk.dtype == torch.bfloat16 and Dk == Dv
            and Dk in (64, 128, 256) and H % (256 // Dk) == 0)
    if full:
        v = _inner_contig(v)
        ns = chunk_segments(B * H // (256 // Dk), T) if nseg is None else int(nseg)
        ws = _workspace("k2b", int(be.lib.lina_gla_chunk_bwd_full_workspace(B, H, T, Dk, Dv, ns)), q.device)
        rc = be.lib.lina_gla_chunk_bwd_full(_ptr(q), _ptr(k), _ptr(v), _ptr(gk), _ptr(d_o), _ptr(h0), _ptr(dht),
                                            _ptr(dg_tail), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(dg), _ptr(dh0), _ptr(ws),
                                            _ptr(seg_states if ns > 1 else None), ns, B, H, T, Dk, Dv, _bht(q), _bht(k), _bht(v), _bht(gk), _bht(d_o),
                                            _bht(dq), _bht(dk), _bht(dv), _bht(dg), _dt(q), _dt(gk), float(scale),
                                            be.stream(q))
        if rc == 0:
            return dq, dk, dv, dg, dh0
        if rc != -2:                             # -2 = layout not eligible: the generic kernel below
            _check(rc)
    nbytes = int(be.lib.lina_gla_chunk_bwd_workspace(B, H, T, Dk, Dv))
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=q.device)
    _check(be.lib.lina_gla_chunk_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(gk), _ptr(d_o), _ptr(h0), _ptr(dht),
                                     _ptr(dg_tail), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(dg), _ptr(dh0), _ptr(ws),
                                     B, H, T, Dk, Dv, _bht(q), _bht(k), _bht(v), _bht(gk), _bht(d_o), _bht(dq),
                                     _bht(dk), _bht(dv), _bht(dg), _dt(q), _dt(gk), float(scale), be.stream(q)))
    return dq, dk, dv, dg, dh0


# --------------------------------------------------------------------------- short conv (K3 / K4)
def _short_conv_launch(x, w, bias, mask, cache, act):
    B, T, D = x.shape
    W = w.shape[1]
    be = _backend._BACKEND
    y = torch.empty(B, T, D, dtype=x.dtype, device=x.device)
    if cache is not None and T == 1:
        if mask is not None:
            x = x * mask.unsqueeze(-1).to(x.dtype)
        _check(be.lib.lina_short_conv_step(_ptr(x), _ptr(w), _ptr(bias), _ptr(cache), _ptr(y), B, D, W,
                                           x.stride(0), y.stride(0), act, _dt(x), be.stream(x)))
    else:
        _check(be.lib.lina_short_conv_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(mask), _ptr(cache), _ptr(y), B, T, D, W,
                                          x.stride(0), x.stride(1), y.stride(0), y.stride(1), act, _dt(x),
                                          be.stream(x)))
    return y


def _sum_partials(part, out_dtype=torch.float32):
    """K13: sum over the partial-row axis of the fp32 ``*_partial`` buffer of a backward kernel -- ``part`` [P, ...] ->
    [...], or with ``outer`` leading slabs [O, P, ...] -> [O, ...] when ``part.dim() - 1`` trailing dims are given as one."""
    P = part.shape[0]
    N = part.numel() // max(P, 1)
    if N % 4 or N == 0 or out_dtype not in (torch.float32, torch.bfloat16):
        return part.sum(0).to(out_dtype)
    be = _backend._BACKEND
    out = torch.empty(part.shape[1:], dtype=out_dtype, device=part.device)
    _check(be.lib.lina_sum_partials(_ptr(part), _ptr(out), 1, P, N, _dt(out), be.stream(part)))
    return out


def _sum_partials2(part):
    """``part`` fp32 [O, P, N] -> [O, N] (K13 with an outer axis: the two parameter gradients of the LayerNorm)."""
    O, P, N = part.shape
    if N % 4 or N == 0:
        return part.sum(1)
    be = _backend._BACKEND
    out = torch.empty(O, N, dtype=torch.float32, device=part.device)
    _check(be.lib.lina_sum_partials(_ptr(part), _ptr(out), O, P, N, _dt(out), be.stream(part)))
    return out


def column_sum(x2, out_dtype=torch.float32):
    """``x2.sum(0)`` of a matrix [M, N] with fp32 accumulation (K13a + K13: deterministic, no global semaphores -- torch's
    two-stage reduction for this shape does not survive a hipGraph replay on ROCm 7.2, tools/probe_graph_memset.py)."""
    M, N = x2.shape
    if (not fused_ops_available(x2) or x2.dtype not in (torch.float32, torch.bfloat16) or N % 4 or x2.stride(1) != 1
            or x2.stride(0) % 4 or M == 0 or M > 65535 * 128):
        return x2.sum(0, dtype=torch.float32).to(out_dtype)
    be = _backend._BACKEND
    part = torch.empty(int(be.lib.lina_swiglu_bwd_partials(M)), N, dtype=torch.float32, device=x2.device)
    _check(be.lib.lina_colsum(_ptr(x2), _ptr(part), M, N, x2.stride(0), _dt(x2), be.stream(x2)))
    return _sum_partials(part, out_dtype)


def _sum_vector(v):
    """Sum of a long fp32 vector as a 0-dim tensor through K13 (rows of 4 as the "partials") and a 4-element tail -- no
    multi-block torch reduction (see ``column_sum``)."""
    n = v.numel()
    if n < 8 or n % 4 or not fused_ops_available(v) or v.dtype != torch.float32:
        return v.sum()
    w = 256 if n % 256 == 0 else 4
    return _sum_partials(v.contiguous().view(n // w, w)).sum()


# --------------------------------------------------------------------------- codec head (K6)
def _embed_sum_launch(table, flat, out=None):
    be = _backend._BACKEND
    Q, n_emb, d = table.shape
    N = flat.shape[1]
    if out is None:
        out = torch.empty(N, d, dtype=table.dtype, device=table.device)
    elif tuple(out.shape) != (N, d) or out.dtype != table.dtype or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous {table.dtype} tensor [{N}, {d}]")
    _check(be.lib.lina_embed_sum(_ptr(flat), _ptr(table.contiguous()), _ptr(out), Q, N, n_emb, d, _dt(table),
                                 be.stream(table)))
    return out


def argmax_rows(logits, out=None):
    """Greedy pick over the last dim, lowest index on ties (topk_sampling(k=1), reference tools.py:38-44)."""
    be = _backend._BACKEND
    be.require(logits)
    lg = _inner_contig(logits)
    n = lg.shape[-1]
    lg2 = lg.reshape(-1, n)
    if out is None:
        out = torch.empty(lg2.shape[0], dtype=torch.int64, device=lg.device)
    _check(be.lib.lina_argmax_rows(_ptr(lg2), _ptr(out), lg2.shape[0], n, lg2.stride(0), _dt(lg2), be.stream(lg2)))
    return out.view(logits.shape[:-1])


def _loop_ctl_ptr(be, loop_ctl, B):
    if loop_ctl is None:
        return None
    be.require(loop_ctl)
    if loop_ctl.dtype != torch.int32 or not loop_ctl.is_contiguous() or loop_ctl.numel() < LOOP_CTL_ROWS + B:
        raise ValueError("loop_ctl must be a contiguous int32 tensor of LOOP_CTL_ROWS + B words (new_loop_ctl)")
    return _ptr(loop_ctl)


def new_loop_ctl(B: int, device, seed_word: int = 0) -> torch.Tensor:
    """A fresh loop-control block (include/lina_gla.h): no row stopped, stop step -1, the per-call seed word."""
    host = torch.zeros(LOOP_CTL_ROWS + B, dtype=torch.int32)
    host[1] = -1
    lo, hi = seed_word & 0xFFFFFFFF, (seed_word >> 32) & 0xFFFFFFFF
    host[2] = lo - (1 << 32) if lo >= (1 << 31) else lo
    host[3] = hi - (1 << 32) if hi >= (1 << 31) else hi
    return host.to(device)


def greedy_pick_embed(logits, table, x_out, tok_log, step, counter, x_packed=None, loop_ctl=None):
    """K6d (lina_greedy_pick_embed): arg-max per quantizer of ``logits [B, Q, L]``, the picks logged at
    ``tok_log[step[0]]`` ([max_steps, Q, B] int64), the next input ``x_out [B, d] = sum_q table[q, pick_q]`` and
    ``step[0] += 1`` -- one launch.  ``counter``: int32 [1], zero."""
    be = _backend._BACKEND
    be.require(logits, table, x_out, tok_log, step, counter)
    B, Q, L = logits.shape
    Qt, n_emb, d = table.shape
    if Qt != Q or logits.stride(2) != 1 or logits.stride(1) != L:
        raise ValueError("logits must be [B, Q, L] with contiguous (Q, L)")
    if tuple(x_out.shape) != (B, d) or not x_out.is_contiguous() or x_out.dtype != table.dtype or logits.dtype != table.dtype:
        raise ValueError("x_out must be a contiguous [B, d] tensor of the table's dtype")
    if tok_log.dtype != torch.int64 or tok_log.dim() != 3 or tuple(tok_log.shape[1:]) != (Q, B) or not tok_log.is_contiguous():
        raise ValueError("tok_log must be a contiguous int64 [max_steps, Q, B] tensor")
    if step.dtype != torch.int64 or counter.dtype != torch.int32:
        raise ValueError("step must be int64, counter int32")
    be.require(x_packed)
    if x_packed is not None and x_packed.numel() < packed_numel(B, d):
        raise ValueError("packed x buffer is too small")
    _check(be.lib.lina_greedy_pick_embed(_ptr(logits), logits.stride(0), _ptr(table.contiguous()), _ptr(x_out), _ptr(x_packed),
                                         _ptr(tok_log),
                                         _ptr(step), _ptr(counter), _loop_ctl_ptr(be, loop_ctl, B), B, Q, L, n_emb, d,
                                         tok_log.shape[0], _dt(table),
                                         be.stream(table)))


def sample_pick_embed(logits, table, x_out, tok_log, step, counter, n_sampled: int, k: int, temp: float = 1.0,
                      seed: int = 0, x_packed=None, loop_ctl=None):
    """K6e (lina_sample_pick_embed): greedy_pick_embed for the reference's default generation mode -- quantizers
    ``q < n_sampled`` are sampled (top-``k``, temperature, the draw of row ``b*Q + q`` of topk_sample_rows at the same
    (seed, step)), the others take the arg-max; token log, next-input embedding and ``step[0] += 1`` in the same launch."""
    be = _backend._BACKEND
    be.require(logits, table, x_out, tok_log, step, counter, x_packed)
    B, Q, L = logits.shape
    Qt, n_emb, d = table.shape
    if Qt != Q or logits.stride(2) != 1 or logits.stride(1) != L:
        raise ValueError("logits must be [B, Q, L] with contiguous (Q, L)")
    if tuple(x_out.shape) != (B, d) or not x_out.is_contiguous() or x_out.dtype != table.dtype or logits.dtype != table.dtype:
        raise ValueError("x_out must be a contiguous [B, d] tensor of the table's dtype")
    if tok_log.dtype != torch.int64 or tok_log.dim() != 3 or tuple(tok_log.shape[1:]) != (Q, B) or not tok_log.is_contiguous():
        raise ValueError("tok_log must be a contiguous int64 [max_steps, Q, B] tensor")
    if step.dtype != torch.int64 or counter.dtype != torch.int32:
        raise ValueError("step must be int64, counter int32")
    if x_packed is not None and x_packed.numel() < packed_numel(B, d):
        raise ValueError("packed x buffer is too small")
    _check(be.lib.lina_sample_pick_embed(_ptr(logits), logits.stride(0), _ptr(table.contiguous()), _ptr(x_out),
                                         _ptr(x_packed), _ptr(tok_log), _ptr(step), _ptr(counter),
                                         _loop_ctl_ptr(be, loop_ctl, B), B, Q, L, n_emb, d,
                                         tok_log.shape[0], int(n_sampled), int(k), float(temp),
                                         int(seed) & 0xFFFFFFFFFFFFFFFF, _dt(table), be.stream(table)))


# --------------------------------------------------------------------------- decode-step fusions
def topk_sample_rows(logits, k: int, temp: float = 1.0, u: Optional[torch.Tensor] = None, seed: int = 0,
                     step: Optional[torch.Tensor] = None, out=None):
    """K6c: one top-k / temperature sample per row of ``logits [..., n]`` -> int64 ``[...]`` (reference
    tools.py:38-44 for k > 1).  ``u``: fp32 uniforms [rows] (else hashed from (seed, step[0], row); ``step`` is a
    device int64 tensor)."""
    be = _backend._BACKEND
    be.require(logits, u, step)
    n = logits.shape[-1]
    flat = logits.reshape(-1, n)
    if flat.stride(-1) != 1:
        flat = flat.contiguous()
    rows = flat.shape[0]
    if u is not None:
        u = u.reshape(-1).to(torch.float32).contiguous()
        if u.numel() != rows:
            raise ValueError("u must hold one uniform number per row")
    if step is not None and (step.dtype != torch.int64 or step.numel() < 1):
        raise ValueError("step must be an int64 tensor")
    if out is None:
        out = torch.empty(rows, dtype=torch.int64, device=logits.device)
    _check(be.lib.lina_topk_sample_rows(_ptr(flat), _ptr(out), rows, n, flat.stride(0), int(k), float(temp), _ptr(u),
                                        int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(step), _dt(flat), be.stream(flat)))
    return out.view(logits.shape[:-1])


def gla_decode_prologue(z, off_q, off_k, off_v, off_lr, wq, wk, wv, cq, ck, cv, w2, b2, qkv, gk,
                        normalizer: float = 16.0, clamp_min: Optional[float] = None):
    """K4x3 + K7 in one launch (reference model/gla.py:158-163,174-180 at T = 1). See lina_gla.h."""
    be = _backend._BACKEND
    be.require(z, wq, wk, wv, cq, ck, cv, w2, b2, qkv, gk)
    B = z.shape[0]
    Kd, W = wq.shape[0], wq.shape[-1]
    Vd = wv.shape[0]
    R = w2.shape[1]
    _check(be.lib.lina_gla_decode_prologue(_ptr(z), z.stride(0), off_q, off_k, off_v, off_lr, _ptr(wq), _ptr(wk),
                                           _ptr(wv), _ptr(cq), _ptr(ck), _ptr(cv), _ptr(w2), _ptr(b2), _ptr(qkv),
                                           _ptr(gk), B, Kd, Vd, W, R, float(normalizer),
                                           float("nan") if clamp_min is None else float(clamp_min), _dt(z),
                                           be.stream(z)))


def swiglu(u, hidden: int, out=None, pad_to: Optional[int] = None):
    """y = silu(u[..., :hidden]) * u[..., hidden:2*hidden]  (reference base_blocks.py:48-50).
    ``pad_to`` > hidden: the row is padded; column ``hidden`` holds 1 (bias column), the rest 0."""
    _no_grad(u)
    be = _backend._BACKEND
    be.require(u)
    u2 = _inner_contig(u).reshape(-1, u.shape[-1])
    ld_y = pad_to or hidden
    if out is None:
        out = torch.empty(u2.shape[0], ld_y, dtype=u.dtype, device=u.device)
    _check(be.lib.lina_swiglu(_ptr(u2), _ptr(out), u2.shape[0], hidden, u2.stride(0), out.stride(0), _dt(u2),
                              be.stream(u2)))
    return out.view(*u.shape[:-1], ld_y)


def gla_decode_update(q, k, v, gk, o_part, state, scale=None):
    """K1d: in-place decode-step state update, row-split (see lina_gla.h).  q,k,gk [B,H,Dk], v [B,H,Dv]
    (strided views, last dim contiguous); state fp32 [B,H,Dk,Dv]; o_part fp32 [Dk/64, B, H, Dv]."""
    be = _backend._BACKEND
    be.require(q, k, v, gk, o_part, state)
    B, H, Dk = q.shape
    Dv = v.shape[-1]
    if state.dtype != torch.float32 or not state.is_contiguous() or tuple(state.shape) != (B, H, Dk, Dv):
        raise ValueError("state must be contiguous fp32 [B,H,Dk,Dv]")
    if o_part.dtype != torch.float32 or not o_part.is_contiguous() or tuple(o_part.shape) != (Dk // 64, B, H, Dv):
        raise ValueError("o_part must be contiguous fp32 [Dk/64,B,H,Dv]")
    for t in (q, k, v, gk):
        if t.stride(-1) != 1:
            raise ValueError("innermost dimension must be contiguous")
    _check(be.lib.lina_gla_decode_update(_ptr(q), _ptr(k), _ptr(v), _ptr(gk), _ptr(o_part), _ptr(state), B, H, Dk, Dv,
                                         q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                                         gk.stride(0), gk.stride(1), _dt(q), _dt(gk),
                                         float(Dk ** -0.5 if scale is None else scale), be.stream(q)))
    return o_part


def _kstep(dtype) -> tuple:
    """(KSTEP, KL): contraction elements per MFMA step / per lane (16 bytes) for bf16 and fp32 fragments."""
    return (32, 8) if dtype == torch.bfloat16 else (16, 4)


def packed_numel(rows: int, cols: int) -> int:
    return (rows + 63) // 64 * 64 * cols


def pack_rows(t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[M, K] -> the fragment-major layout of include/lina_gla.h (flat tensor, rows zero-padded to a multiple of 64):
    element (m, k) at ((m/16 * K/KSTEP + k/KSTEP) * 64 + m%16 + 16*((k%KSTEP)/KL)) * KL + k%KL.  Plain torch ops: used
    once per weight at engine construction and to seed packed activation buffers."""
    M, K = t.shape
    ks, kl = _kstep(t.dtype)
    if K % ks:
        raise ValueError(f"K={K} must be a multiple of {ks}")
    Mp = (M + 63) // 64 * 64
    src = t
    if Mp != M:
        src = torch.zeros(Mp, K, dtype=t.dtype, device=t.device)
        src[:M] = t
    p = src.view(Mp // 16, 16, K // ks, 4, kl).permute(0, 2, 3, 1, 4).reshape(-1)
    if out is None:
        return p.contiguous()
    out.view(-1)[:p.numel()].copy_(p)
    return out


def unpack_rows(p: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """Inverse of pack_rows (tests / debugging)."""
    ks, kl = _kstep(p.dtype)
    Mp = (rows + 63) // 64 * 64
    return p.view(-1)[:Mp * cols].view(Mp // 16, cols // ks, 4, 16, kl).permute(0, 3, 1, 2, 4).reshape(Mp, cols)[:rows]


def linear_skinny_packed(a_packed, w_packed, M: int, N: int, K: int, c1=None, c2=None, resid=None, out=None,
                         out_packed=None, out_packed_width: int = 0, swiglu_hidden: int = 0, ln_dim: int = 0,
                         ln_eps: float = 1e-5, w_half_rows: Optional[int] = None, dtype=None, w_stream: bool = False):
    """lina_linear_skinny_ex with fragment-major A [M,K] and W (pack_rows; for SwiGLU both weight halves packed
    separately and concatenated, ``w_half_rows`` = padded rows of one half).  ``out`` [M,N] row-major and / or
    ``out_packed`` (the packed A operand of the next projection, width ``out_packed_width`` >= N)."""
    be = _backend._BACKEND
    be.require(a_packed, w_packed, c1, c2, resid, out, out_packed)
    dt = a_packed.dtype
    if out is None and out_packed is None:
        out = torch.empty(M, N, dtype=dt, device=a_packed.device)
    for t in (c1, c2):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise ValueError("c1/c2 must be contiguous fp32 vectors")
    if a_packed.numel() < packed_numel(M, K):
        raise ValueError("packed A is too small for [M, K]")
    n_w = 2 * swiglu_hidden if swiglu_hidden else N
    half = w_half_rows if w_half_rows is not None else (n_w + 63) // 64 * 64
    if half % 64 or half < (N + 31) // 32 * 32:        # the launch reads whole 16/32-row weight tiles up to column N
        raise ValueError("packed W: rows per half must be a multiple of 64 covering the N output columns")
    if w_packed.numel() < half * K * (2 if swiglu_hidden else 1):
        raise ValueError("packed W is too small")
    if out_packed is not None and out_packed.numel() < packed_numel(M, out_packed_width):
        raise ValueError("packed output buffer is too small")
    _check(be.lib.lina_linear_skinny_ex(_ptr(a_packed), 0, _ptr(w_packed), 0, 3 if w_stream else 1, int(half), _ptr(c1), _ptr(c2),
                                        _ptr(resid), 0 if (resid is None or resid.dim() < 2) else resid.stride(0), _ptr(out),
                                        0 if out is None else out.stride(0), _ptr(out_packed), int(out_packed_width),
                                        M, N, K, swiglu_hidden, ln_dim, float(ln_eps), _dt(a_packed),
                                        be.stream(a_packed)))
    return out if out is not None else out_packed


def linear_skinny(a, w, c1=None, c2=None, resid=None, out=None, swiglu_hidden: int = 0, ln_dim: int = 0,
                  ln_eps: float = 1e-5, n_out: Optional[int] = None, out_packed=None, out_packed_width: int = 0):
    """Decode-step projection with fused LayerNorm fold / bias / residual / SwiGLU (see lina_gla.h).
    a [M,K] (row stride free), w [N_w,K]; returns out [M, n_out] (n_out defaults to N_w, or to the padded
    SwiGLU width the caller asks for)."""
    be = _backend._BACKEND
    be.require(a, w, c1, c2, resid, out)
    M, K = a.shape
    if w.shape[1] != K or a.stride(1) != 1 or w.stride(1) != 1:
        raise ValueError("a [M,K], w [N,K] with contiguous rows expected")
    N = n_out if n_out is not None else (swiglu_hidden if swiglu_hidden else w.shape[0])
    if out is None:
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    for t in (c1, c2):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise ValueError("c1/c2 must be contiguous fp32 vectors")
    if out_packed is not None:
        be.require(out_packed)
        if out_packed.numel() < packed_numel(M, out_packed_width):
            raise ValueError("packed output buffer is too small")
        _check(be.lib.lina_linear_skinny_ex(_ptr(a), a.stride(0), _ptr(w), w.stride(0), 0, 0, _ptr(c1), _ptr(c2),
                                            _ptr(resid), 0 if resid is None else resid.stride(0), _ptr(out),
                                            out.stride(0), _ptr(out_packed), int(out_packed_width), M, N, K,
                                            swiglu_hidden, ln_dim, float(ln_eps), _dt(a), be.stream(a)))
        return out
    _check(be.lib.lina_linear_skinny(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(c1), _ptr(c2), _ptr(resid),
                                     0 if resid is None else resid.stride(0), _ptr(out), out.stride(0), M, N, K,
                                     swiglu_hidden, ln_dim, float(ln_eps), _dt(a), be.stream(a)))
    return out


def gla_decode_inproj(x, w_in, c1, c2, wq, wk, wv, cq, ck, cv, w2, b2, qkv, g_out, gk, ln_eps: float = 1e-5,
                      normalizer: float = 16.0, clamp_min: Optional[float] = None):
    """LayerNorm-1 + fused projection + conv steps + gate of one GLA mixer at T = 1, one launch
    (lina_gla_decode_inproj, see lina_gla.h)."""
    be = _backend._BACKEND
    be.require(x, w_in, c1, c2, wq, wk, wv, cq, ck, cv, w2, b2, qkv, g_out, gk)
    B, K = x.shape
    Kd, W = wq.shape[0], wq.shape[-1]
    Vd, R = wv.shape[0], w2.shape[1]
    _check(be.lib.lina_gla_decode_inproj(_ptr(x), x.stride(0), _ptr(w_in), w_in.stride(0), _ptr(c1), _ptr(c2),
                                         _ptr(wq), _ptr(wk), _ptr(wv), _ptr(cq), _ptr(ck), _ptr(cv), _ptr(w2),
                                         _ptr(b2), _ptr(qkv), _ptr(g_out), _ptr(gk), B, K, Kd, Vd, W, R,
                                         float(ln_eps), float(normalizer),
                                         float("nan") if clamp_min is None else float(clamp_min), _dt(x),
                                         be.stream(x)))


def gla_decode_inproj_packed(x_packed, w_in_packed, B, K, c1, c2, wq, wk, wv, cq, ck, cv, w2, b2, qkv, g_out, gk,
                             ln_eps: float = 1e-5, normalizer: float = 16.0, clamp_min: Optional[float] = None,
                             w_stream: bool = False):
    """gla_decode_inproj with the block input and the fused projection weight in the fragment-major layout."""
    be = _backend._BACKEND
    be.require(x_packed, w_in_packed, c1, c2, wq, wk, wv, cq, ck, cv, w2, b2, qkv, g_out, gk)
    Kd, W = wq.shape[0], wq.shape[-1]
    Vd, R = wv.shape[0], w2.shape[1]
    if x_packed.numel() < packed_numel(B, K) or w_in_packed.numel() < packed_numel(2 * Kd + 2 * Vd + R, K):
        raise ValueError("packed operand too small")
    _check(be.lib.lina_gla_decode_inproj_packed(_ptr(x_packed), _ptr(w_in_packed), _ptr(c1), _ptr(c2), _ptr(wq), _ptr(wk),
                                                _ptr(wv), _ptr(cq), _ptr(ck), _ptr(cv), _ptr(w2), _ptr(b2), _ptr(qkv),
                                                _ptr(g_out), _ptr(gk), B, K, Kd, Vd, W, R, float(ln_eps),
                                                float(normalizer),
                                                float("nan") if clamp_min is None else float(clamp_min),
                                                1 if w_stream else 0, _dt(x_packed), be.stream(x_packed)))


def gla_decode_update_norm(q, k, v, gk, o_part, state, gate, norm_weight, og, counters, eps: float = 1e-5, scale=None):
    """K1d + K5 in one launch (lina_gla_decode_update_norm): in-place state update and, by the last row-block
    workgroup of each head, partial-sum + RMSNorm (x) swish gate -> og [B,H,Dv].  counters: int32 [B*H] zeros."""
    be = _backend._BACKEND
    be.require(q, k, v, gk, o_part, state, gate, norm_weight, og, counters)
    B, H, Dk = q.shape
    Dv = v.shape[-1]
    if state.dtype != torch.float32 or not state.is_contiguous() or tuple(state.shape) != (B, H, Dk, Dv):
        raise ValueError("state must be contiguous fp32 [B,H,Dk,Dv]")
    if o_part.dtype != torch.float32 or not o_part.is_contiguous() or tuple(o_part.shape) != (Dk // 64, B, H, Dv):
        raise ValueError("o_part must be contiguous fp32 [Dk/64,B,H,Dv]")
    if counters.dtype != torch.int32 or counters.numel() < B * H or not counters.is_contiguous():
        raise ValueError("counters must be a contiguous int32 tensor with B*H entries")
    if not og.is_contiguous() or og.dtype != q.dtype or gate.dtype != q.dtype or gate.stride(-1) != 1:
        raise ValueError("og/gate must be model-dtype tensors, og contiguous, gate row-contiguous")
    _check(be.lib.lina_gla_decode_update_norm(_ptr(q), _ptr(k), _ptr(v), _ptr(gk), _ptr(o_part), _ptr(state), _ptr(gate),
                                              _ptr(norm_weight), _ptr(og), _ptr(counters), B, H, Dk, Dv,
                                              q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0),
                                              v.stride(1), gk.stride(0), gk.stride(1), gate.stride(0), gate.stride(1),
                                              float(eps), _dt(q), _dt(gk),
                                              float(Dk ** -0.5 if scale is None else scale), be.stream(q)))
    return og


def gla_decode_window(q, k, v, gk, state, gate, norm_weight, og, hist_k, hist_c, hist_v, step, origin,
                      window: int, eps: float = 1e-5, scale=None, og_packed: bool = False, o_exchange=None, counters=None):
    """K1w + K5 (lina_gla_decode_window): decode-step update with a lazily written state -- ``state`` is read every
    step and rewritten every ``window``-th one, the steps in between live in hist_k / hist_c [window,B*H,Dk] and
    hist_v [window,B*H,Dv] (fp32).  ``step`` / ``origin``: int64 device tensors (window position = (step-origin) %
    window).  Call gla_decode_window_flush before anybody else reads ``state``.  ``state``: fp32, or (opt-in, bf16 activations
    only) bf16 -- the reference's state dtype for a bf16 model, rounded at every write-back (lina_gla_decode_window_s)."""
    be = _backend._BACKEND
    be.require(q, k, v, gk, state, gate, norm_weight, og, hist_k, hist_c, hist_v, step, origin)
    B, H, Dk = q.shape
    Dv = v.shape[-1]
    if state.dtype not in (torch.float32, torch.bfloat16) or not state.is_contiguous() or tuple(state.shape) != (B, H, Dk, Dv):
        raise ValueError("state must be contiguous fp32 (or, opt-in, bf16) [B,H,Dk,Dv]")
    if state.dtype == torch.bfloat16 and q.dtype != torch.bfloat16:
        raise ValueError("a bf16 state needs bf16 activations")
    for t, shp in ((hist_k, (window, B * H, Dk)), (hist_c, (window, B * H, Dk)), (hist_v, (window, B * H, Dv))):
        if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != shp:
            raise ValueError(f"history buffers must be contiguous fp32 {shp}")
    if step.dtype != torch.int64 or origin.dtype != torch.int64:
        raise ValueError("step / origin must be int64 device tensors")
    if not og.is_contiguous() or og.dtype != q.dtype or gate.dtype != q.dtype or gate.stride(-1) != 1:
        raise ValueError("og/gate must be model-dtype tensors, og contiguous, gate row-contiguous")
    if og_packed and og.numel() < packed_numel(B, H * Dv):
        raise ValueError("packed og buffer is too small")
    for t in (q, k, v, gk):
        if t.stride(-1) != 1:
            raise ValueError("innermost dimension must be contiguous")
    be.require(o_exchange, counters)
    if Dv > 256 and (o_exchange is None or counters is None or o_exchange.dtype != torch.float32
                     or o_exchange.numel() < B * H * Dv or counters.dtype != torch.int32 or counters.numel() < B * H):
        raise ValueError("Dv > 256 needs o_exchange (fp32 [B*H*Dv]) and counters (int32 [B*H], zero)")
    _check(be.lib.lina_gla_decode_window_s(_ptr(q), _ptr(k), _ptr(v), _ptr(gk), _ptr(state), _dt(state), _ptr(gate),
                                         _ptr(norm_weight), _ptr(og), _ptr(o_exchange), _ptr(counters), _ptr(hist_k),
                                         _ptr(hist_c),
                                         _ptr(hist_v), _ptr(step), _ptr(origin), int(window), B, H, Dk, Dv,
                                         q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                                         gk.stride(0), gk.stride(1), gate.stride(0), gate.stride(1), float(eps),
                                         1 if og_packed else 0, _dt(q), _dt(gk),
                                         float(Dk ** -0.5 if scale is None else scale), be.stream(q)))
    return og


def gla_decode_window_flush(state, hist_k, hist_c, hist_v, n_pending: int):
    """Apply the first ``n_pending`` steps of the current window to ``state`` (in place)."""
    be = _backend._BACKEND
    be.require(state, hist_k, hist_c, hist_v)
    B, H, Dk, Dv = state.shape
    _check(be.lib.lina_gla_decode_window_flush_s(_ptr(state), _dt(state), _ptr(hist_k), _ptr(hist_c), _ptr(hist_v),
                                                 int(n_pending), B, H, Dk, Dv, be.stream(state)))
    return state


def cross_att_step1(q_lin, ln_w, ln_b, ln_eps, kk, pe, att1, xp, scale):
    """Blind cross-attention step 1 (see lina_gla.h).  att1: [B,T_txt] view (row stride free), written in place."""
    be = _backend._BACKEND
    be.require(q_lin, ln_w, ln_b, kk, pe, att1, xp)
    B, d = q_lin.shape
    Tn = kk.shape[1]
    _check(be.lib.lina_cross_att_step1(_ptr(q_lin), _ptr(ln_w), _ptr(ln_b), float(ln_eps), _ptr(kk), _ptr(pe),
                                       _ptr(att1), att1.stride(0), _ptr(xp), B, Tn, d, float(scale), _dt(q_lin),
                                       be.stream(q_lin)))


def cross_att_step2(xp, pe, vv, att2, x, scale):
    """Blind cross-attention step 2: x += softmax(xp . pe^T * scale) . vv   (see lina_gla.h)."""
    be = _backend._BACKEND
    be.require(xp, pe, vv, att2, x)
    B, d = xp.shape
    Tn = vv.shape[1]
    _check(be.lib.lina_cross_att_step2(_ptr(xp), _ptr(pe), _ptr(vv), _ptr(att2), att2.stride(0), _ptr(x), B, Tn, d,
                                       float(scale), _dt(xp), be.stream(xp)))


def cross_scores(q_lin, ln_w, ln_b, ln_eps, kk, scores, scale):
    """scores[b,t] = scale * <LayerNorm(q_lin[b]), kk[b,t,:]> (fp32 [B,T_txt]); see lina_gla.h."""
    be = _backend._BACKEND
    be.require(q_lin, ln_w, ln_b, kk, scores)
    B, d = q_lin.shape
    _check(be.lib.lina_cross_scores(_ptr(q_lin), _ptr(ln_w), _ptr(ln_b), float(ln_eps), _ptr(kk), _ptr(scores), B,
                                    kk.shape[1], d, float(scale), _dt(q_lin), be.stream(q_lin)))


def cross_scores_softmax(q_lin, ln_w, ln_b, ln_eps, kk, att, attc, scale):
    """att[b,:Tn] = softmax(scale * <LayerNorm(q_lin[b]), kk[b,t,:]>) into the strided ``att`` rows and the contiguous
    zero-padded copy attc [B,Tp] -- cross_scores + softmax_rows in one launch (lina_cross_scores_softmax)."""
    be = _backend._BACKEND
    be.require(q_lin, ln_w, ln_b, kk, att, attc)
    B, d = q_lin.shape
    Tn = kk.shape[1]
    _check(be.lib.lina_cross_scores_softmax(_ptr(q_lin), _ptr(ln_w), _ptr(ln_b), float(ln_eps), _ptr(kk), _ptr(att),
                                            att.stride(0), _ptr(attc), B, Tn, attc.shape[1], d, float(scale), _dt(q_lin),
                                            be.stream(q_lin)))


def softmax_weighted_rows_add(scores, scale, att, vv, x, x_packed=None):
    """att[b,:Tn] = softmax(scores[b,:Tn] * scale);  x[b,:] += att[b,:] . vv[b]  -- softmax_rows + weighted_rows_add in
    one launch.  With ``x_packed`` the residual stream is the fragment-major buffer (``x`` is not touched)."""
    be = _backend._BACKEND
    be.require(scores, att, vv, x, x_packed)
    B, Tn, d = vv.shape
    if scores.dtype != vv.dtype or att.dtype != vv.dtype:
        raise TypeError("scores / att / vv must share the model dtype")
    _check(be.lib.lina_softmax_weighted_rows_add(_ptr(scores), scores.stride(0), float(scale), _ptr(att), att.stride(0),
                                                 _ptr(vv), _ptr(x), _ptr(x_packed), B, Tn, d, _dt(vv), be.stream(vv)))


def _att_log_args(att, att_step, att_step_stride, att_steps):
    """(step pointer, stride, number of steps) of the att-log form of the cross-attention launches (lina_gla.h)."""
    if att_step is None:
        return None, 0, 0
    if att_step.dtype != torch.int64 or att_step.numel() < 1:
        raise TypeError("att_step must be a device int64 step counter")
    return _ptr(att_step), int(att_step_stride), int(att_steps)


def pe_softmax_weighted_rows_add(xp, pe, scale, att, vv, x, x_packed=None, xp_is_packed: bool = False,
                                 att_step=None, att_step_stride: int = 0, att_steps: int = 0):
    """The last two launches of the cross-attention step as one (round 4):  sc[b,t] = <xp[b,:], pe[t,:]> rounded to the model
    dtype, att[b,:Tn] = softmax(sc[b,:Tn] * scale), x[b,:] += att[b,:] . vv[b].  ``xp``: [B,d] row-major, or the
    fragment-major buffer when ``xp_is_packed``; ``x_packed``: the residual stream in fragment-major form."""
    be = _backend._BACKEND
    be.require(xp, pe, att, vv, x, x_packed)
    B, Tn, d = vv.shape
    if pe.dtype != vv.dtype or att.dtype != vv.dtype or xp.dtype != vv.dtype:
        raise TypeError("xp / pe / att / vv must share the model dtype")
    if pe.shape[0] < Tn or pe.shape[1] != d or pe.stride(1) != 1 or pe.stride(0) != d:
        raise ValueError("pe must be a contiguous [>= T_txt, d] table")
    be.require(att_step)
    _check(be.lib.lina_pe_softmax_weighted_rows_add(_ptr(xp), 1 if xp_is_packed else 0, _ptr(pe), float(scale), _ptr(att),
                                                    att.stride(0), *_att_log_args(att, att_step, att_step_stride, att_steps),
                                                    _ptr(vv), _ptr(x), _ptr(x_packed), B, Tn, d, _dt(vv),
                                                    be.stream(vv)))


def softmax_pe_rows(scores, att, pe, xp, xp_packed=None, att_step=None, att_step_stride: int = 0, att_steps: int = 0):
    """att[b,:Tn] = softmax(scores[b,:Tn]) (fp32 scores, already scaled);  xp[b,:] = att[b,:] . pe[:Tn,:] -- one launch;
    ``xp_packed``: also the fragment-major copy of xp (the A operand of the next projection)."""
    be = _backend._BACKEND
    be.require(scores, att, pe, xp, xp_packed)
    B, Tn = scores.shape
    d = pe.shape[1]
    if scores.dtype != torch.float32 or scores.stride(1) != 1:
        raise TypeError("scores must be fp32 [B, Tn] with contiguous rows")
    if att.dtype != pe.dtype or xp.dtype != pe.dtype or pe.shape[0] < Tn or not pe.is_contiguous() or not xp.is_contiguous():
        raise TypeError("att / pe / xp must share the model dtype; pe [>= Tn, d] and xp [B, d] contiguous")
    if xp_packed is not None and xp_packed.numel() < packed_numel(B, d):
        raise ValueError("packed xp buffer is too small")
    be.require(att_step)
    _check(be.lib.lina_softmax_pe_rows(_ptr(scores), scores.stride(0), _ptr(att), att.stride(0),
                                       *_att_log_args(att, att_step, att_step_stride, att_steps), _ptr(pe), _ptr(xp),
                                       _ptr(xp_packed), B, Tn, d, _dt(pe), be.stream(pe)))


def softmax_rows(x, scale, att, attc, Tn):
    """att[b,:Tn] = softmax(x[b,:Tn]*scale) into the strided `att` rows and the contiguous padded copy attc [B,Tp]."""
    be = _backend._BACKEND
    be.require(x, att, attc)
    B = x.shape[0]
    _check(be.lib.lina_softmax_rows(_ptr(x), x.stride(0), _dt(x), float(scale), _ptr(att), att.stride(0), _ptr(attc), B,
                                    Tn, attc.shape[1], _dt(attc), be.stream(x)))


def weighted_rows_add(attc, vv, x, x_packed=None):
    """x[b,:] += sum_t attc[b,t] * vv[b,t,:].  With ``x_packed`` the residual stream is the fragment-major buffer: it is
    updated in place there and ``x`` is not touched."""
    be = _backend._BACKEND
    be.require(attc, vv, x, x_packed)
    B, Tn, d = vv.shape
    if x_packed is None:
        _check(be.lib.lina_weighted_rows_add(_ptr(attc), attc.shape[1], _ptr(vv), _ptr(x), B, Tn, d, _dt(x), be.stream(x)))
        return
    if x_packed.numel() < packed_numel(B, d):
        raise ValueError("packed x buffer is too small")
    _check(be.lib.lina_weighted_rows_add_packed(_ptr(attc), attc.shape[1], _ptr(vv), _ptr(x), _ptr(x_packed), B, Tn, d,
                                                _dt(x), be.stream(x)))


# --------------------------------------------------------------------------- codes -> waveform (f-3)
def dwconv7_ln(x, weight, bias=None, scale=None, shift=None, eps: float = 1e-6):
    """K8: depthwise conv (k = 7, 'same') + LayerNorm over channels, channels-last ``x [B,L,C]``
    (ConvNeXtBlock.dwconv + norm, reference 3rdparty/decoder/modules.py:44-50).  ``weight`` [C,1,7]|[C,7];
    ``scale`` / ``shift``: [C] (LayerNorm affine) or [B,C] (AdaLayerNorm rows) or None."""
    _no_grad(x, weight, bias, scale, shift)
    be = _backend._BACKEND
    be.require(x, weight, bias, scale, shift)
    B, L, Cc = x.shape
    x = x.contiguous()
    w = weight.reshape(Cc, 7).to(x.dtype).contiguous()
    b = None if bias is None else bias.to(x.dtype).contiguous()
    sb = 0
    if scale is not None:
        scale = scale.to(x.dtype).contiguous()
        sb = Cc if scale.dim() == 2 and scale.shape[0] == B and B > 1 else 0
        if scale.dim() == 2 and scale.shape[0] not in (1, B):
            raise ValueError("scale must be [C], [1,C] or [B,C]")
    if shift is not None:
        shift = shift.to(x.dtype).contiguous()
        if scale is not None and tuple(shift.shape) != tuple(scale.shape):
            raise ValueError("scale and shift must have the same shape")
        if scale is None:
            sb = Cc if shift.dim() == 2 and shift.shape[0] == B and B > 1 else 0
    y = torch.empty_like(x)
    _check(be.lib.lina_dwconv7_ln(_ptr(x), _ptr(w), _ptr(b), _ptr(scale), _ptr(shift), _ptr(y), B, L, Cc, sb, float(eps),
                                  _dt(x), be.stream(x)))
    return y


def istft_ola(frames, window, hop: int):
    """K9: windowed overlap-add + envelope normalisation with 'same' padding (reference spectral_ops.py:56-75).
    ``frames`` fp32 [B,T,win] inverse-transformed frames, ``window`` fp32 [win] -> fp32 [B, T*hop] (win - hop even)."""
    be = _backend._BACKEND
    be.require(frames, window)
    B, T, win = frames.shape
    frames = frames.float().contiguous()
    window = window.float().contiguous()
    pad = (win - hop) // 2
    y = torch.empty(B, (T - 1) * hop + win - 2 * pad, dtype=torch.float32, device=frames.device)
    _check(be.lib.lina_istft_ola(_ptr(frames), _ptr(window), _ptr(y), B, T, win, int(hop), be.stream(frames)))
    return y
