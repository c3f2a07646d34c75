"""The differentiable operators: the names the reference imports from ``fla`` (chunk_gla, fused_chunk_gla,
fused_recurrent_gla, ...) and the fused training-path ops (short conv, norm-gate, LayerNorm + residual, SwiGLU, linear with a
token-split weight gradient, gate projections, cross-entropy, embedding sum) as ``torch.autograd.Function``s over the launchers
of kernels.py."""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch.utils.weak import WeakIdKeyDictionary

from . import _lib
from . import backend as _backend
from .backend import _check, _dt, _inner_contig, _ptr, fused_ops_available
from .kernels import (
    _gla_prepare, _gla_launch, gla_chunk_bwd, _short_conv_launch, _sum_partials, _sum_partials2, column_sum,
    _sum_vector, _embed_sum_launch)
from .policy import POLICY, _MLP_PAD, _linear_split


class _GLAFunction(torch.autograd.Function):
    """K2 forward + K2b backward (the training path of reference model/gla.py:193,195).  The forward of a
    call that needs gradients always takes the chunk kernel, whichever fla name it was reached through:
    the recurrence is the same and K2b recomputes the states chunk-wise."""

    @staticmethod
    def forward(ctx, q, k, v, gk, scale, initial_state, output_final_state, nseg=None):
        kept: list = []
        o, ht = _gla_launch("lina_gla_chunk_fwd", q, k, v, gk, scale, initial_state, output_final_state, nseg=nseg,
                            keep_seg_states=kept)
        ctx.nseg = kept[0][1] if kept else nseg
        ctx.save_for_backward(q, k, v, gk, initial_state, ht, *[ws for ws, _ in kept])   # one workspace per value block
        ctx.scale = scale
        ctx.need_dh0 = initial_state is not None and initial_state.requires_grad
        if ht is None:
            return o, None
        return o, ht

    @staticmethod
    def backward(ctx, d_o, d_ht):
        q, k, v, gk, h0, ht, *seg_ws = ctx.saved_tensors
        seg_ws = None if not seg_ws else (seg_ws[0] if len(seg_ws) == 1 else seg_ws)
        if d_o is None:                                   # only the final state was used downstream
            d_o = torch.zeros(q.shape[0], q.shape[2], q.shape[1], v.shape[-1], dtype=q.dtype,
                              device=q.device).transpose(1, 2)
        dq, dk, dv, dg, dh0 = gla_chunk_bwd(q, k, v, gk, d_o, ctx.scale, h0, ht, d_ht, ctx.need_dh0, nseg=ctx.nseg,
                                            seg_states=seg_ws)
        if dh0 is not None and h0 is not None and dh0.dtype != h0.dtype:
            dh0 = dh0.to(h0.dtype)
        return dq, dk, dv, dg, None, dh0, None, None


def _needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _gla(entry: str, q, k, v, gk, scale, initial_state, output_final_state, inplace_state=False, nseg=None):
    q, k, v, gk, scale = _gla_prepare(q, k, v, gk, scale, initial_state)
    if _needs_grad(q, k, v, gk, initial_state):
        return _GLAFunction.apply(q, k, v, gk, scale, initial_state, bool(output_final_state), nseg)
    return _gla_launch(entry, q, k, v, gk, scale, initial_state, output_final_state, inplace_state, nseg)


def fused_recurrent_gla(q, k, v, gk, scale=None, initial_state=None, output_final_state=False,
                        inplace_state: bool = False):
    """fla.ops.gla.fused_recurrent_gla (reference call sites model/gla.py:188,190,201) -> K1.
    ``inplace_state=True`` updates ``initial_state`` in place and returns it as the final state."""
    return _gla("lina_gla_recurrent_fwd", q, k, v, gk, scale, initial_state, output_final_state, inplace_state)


def naive_recurrent_gla(q, k, v, gk, initial_state=None, output_final_state=False):
    """fla.ops.gla.naive.naive_recurrent_gla (reference model/gla.py:197): same recurrence -> K1."""
    return _gla("lina_gla_recurrent_fwd", q, k, v, gk, None, initial_state, output_final_state)


def chunk_gla(q, k, v, g, scale=None, initial_state=None, output_final_state=False, nseg=None):
    """fla.ops.gla.chunk_gla (reference model/gla.py:195) -> K2 (MFMA chunk scan).  ``nseg`` (not an fla argument)
    forces the number of concurrent sequence segments; default: chunk_segments(B*H, T)."""
    return _gla("lina_gla_chunk_fwd", q, k, v, g, scale, initial_state, output_final_state, nseg=nseg)


def fused_chunk_gla(q, k, v, g, scale=None, initial_state=None, output_final_state=False):
    """fla.ops.gla.fused_chunk_gla (reference model/gla.py:193; the mixer's default mode) -> K2."""
    return _gla("lina_gla_chunk_fwd", q, k, v, g, scale, initial_state, output_final_state)


def chunk_simple_gla(q, k, v, g, scale=None, initial_state=None, output_final_state=False):
    """fla.ops.simple_gla.chunk_simple_gla: scalar gate per head g [B,H,T] -> K2 with the gate
    broadcast over Dk."""
    gk = g.unsqueeze(-1).expand(*g.shape, q.shape[-1])
    return _gla("lina_gla_chunk_fwd", q, k, v, gk.contiguous(), scale, initial_state, output_final_state)


class GradSlab:
    """Backward-time buffer [..., sum(sizes)] for the output gradient of a stacked projection: the consumers of its column
    slices write their input gradients straight into their columns (``part``), so the projection's backward finds dZ
    assembled -- torch's split backward concatenated the pieces in one more pass over all of them."""

    def __init__(self, lead_shape, sizes, dtype, device):
        self.lead_shape, self.sizes, self.dtype, self.device = tuple(lead_shape), list(sizes), dtype, device
        self.offsets = [sum(self.sizes[:i]) for i in range(len(self.sizes))]
        self.buf = None
        self.copied = []                   # slices the last backward had to copy in (not written in place): diagnostics

    def part(self, i):
        if self.buf is None:
            self.buf = torch.empty(*self.lead_shape, sum(self.sizes), dtype=self.dtype, device=self.device)
        return self.buf[..., self.offsets[i]:self.offsets[i] + self.sizes[i]]

    def take(self):
        buf, self.buf = self.buf, None
        return buf


def _slab_part(grad_slab, like):
    """The slab columns for a gradient shaped like ``like`` ([..., size] with the slab's leading shape), or None."""
    slab, i = grad_slab
    if (like.dtype != slab.dtype or like.device != slab.device or like.shape[-1] != slab.sizes[i]
            or tuple(like.shape[:-1]) != slab.lead_shape):
        return None
    return slab.part(i)


class _SplitSlabFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, slab):
        ctx.slab = slab
        ctx.set_materialize_grads(False)
        return tuple(z.split(slab.sizes, dim=-1))

    @staticmethod
    def backward(ctx, *grads):
        slab = ctx.slab
        slab.copied = []
        for i, g in enumerate(grads):
            part = slab.part(i)
            if g is None:
                part.zero_()
            elif not (g.data_ptr() == part.data_ptr() and g.shape == part.shape and g.stride() == part.stride()):
                part.copy_(g)                      # a consumer that did not write in place (or an accumulated gradient)
                slab.copied.append(i)
        return slab.take(), None


def split_slab(z, sizes):
    """``z.split(sizes, -1)`` plus the ``GradSlab`` its consumers may write their input gradients into (``grad_slab=(slab,
    i)`` of ``short_conv`` / ``rmsnorm_swish_gate``); slices whose consumers do not are copied in by the backward.  Without
    gradients (or off the fused-op devices) this is the plain split and the slab is None."""
    if not (torch.is_grad_enabled() and z.requires_grad and fused_ops_available(z)):
        return z.split(list(sizes), dim=-1), None
    slab = GradSlab(z.shape[:-1], sizes, z.dtype, z.device)
    return _SplitSlabFunction.apply(z, slab), slab


class _ShortConvFunction(torch.autograd.Function):
    """K3 forward + K3b backward (cache-less prefill form, the training path)."""

    @staticmethod
    def forward(ctx, x, w, bias, mask, act, cache=None, grad_slab=None):
        # w / bias arrive in the PARAMETER dtype (fp32 master weights under autocast): cast here, once, outside autograd --
        # the gradients leave in the parameter dtype straight from the fp32 partial sums
        ctx.w_dtype, ctx.b_dtype = w.dtype, (None if bias is None else bias.dtype)
        w = w.to(x.dtype).contiguous()
        bias = None if bias is None else bias.to(x.dtype).contiguous()
        ctx.save_for_backward(x, w, bias, mask)
        ctx.act = act
        ctx.grad_slab = grad_slab
        # a cache given to the prefill form only RECEIVES the last W inputs (training with an initial state,
        # reference model/gla.py:146-163 with use_cache=True): it does not enter y, so the backward is the same
        return _short_conv_launch(x, w, bias, mask, cache, act)

    @staticmethod
    def backward(ctx, dy):
        x, w, bias, mask = ctx.saved_tensors
        B, T, D = x.shape
        W = w.shape[1]
        be = _backend._BACKEND
        dy = _inner_contig(dy.to(x.dtype))
        dx = _slab_part(ctx.grad_slab, x) if ctx.grad_slab is not None else None
        if dx is None:
            dx = torch.empty(B, T, D, dtype=x.dtype, device=x.device)
        nblk = B * ((T + _lib.CONV_BWD_TT - 1) // _lib.CONV_BWD_TT)
        part = torch.empty(nblk, D, W + 1, dtype=torch.float32, device=x.device)
        _check(be.lib.lina_short_conv_bwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(mask), _ptr(dy), _ptr(dx), _ptr(part),
                                          B, T, D, W, x.stride(0), x.stride(1), dy.stride(0), dy.stride(1),
                                          dx.stride(0), dx.stride(1), ctx.act, _dt(x), be.stream(x)))
        red = _sum_partials(part)
        dw = red[:, :W].to(ctx.w_dtype, copy=True)               # contiguous [D, W] in the parameter's dtype
        db = None if bias is None else red[:, W].to(ctx.b_dtype, copy=True)
        return dx, dw, db, None, None, None, None


def short_conv(x, weight, bias=None, mask=None, cache=None, activation: Optional[str] = "silu", grad_slab=None):
    """ShortConvolution.forward semantics: x [B,T,D], weight [D,1,W]|[D,W],
    mask [B,T]|None, cache [B,D,W]|None (mutated in place).  Differentiable when cache is None.
    ``grad_slab``: ``(GradSlab, index)`` when ``x`` is column slice ``index`` of a stacked projection (``split_slab``):
    the backward then writes dx into the slab in place."""
    B, T, D = x.shape
    w = weight.reshape(D, -1)
    W = w.shape[1]
    be = _backend._BACKEND
    be.require(x, w, bias, mask, cache)
    x = _inner_contig(x)
    w_param, bias_param = w, bias
    act = 1 if activation in ("silu", "swish") else 0
    if activation not in ("silu", "swish", None):
        raise ValueError(f"activation {activation!r} not supported")
    user_cache = None
    if cache is not None:
        if tuple(cache.shape) != (B, D, W) or not cache.is_contiguous():
            raise ValueError(f"cache must be a contiguous tensor [B,D,W]={B, D, W}")
        if cache.dtype != x.dtype:
            # e.g. an fp32 cache from init_state() with bf16 activations under autocast: the reference's
            # cache.copy_(...) casts; run on a cache of the activation dtype and cast back into the caller's tensor
            user_cache, cache = cache, cache.to(x.dtype)
    m = None if mask is None else mask.to(torch.float32).contiguous()
    if _needs_grad(x, w, bias):
        if cache is not None and T == 1:
            raise NotImplementedError("short_conv: gradients are built for the prefill form (T > 1 or no cache) only")
        y = _ShortConvFunction.apply(x, w_param, bias_param, m, act, cache, grad_slab)
    else:
        if cache is not None and T == 1:
            m = mask
        w = w.to(x.dtype).contiguous()
        bias = None if bias is None else bias.to(x.dtype).contiguous()
        y = _short_conv_launch(x, w, bias, m, cache, act)
    if user_cache is not None:
        user_cache.copy_(cache)
    return y


def _adjacent_columns(ts):
    """True when the tensors [B, T, D_i] are consecutive column slices of one row-major buffer (same strides, inner
    contiguous, each starting where the previous one ends)."""
    a = ts[0]
    if a.dim() != 3 or a.stride(2) != 1:
        return False
    off = a.data_ptr()
    for t in ts:
        if (t.dim() != 3 or t.dtype != a.dtype or t.device != a.device or t.shape[:2] != a.shape[:2]
                or t.stride() != a.stride() or t.data_ptr() != off):
            return False
        off += t.shape[2] * t.element_size()
    return True


def _stack_rows(parts, dtype, pad: int = 0):
    """``torch.cat(parts, 0)`` (+ ``pad`` zero rows) cast to ``dtype`` -- ONE pass on K16 ``lina_stack_rows`` when the blocks are
    contiguous, 16-byte aligned fp32 [r_i, cols] with cols % 4 == 0 (master weights), else the torch expression."""
    cols = parts[0].shape[1]
    if (POLICY.one_pass_operands and dtype in (torch.float32, torch.bfloat16) and 1 <= len(parts) <= 8 and cols % 4 == 0
            and fused_ops_available(parts[0])
            and all(p.dtype == torch.float32 and p.dim() == 2 and p.shape[1] == cols and p.is_contiguous()
                    and p.data_ptr() % 16 == 0 for p in parts)):
        import ctypes as C
        be = _backend._BACKEND
        rows = [int(p.shape[0]) for p in parts]
        out = torch.empty(sum(rows) + pad, cols, dtype=dtype, device=parts[0].device)
        srcs = (C.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
        _check(be.lib.lina_stack_rows(srcs, (C.c_int * len(parts))(*rows), len(parts), cols, out.shape[0], _ptr(out), _dt(out),
                                      be.stream(out)))
        return out
    w = [p.detach() for p in parts]
    if pad:
        w.append(parts[0].new_zeros(pad, cols))
    return torch.cat(w, dim=0).to(dtype).contiguous()


class _ShortConv3Function(torch.autograd.Function):
    """K3 / K3b over the q | k | v column slices of a stacked projection in ONE launch each way: the three slices are
    adjacent columns of the same rows (6 KB contiguous per token at L169 instead of three 2 KB pieces in three launches), the
    three depthwise filters are stacked to one [sum D_i, W] filter, the outputs are column slices of one buffer (the chunk
    kernels take them strided) and so are the incoming gradients when K2b produced them (kernels.gla_chunk_bwd)."""

    @staticmethod
    def forward(ctx, x0, x1, x2, w0, w1, w2, b0, b1, b2, mask, act, grad_slab):
        xs, ws, bs = (x0, x1, x2), (w0, w1, w2), (b0, b1, b2)
        B, T, _ = x0.shape
        sizes = [x.shape[2] for x in xs]
        D = sum(sizes)
        ctx.w_dtypes = [w.dtype for w in ws]
        ctx.b_dtypes = [None if b is None else b.dtype for b in bs]
        w = _stack_rows([wi.reshape(wi.shape[0], -1) for wi in ws], x0.dtype)   # the three filters as one, one pass (K16)
        bias = None if b0 is None else torch.cat(bs, dim=0).to(x0.dtype).contiguous()
        x = x0.as_strided((B, T, D), x0.stride())              # the three slices as one [B, T, D] view of the same rows
        ctx.save_for_backward(x, w, bias, mask)
        ctx.act, ctx.sizes, ctx.grad_slab = act, sizes, grad_slab
        y = _short_conv_launch(x, w, bias, mask, None, act)
        return tuple(y.split(sizes, dim=-1))

    @staticmethod
    def backward(ctx, *dys):
        x, w, bias, mask = ctx.saved_tensors
        B, T, D = x.shape
        W = w.shape[1]
        sizes = ctx.sizes
        be = _backend._BACKEND
        dys = [torch.zeros(B, T, n, dtype=x.dtype, device=x.device) if g is None else g.to(x.dtype)
               for g, n in zip(dys, sizes)]
        dy = dys[0].as_strided((B, T, D), dys[0].stride()) if _adjacent_columns(dys) else torch.cat(dys, dim=-1)
        dx = None
        if ctx.grad_slab is not None:                           # columns 0 .. D of the projection's gradient slab
            slab, first = ctx.grad_slab
            parts = [slab.part(first + i) for i in range(3)]
            if (all(p.dtype == x.dtype and p.device == x.device and p.shape == (B, T, n) for p, n in zip(parts, sizes))
                    and _adjacent_columns(parts)):
                dx = parts[0].as_strided((B, T, D), parts[0].stride())
                parts_out = parts
        if dx is None:
            dx = torch.empty(B, T, D, dtype=x.dtype, device=x.device)
            parts_out = list(dx.split(sizes, dim=-1))
        nblk = B * ((T + _lib.CONV_BWD_TT - 1) // _lib.CONV_BWD_TT)
        part = torch.empty(nblk, D, W + 1, dtype=torch.float32, device=x.device)
        _check(be.lib.lina_short_conv_bwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(mask), _ptr(dy), _ptr(dx), _ptr(part),
                                          B, T, D, W, x.stride(0), x.stride(1), dy.stride(0), dy.stride(1),
                                          dx.stride(0), dx.stride(1), ctx.act, _dt(x), be.stream(x)))
        red = _sum_partials(part)
        dws, dbs, o = [], [], 0
        rw = red[:, :W].contiguous()                           # ONE compaction: the filters' gradients are row ranges of it
        for n, wdt, bdt in zip(sizes, ctx.w_dtypes, ctx.b_dtypes):
            dws.append(rw[o:o + n].to(wdt))
            dbs.append(None if bdt is None else red[o:o + n, W].to(bdt, copy=True))
            o += n
        return (*parts_out, *dws, *dbs, None, None, None)


def short_conv3(xs, weights, biases, mask=None, activation: Optional[str] = "silu", grad_slab=None):
    """``[short_conv(x_i, w_i, b_i, mask) for i in 0..2]`` for three column slices ``xs`` of one stacked projection (the
    mixer's q | k | v, reference model/gla.py:161-163) in one launch each way, or None when the fused form does not apply
    (the caller then runs the three convolutions).  ``grad_slab``: ``(GradSlab, index of xs[0])``."""
    if not (len(xs) == 3 and _adjacent_columns(xs) and fused_ops_available(xs[0])):
        return None
    if activation not in ("silu", "swish", None):
        return None
    ws = [w.reshape(w.shape[0], -1) for w in weights]
    if len({w.shape[1] for w in ws}) != 1 or any(w.shape[0] != x.shape[2] for w, x in zip(ws, xs)):
        return None
    if any(b is None for b in biases) != all(b is None for b in biases):
        return None
    if not _needs_grad(*xs, *ws, *biases):
        return None
    be = _backend._BACKEND
    be.require(*xs, *ws, *biases, mask)
    m = None if mask is None else mask.to(torch.float32).contiguous()
    act = 1 if activation in ("silu", "swish") else 0
    return _ShortConv3Function.apply(*xs, *ws, *biases, m, act, grad_slab)


# --------------------------------------------------------------------------- norm (K5)
class _RMSNormGateFunction(torch.autograd.Function):
    """K5 forward + K5b backward on contiguous rows x [rows, D]; the gate is [rows, D] or a strided [R, H, D] view (head
    slices of wider rows, rows = R H) read in place, its gradient written the same way (into a GradSlab if given)."""

    @staticmethod
    def _gate_strides(g, D):
        if g is None or g.dim() == 2:
            return 1, D, 0
        return g.shape[1], g.stride(0), g.stride(1)

    @staticmethod
    def forward(ctx, x, g, w, eps, grad_slab=None):
        be = _backend._BACKEND
        rows, D = x.shape
        ctx.w_dtype = None if w is None else w.dtype       # the PARAMETER dtype: cast here, gradient returned in it
        w = None if w is None else w.to(x.dtype).contiguous()
        y = torch.empty_like(x)
        ri, go, gi = _RMSNormGateFunction._gate_strides(g, D)
        _check(be.lib.lina_rmsnorm_gate_fwd(_ptr(x), _ptr(g), _ptr(w), _ptr(y), rows, ri, D, D * ri, D if ri > 1 else 0,
                                            go, gi, D * ri, D if ri > 1 else 0, 1, 0, eps, _dt(x), _dt(y), be.stream(x)))
        ctx.save_for_backward(x, g, w)
        ctx.eps = eps
        ctx.grad_slab = grad_slab
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, w = ctx.saved_tensors
        be = _backend._BACKEND
        rows, D = x.shape
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        dg = None
        ri, go, gi = _RMSNormGateFunction._gate_strides(g, D)
        dgo, dgi = D * ri, (D if ri > 1 else 0)
        if g is not None:
            if ctx.grad_slab is not None and g.dim() == 3:
                slab, i = ctx.grad_slab
                if g.dtype == slab.dtype and slab.sizes[i] == ri * D and math.prod(slab.lead_shape) == g.shape[0]:
                    dg = slab.part(i).view(g.shape)              # [lead..., H D] columns of the slab as [R, H, D]
                    dgo, dgi = dg.stride(0), dg.stride(1)
            if dg is None:
                dg = torch.empty(g.shape, dtype=g.dtype, device=g.device)
        npart = int(be.lib.lina_rmsnorm_gate_bwd_partials(rows))
        part = torch.empty(npart, D, dtype=torch.float32, device=x.device)
        _check(be.lib.lina_rmsnorm_gate_bwd(_ptr(x), _ptr(g), _ptr(w), _ptr(dy), _ptr(dx), _ptr(dg), _ptr(part),
                                            rows, ri, D, go, gi, dgo, dgi, ctx.eps, _dt(x), be.stream(x)))
        dw = None if w is None else _sum_partials(part, ctx.w_dtype)
        return dx, dg, dw, None, None


def _gate_rows_view(g, D):
    """``g`` [..., H, D] as a [R, H, D] VIEW with 4-element-aligned strides (head slices of wider rows), or None."""
    H = g.shape[-2]
    if g.stride(-1) != 1 or g.stride(-2) % 4 or g.is_contiguous():
        return None
    try:
        v = g.view(-1, H, D)
    except RuntimeError:
        return None
    # the kernels read / write the gate (and its gradient) with 4-element vector accesses: strides AND the base must allow it
    return v if (v.stride(0) % 4 == 0 and v.data_ptr() % (4 * v.element_size()) == 0) else None


def rmsnorm_swish_gate(x, g=None, weight=None, eps: float = 1e-5, n_partial: int = 1, out_dtype=None, out=None,
                       grad_slab=None):
    """FusedRMSNormSwishGate / RMSNorm forward over the last dim.
    ``n_partial`` > 1: ``x`` is [n_partial, ..., D] partial sums (fp32) that are added first.
    A gate ``g`` whose rows are head slices of wider rows ([..., H, D] view of a column slice) is read in place.
    ``grad_slab``: ``(GradSlab, index)`` when ``g`` is column slice ``index`` of a stacked projection (``split_slab``)."""
    be = _backend._BACKEND
    be.require(x, g, weight)
    if _needs_grad(x, g, weight):
        if n_partial != 1 or out is not None:
            raise NotImplementedError("rmsnorm_swish_gate: gradients are built for the plain (n_partial=1) form only")
        odt = out_dtype or (g.dtype if g is not None else x.dtype)
        D = x.shape[-1]
        x2 = x.to(odt).reshape(-1, D).contiguous()
        g2 = None
        if g is not None:
            g2 = g.to(odt)
            g3 = _gate_rows_view(g2, D) if g2.shape == x.shape and g2.dim() >= 3 else None
            g2 = g3 if g3 is not None else g2.reshape(-1, D).contiguous()
        return _RMSNormGateFunction.apply(x2, g2, weight, float(eps), grad_slab).view(x.shape)
    xs = x.contiguous()
    part_stride = xs.stride(0) if n_partial > 1 else 0
    shape = xs.shape[1:] if n_partial > 1 else xs.shape
    D = shape[-1]
    rows = int(math.prod(shape[:-1]))
    odt = out_dtype or (g.dtype if g is not None else xs.dtype)
    rows_inner, g_outer, g_inner = 1, D, 0
    gs = None
    if g is not None:
        gs = g if g.dtype == odt else g.to(odt)
        if (gs.dim() == 3 and gs.stride(-1) == 1 and tuple(gs.shape) == tuple(shape[-3:]) and rows == gs.shape[0] * gs.shape[1]
                and gs.stride(0) % 4 == 0 and gs.stride(1) % 4 == 0 and gs.data_ptr() % (4 * gs.element_size()) == 0):
            rows_inner, g_outer, g_inner = gs.shape[1], gs.stride(0), gs.stride(1)
        else:
            gs = gs.contiguous()
    ws = None if weight is None else weight.to(odt).contiguous()
    y = out if out is not None else torch.empty(shape, dtype=odt, device=xs.device)
    _check(be.lib.lina_rmsnorm_gate_fwd(_ptr(xs), _ptr(gs), _ptr(ws), _ptr(y), rows, rows_inner, D,
                                        D * rows_inner, D if rows_inner > 1 else 0, g_outer, g_inner,
                                        D * rows_inner, D if rows_inner > 1 else 0,
                                        n_partial, part_stride, float(eps), _dt(xs), _dt(y), be.stream(xs)))
    return y


def rmsnorm(x, weight=None, eps: float = 1e-5):
    return rmsnorm_swish_gate(x, None, weight, eps)


# --------------------------------------------------------------------------- training glue (K10 / K11)
class _LayerNormFunction(torch.autograd.Function):
    """K10: y = LayerNorm(x [+ r]) (and x + r when a branch is added), contiguous rows [N, D]; see lina_gla.h."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps, y_dtype):
        be = _backend._BACKEND
        N, D = x.shape
        y = torch.empty(N, D, dtype=y_dtype, device=x.device)
        xsum = torch.empty_like(x) if r is not None else None
        mean = torch.empty(N, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        rdt = _dt(r) if r is not None else _dt(y)
        _check(be.lib.lina_layernorm_fwd(_ptr(x), _ptr(r), _ptr(gamma), _ptr(beta), _ptr(xsum), _ptr(y), _ptr(mean),
                                         _ptr(rstd), N, D, float(eps), _dt(x), rdt, _dt(y), be.stream(x)))
        ctx.save_for_backward(xsum if r is not None else x, mean, rstd, gamma)
        ctx.has_r, ctx.r_dtype, ctx.rdt = r is not None, (r.dtype if r is not None else None), rdt
        ctx.mark_non_differentiable(mean, rstd)
        if r is not None:
            return y, xsum
        return y

    @staticmethod
    def backward(ctx, dy, dxsum=None):
        xs, mean, rstd, gamma = ctx.saved_tensors
        be = _backend._BACKEND
        N, D = xs.shape
        dy = dy.contiguous()
        dpass = None if dxsum is None else dxsum.to(xs.dtype).contiguous()
        dx = torch.empty_like(xs)
        dr = torch.empty(N, D, dtype=ctx.r_dtype, device=xs.device) if ctx.has_r and ctx.r_dtype != xs.dtype else None
        npart = int(be.lib.lina_layernorm_bwd_partials(N))
        part = torch.empty(2, npart, D, dtype=torch.float32, device=xs.device)
        _check(be.lib.lina_layernorm_bwd(_ptr(dy), _ptr(xs), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dpass), _ptr(dx),
                                         _ptr(dr), _ptr(part[0]), _ptr(part[1]), N, D, _dt(xs), ctx.rdt, _dt(dy),
                                         be.stream(xs)))
        sums = _sum_partials2(part)
        d_r = None
        if ctx.has_r:
            d_r = dr if dr is not None else dx          # same values: the add passes the gradient through unchanged
        return dx, d_r, sums[0], sums[1], None, None


_LN_TRIPLES = {(torch.float32, torch.float32, torch.float32), (torch.float32, torch.bfloat16, torch.bfloat16),
               (torch.float32, torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16, torch.bfloat16)}


def layer_norm(x, weight, bias, eps: float = 1e-5, residual=None, out_dtype=None):
    """K10: ``LayerNorm(x [+ residual])`` over the last dimension (reference model/base_blocks.py:65-69).  Returns ``y``, or
    ``(y, x + residual)`` when a residual branch is given -- the add rides in the norm's pass, forward and backward.
    ``out_dtype``: dtype of ``y`` (default: the CUDA autocast dtype when autocast is on, else x.dtype).  Falls back to
    torch for shapes / dtype combinations the kernel is not built for."""
    be = _backend._BACKEND
    D = x.shape[-1]
    if out_dtype is None:
        out_dtype = torch.get_autocast_dtype("cuda") if (x.is_cuda and torch.is_autocast_enabled()) else x.dtype
    rdt = residual.dtype if residual is not None else out_dtype
    ok = (fused_ops_available(x) and D % 4 == 0 and D <= 2048 and weight is not None and bias is not None
          and (x.dtype, rdt, out_dtype) in _LN_TRIPLES and (residual is None or residual.shape == x.shape))
    if not ok:
        xs = x if residual is None else x + residual
        y = torch.nn.functional.layer_norm(xs, (D,), weight, bias, eps)
        return y if residual is None else (y, xs)
    be.require(x, residual, weight, bias)
    x2 = x.reshape(-1, D).contiguous()
    r2 = None if residual is None else residual.reshape(-1, D).contiguous()
    g32, b32 = weight.float().contiguous(), bias.float().contiguous()
    out = _LayerNormFunction.apply(x2, r2, g32, b32, float(eps), out_dtype)
    if residual is None:
        return out.view(x.shape)
    return out[0].view(x.shape), out[1].view(x.shape)


class _SwiGLUFunction(torch.autograd.Function):
    """K11 forward (lina_swiglu) + K11b backward on rows [N, 2 Hd] -> [N, Hd]."""

    @staticmethod
    def forward(ctx, u, hidden):
        be = _backend._BACKEND
        N = u.shape[0]
        y = torch.empty(N, hidden, dtype=u.dtype, device=u.device)
        _check(be.lib.lina_swiglu(_ptr(u), _ptr(y), N, hidden, u.stride(0), y.stride(0), _dt(u), be.stream(u)))
        ctx.save_for_backward(u)
        ctx.hidden = hidden
        return y

    @staticmethod
    def backward(ctx, dy):
        (u,) = ctx.saved_tensors
        be = _backend._BACKEND
        dy = dy.to(u.dtype).contiguous()
        du = torch.empty_like(u)
        _check(be.lib.lina_swiglu_bwd(_ptr(dy), _ptr(u), _ptr(du), u.shape[0], ctx.hidden, u.stride(0), dy.stride(0),
                                      du.stride(0), _dt(u), be.stream(u)))
        return du, None


def swiglu_gate(u):
    """``silu(a) * b`` with ``(a, b) = u.chunk(2, -1)`` (reference model/base_blocks.py:48-50), differentiable: one pass
    forward, one pass backward (autograd through chunk / silu / mul makes five, all unvectorised at L169's odd width)."""
    hidden = u.shape[-1] // 2
    if not fused_ops_available(u) or u.dtype not in (torch.float32, torch.bfloat16) or u.shape[-1] % 2:
        a, b = u.chunk(2, dim=-1)
        return torch.nn.functional.silu(a) * b
    _backend._BACKEND.require(u)
    u2 = u.reshape(-1, u.shape[-1]).contiguous()
    return _SwiGLUFunction.apply(u2, hidden).view(*u.shape[:-1], hidden)


_MM_OUT_DTYPE = None


def _mm_has_out_dtype(dev) -> bool:
    """torch.mm / torch.bmm with ``out_dtype=`` (bf16 operands, fp32 result) exist only in recent torch: probed once, on the
    device that asks, instead of failing with a TypeError in the middle of a backward pass."""
    global _MM_OUT_DTYPE
    if _MM_OUT_DTYPE is None:
        try:
            a = torch.zeros(8, 8, dtype=torch.bfloat16, device=dev)
            _MM_OUT_DTYPE = torch.mm(a, a, out_dtype=torch.float32).dtype == torch.float32
        except (TypeError, RuntimeError):
            _MM_OUT_DTYPE = False
    return _MM_OUT_DTYPE


def linear_weight_grad(dy2, x2, split=None):
    """dW [out, in] (fp32) = dy2^T x2 for dy2 [rows, out], x2 [rows, in] of one GEMM dtype: token-split batched GEMM with
    fp32 partial products (see above); ``split`` None = by shape."""
    rows, n_out = dy2.shape
    n_in = x2.shape[1]
    S = _linear_split(rows, n_out, n_in) if split is None else split
    f32 = {}
    if dy2.dtype != torch.float32:
        if dy2.is_cuda and _mm_has_out_dtype(dy2.device):
            f32 = {"out_dtype": torch.float32}
        else:
            dy2, x2 = dy2.float(), x2.float()      # (CPU / older torch: no fp32-output bf16 GEMM; same sum, fp32 operands)
    if S == 1:
        return torch.mm(dy2.t(), x2, **f32)
    if n_in % 8:                                   # rows of x2 not 16-byte aligned: the transposed problem is the faster one
        return torch.bmm(x2.view(S, rows // S, n_in).transpose(1, 2), dy2.view(S, rows // S, n_out), **f32).sum(0).t()
    return torch.bmm(dy2.view(S, rows // S, n_out).transpose(1, 2), x2.view(S, rows // S, n_in), **f32).sum(0)


class _LinearFunction(torch.autograd.Function):
    """y = x W^T + b in the GEMM dtype (the autocast dtype when autocast is on, like F.linear under autocast); backward:
    dX on the library GEMM, dW by ``linear_weight_grad``, db as the fp32-accumulated column sum."""

    @staticmethod
    def forward(ctx, x, w, b):
        cd = x.dtype
        if x.is_cuda and torch.is_autocast_enabled("cuda"):
            cd = torch.get_autocast_dtype("cuda")
        xc, wc = x.to(cd), w.to(cd)
        with torch.autocast(x.device.type, enabled=False):
            y = F.linear(xc, wc, None if b is None else b.to(cd))
        ctx.save_for_backward(xc, wc)
        ctx.meta = (x.dtype, w.dtype, None if b is None else b.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        xdt, wdt, bdt = ctx.meta
        n_out, n_in = wc.shape
        dy2 = dy.to(xc.dtype).reshape(-1, n_out)
        x2 = xc.reshape(-1, n_in)
        with torch.autocast(xc.device.type, enabled=False):
            dx = dw = db = None
            if ctx.needs_input_grad[0]:
                dx = torch.mm(dy2, wc).view(xc.shape).to(xdt)
            if ctx.needs_input_grad[1]:
                dw = linear_weight_grad(dy2.contiguous(), x2.contiguous()).to(wdt)
            if bdt is not None and ctx.needs_input_grad[2]:
                db = column_sum(dy2.contiguous()).to(bdt)
        return dx, dw, db


class _StackedLinearFunction(torch.autograd.Function):
    """``F.linear(x, cat(parts, 0))`` for row blocks ``parts`` that are separate fp32 parameters (the q | k | v | g | low-rank
    projections of a mixer input, reference model/gla.py:158-160,216): the stacked GEMM-dtype operand is ONE pass over the
    master weights (K16 ``lina_stack_rows``, with ``pad`` zero rows after the blocks) instead of ``torch.cat`` (an fp32 copy of
    all of them) + the autocast cast; backward: dX on the library GEMM, dW as in ``_LinearFunction`` (fp32, token-split), and
    every block's gradient is its row range of dW -- a contiguous view, no split pass."""

    @staticmethod
    def forward(ctx, x, pad, *parts):
        be = _backend._BACKEND
        cd = x.dtype
        if x.is_cuda and torch.is_autocast_enabled("cuda"):
            cd = torch.get_autocast_dtype("cuda")
        rows = [int(p.shape[0]) for p in parts]
        n_in = parts[0].shape[1]
        wc = torch.empty(sum(rows) + pad, n_in, dtype=cd, device=x.device)
        import ctypes as C
        srcs = (C.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
        nrow = (C.c_int * len(parts))(*rows)
        _check(be.lib.lina_stack_rows(srcs, nrow, len(parts), n_in, wc.shape[0], _ptr(wc), _dt(wc), be.stream(wc)))
        xc = x.to(cd)
        n_out = wc.shape[0]
        main = _stacked_main(n_out, rows) if POLICY.split_stacked_gemm else n_out
        with torch.autocast(x.device.type, enabled=False):
            if main < n_out:
                # a 256-aligned main product and the narrow tail, both written into ONE [.., n_out] buffer (out= views)
                x2 = xc.reshape(-1, n_in)
                y2 = torch.empty(x2.shape[0], n_out, dtype=cd, device=x.device)
                torch.mm(x2, wc[:main].t(), out=y2[:, :main])
                torch.mm(x2, wc[main:].t(), out=y2[:, main:])
                y = y2.view(*xc.shape[:-1], n_out)
            else:
                y = F.linear(xc, wc)
        ctx.save_for_backward(xc, wc)
        ctx.rows, ctx.xdt, ctx.main = rows, x.dtype, main
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        n_out, n_in = wc.shape
        main = ctx.main
        dy2 = dy.to(xc.dtype).reshape(-1, n_out)
        with torch.autocast(xc.device.type, enabled=False):
            dx = torch.mm(dy2, wc).view(xc.shape).to(ctx.xdt) if ctx.needs_input_grad[0] else None
            grads = [None] * len(ctx.rows)
            if any(ctx.needs_input_grad[2:]):
                x2 = xc.reshape(-1, n_in).contiguous()
                dy2 = dy2.contiguous()
                if main < n_out:
                    # the main rows token-split (the library's one-GEMM form of a [4096, 1024] result from 32768 tokens runs
                    # 334 us, split in four + a sum 259), the narrow tail on its own; every block lies inside one of the two
                    M = dy2.shape[0]
                    S = _linear_split(M, main, n_in)
                    if S == 1 and M % 4 == 0 and M // 4 >= 2048:
                        S = 4
                    pieces = [(0, main, linear_weight_grad(dy2[:, :main], x2, split=S)),
                              (main, n_out, linear_weight_grad(dy2[:, main:].contiguous(), x2))]
                else:
                    pieces = [(0, n_out, linear_weight_grad(dy2, x2))]                  # fp32 [n_out, n_in]
                r0 = 0
                for i, r in enumerate(ctx.rows):
                    if ctx.needs_input_grad[2 + i]:
                        lo, _, t = next(pc for pc in pieces if pc[0] <= r0 and r0 + r <= pc[1])
                        grads[i] = t[r0 - lo:r0 - lo + r]
                    r0 += r
        return (dx, None, *grads)


def _stacked_main(n_out: int, rows) -> int:
    """Rows of a stacked weight that go into the 256-aligned main product (the rest: a narrow tail product), or ``n_out`` when the
    split does not apply: the tail must be short (<= 128 rows), the main part large, and no block may straddle the cut."""
    main = n_out // 256 * 256
    if main == n_out or main < 1024 or n_out - main > 128:
        return n_out
    r0 = 0
    for r in rows:
        if r0 < main < r0 + r:
            return n_out
        r0 += r
    return main


def stacked_linear(x, parts, pad: int = 0):
    """``F.linear(x, torch.cat(parts + [zeros(pad, n_in)], 0))`` -- see ``_StackedLinearFunction``.  Falls back to that very
    expression (through ``linear``) when the blocks are not contiguous fp32 parameters on the fused-op device."""
    ok = (POLICY.one_pass_operands and torch.is_grad_enabled() and fused_ops_available(x) and x.dim() >= 2 and 1 <= len(parts) <= 8
          and all(p.dtype == torch.float32 and p.is_contiguous() and p.dim() == 2 and p.shape[1] == parts[0].shape[1]
                  and p.data_ptr() % 16 == 0 for p in parts) and parts[0].shape[1] % 4 == 0)
    cd = torch.get_autocast_dtype("cuda") if (x.is_cuda and torch.is_autocast_enabled("cuda")) else x.dtype
    if not ok or cd not in (torch.float32, torch.bfloat16):
        w = list(parts)
        if pad:
            w.append(parts[0].new_zeros(pad, parts[0].shape[1]))
        return linear(x, torch.cat(w, dim=0))
    return _StackedLinearFunction.apply(x, pad, *parts)


def linear(x, weight, bias=None):
    """``F.linear(x, weight, bias)`` for the projections of the train path: same forward GEMM (autocast semantics
    included), weight gradient posed as a token-split batched GEMM in fp32.  Without gradients: F.linear itself."""
    if not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)) or x.dim() < 2:
        return F.linear(x, weight, bias)
    return _LinearFunction.apply(x, weight, bias)


class _SwiGLUMLPFunction(torch.autograd.Function):
    """The channel mixer ``p_out(silu(a) * b)``, ``(a, b) = p_in(x).chunk(2)`` (reference model/base_blocks.py:42-50) as ONE
    node for the train path.  L169's hidden size is 1365 = 1024 * 4 // 3: rows of 1365 / 2730 elements are not 16-byte
    aligned and the GEMM library runs every one of the six GEMMs 15-35 % slower on them (profiles/r03_pad_gemm.txt).  Here
    the operands live in a PADDED layout: hidden Hp = the next multiple of 128 above H, the halves of the up-projection at
    rows [0, H) and [Hp, Hp + H) of a zero-padded weight, the pad columns of the gate exactly 0 -- except column H, which
    the bias pack makes exactly 1 (a = 32, b = 1/32: silu(32) * (1/32) == 1 in fp32 and in bf16), so that the
    down-projection's bias is column H of its padded weight and its gradient column H of the padded weight gradient: no
    bias epilogue, no column sum.  The up-projection's bias gradient is summed inside the gate's backward (K11c)."""

    @staticmethod
    def forward(ctx, x, w_in, b_in, w_out, b_out):
        be = _backend._BACKEND
        cd = x.dtype
        if x.is_cuda and torch.is_autocast_enabled("cuda"):
            cd = torch.get_autocast_dtype("cuda")
        H, d_out, d_in = w_out.shape[1], w_out.shape[0], w_in.shape[1]
        Hp = (H + _MLP_PAD) // _MLP_PAD * _MLP_PAD                    # > H: room for the bias column
        dev = x.device
        x2 = x.reshape(-1, d_in).to(cd).contiguous()
        with torch.autocast(dev.type, enabled=False):
            Wi, bi, Wo, Wo_wide = _mlp_padded_weights(w_in, b_in, w_out, b_out, cd, H, Hp)
            u = torch.addmm(bi.view(-1), x2, Wi.view(2 * Hp, d_in).t())
            h = torch.empty(x2.shape[0], Hp, dtype=cd, device=dev)
            _check(be.lib.lina_swiglu(_ptr(u), _ptr(h), x2.shape[0], Hp, u.stride(0), h.stride(0), _dt(u), be.stream(u)))
            y = torch.mm(h, Wo.t())
        ctx.save_for_backward(x2, u, h, Wi, Wo_wide)
        ctx.meta = (x.shape, x.dtype, H, Hp, w_in.dtype, None if b_in is None else b_in.dtype, w_out.dtype,
                    None if b_out is None else b_out.dtype)
        return y.view(*x.shape[:-1], d_out)

    @staticmethod
    def backward(ctx, dy):
        x2, u, h, Wi, Wo_wide = ctx.saved_tensors
        x_shape, xdt, H, Hp, widt, bidt, wodt, bodt = ctx.meta
        be = _backend._BACKEND
        d_out, d_in = Wo_wide.shape[0], x2.shape[1]
        M = x2.shape[0]
        with torch.autocast(x2.device.type, enabled=False):
            dy2 = dy.reshape(M, d_out).to(x2.dtype).contiguous()
            # dX of the down-projection on the WIDE operand ([d_out, Hq], zero columns past H + 1): the library runs
            # [M, 1024] x [1024, 1536] in 90-100 us and [M, 1024] x [1024, 1408] in 112-119 (profiles/r06_inproj_gemm_split.txt);
            # K11c reads the first Hp columns of dh through its row stride
            dh = torch.mm(dy2, Wo_wide)
            # weight gradients as token-split batched GEMMs whose PARTIAL products are summed straight into the parameters'
            # own (unpadded) layouts: the sum over the split reads only the rows / columns that exist in w_in / w_out, so no
            # padded [2 Hp, d] / [d, Hp] gradient is formed and re-packed afterwards (a reshape clone + two AccumulateGrad clones
            # per block, profiles/r06_train_step_ops.txt)
            S = _linear_split(M, 2 * Hp, d_in)
            compact = (S > 1 and S == _linear_split(M, d_out, Hp) and dy2.is_cuda and dy2.dtype != torch.float32
                       and _mm_has_out_dtype(dy2.device) and d_in % 8 == 0 and Hp % 8 == 0)
            Po = None
            if compact:
                Po = torch.bmm(dy2.view(S, M // S, d_out).transpose(1, 2), h.view(S, M // S, Hp), out_dtype=torch.float32)
            else:
                dWo = linear_weight_grad(dy2, h)                                     # [d_out, Hp] fp32; column H = db_out
            du = torch.empty_like(u)
            part = torch.empty(int(be.lib.lina_swiglu_bwd_partials(M)), 2 * Hp, dtype=torch.float32, device=u.device)
            _check(be.lib.lina_swiglu_bwd_colsum(_ptr(dh), _ptr(u), _ptr(du), _ptr(part), M, Hp, u.stride(0), dh.stride(0),
                                                 du.stride(0), _dt(u), be.stream(u)))
            dx = torch.mm(du, Wi.view(2 * Hp, d_in)).view(x_shape).to(xdt) if ctx.needs_input_grad[0] else None
            db_in = None if bidt is None else _sum_partials(part).view(2, Hp)[:, :H].reshape(2 * H).to(bidt)
            if compact:
                Pi = torch.bmm(du.view(S, M // S, 2 * Hp).transpose(1, 2), x2.view(S, M // S, d_in), out_dtype=torch.float32)
                dw_in = Pi.view(S, 2, Hp, d_in)[:, :, :H].sum(0).view(2 * H, d_in).to(widt)
                dw_out = Po[:, :, :H].sum(0).to(wodt)
                db_out = None if bodt is None else Po[:, :, H].sum(0).to(bodt)
            else:
                dWi = linear_weight_grad(du, x2).view(2, Hp, d_in)
                dw_in = dWi[:, :H].reshape(2 * H, d_in).to(widt)
                dw_out = dWo[:, :H].to(wodt)
                db_out = None if bodt is None else dWo[:, H].to(bodt)
        return dx, dw_in, db_in, dw_out, db_out


_MLP_PACK = WeakIdKeyDictionary()            # up-projection weight (the parameter OBJECT) -> (key, Wi, bi, Wo)


def clear_mlp_pack() -> None:
    """Forget the cached padded SwiGLU weights.  The cache key is every parameter's (storage, version): in-place ops on the
    Parameter and ``load_state_dict`` bump the version, writes through ``param.data`` (EMA swaps, weight clipping, some
    third-party optimizers) do NOT -- after such a write call this (``TrainStep.step`` does after every optimizer step,
    ``ops.clear_workspaces`` does too)."""
    _MLP_PACK.clear()


def _mlp_padded_weights(w_in, b_in, w_out, b_out, cd, H, Hp):
    """The zero-padded operands of ``_SwiGLUMLPFunction`` (the cast of the master weights that happens every step anyway, into
    the padded layout).  Kept per parameter VERSION: the weights change only at the optimizer step, so a second forward on the
    same weights -- the recompute of a checkpointed block, gradient accumulation -- reuses the pack instead of rebuilding three
    weight-sized tensors."""
    ver = lambda t: None if t is None else (t.data_ptr(), t._version)
    key = (cd, H, Hp, ver(w_in), ver(b_in), ver(w_out), ver(b_out))
    hit = _MLP_PACK.get(w_in)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2], hit[3], hit[4]
    dev, d_in, d_out = w_in.device, w_in.shape[1], w_out.shape[0]
    Wi = torch.empty(2, Hp, d_in, dtype=cd, device=dev)
    bi = torch.empty(2, Hp, dtype=cd, device=dev)
    # the down-projection's operand lives in rows of Hq >= Hp elements (zero past column H): Hq = the next multiple of 512 when that
    # costs <= 1/8 more columns (1408 -> 1536: the width the GEMM library prefers for the dX product), else Hp
    Hq = -(-Hp // 512) * 512
    if Hp < 1024 or Hq * 8 > Hp * 9 or not POLICY.wide_down_dx:
        Hq = Hp
    Wo_wide = torch.empty(d_out, Hq, dtype=cd, device=dev)
    Wo = Wo_wide[:, :Hp]
    f32c = lambda t: t is None or (t.dtype == torch.float32 and t.is_contiguous())
    if (POLICY.one_pass_operands and f32c(w_in) and f32c(b_in) and f32c(w_out) and f32c(b_out) and d_in % 4 == 0
            and w_in.data_ptr() % 16 == 0):             # (K15 reads w_in's rows with 16-byte accesses)
        # K15: the three operands in ONE pass over the fp32 master weights (torch built them with three fills and five strided
        # copies: nine launches per block and step)
        be = _backend._BACKEND
        _check(be.lib.lina_mlp_pack(_ptr(w_in.detach()), _ptr(None if b_in is None else b_in.detach()), _ptr(w_out.detach()),
                                    _ptr(None if b_out is None else b_out.detach()), _ptr(Wi), _ptr(bi), _ptr(Wo_wide), H, Hp, Hq,
                                    d_in, d_out, _dt(Wi), be.stream(Wi)))
    else:                                       # master weights that are not contiguous fp32: the same values with torch ops
        Wi[:, H:].zero_()
        Wi[:, :H].copy_(w_in.detach().view(2, H, d_in))
        bi.zero_()
        if b_in is not None:
            bi[:, :H].copy_(b_in.detach().view(2, H))
        Wo_wide[:, H:].zero_()
        Wo[:, :H].copy_(w_out.detach())
        if b_out is not None:
            bi[:, H] = _mlp_one(cd, dev)
            Wo[:, H].copy_(b_out.detach())
    _MLP_PACK[w_in] = (key, Wi, bi, Wo, Wo_wide)
    return Wi, bi, Wo, Wo_wide


_MLP_ONE = {}


def _mlp_one(dtype, device):
    """(32, 1/32): the bias pair that makes the gate's column H exactly 1 (cached per dtype / device)."""
    key = (dtype, device)
    if key not in _MLP_ONE:
        _MLP_ONE[key] = torch.tensor([32.0, 1.0 / 32.0], dtype=dtype, device=device)
    return _MLP_ONE[key]


def swiglu_mlp(x, w_in, b_in, w_out, b_out):
    """``F.linear(silu(a) * b, w_out, b_out)`` with ``(a, b) = F.linear(x, w_in, b_in).chunk(2, -1)`` -- the channel mixer of
    a block (reference model/base_blocks.py:42-50).  With gradients on the fused-op devices: one autograd node on padded
    operands (see ``_SwiGLUMLPFunction``); otherwise the three ops."""
    ok = (torch.is_grad_enabled() and (x.requires_grad or w_in.requires_grad or w_out.requires_grad)
          and fused_ops_available(x) and w_in.shape[0] == 2 * w_out.shape[1])
    cd = torch.get_autocast_dtype("cuda") if (x.is_cuda and torch.is_autocast_enabled("cuda")) else x.dtype
    if not ok or cd not in (torch.float32, torch.bfloat16) or x.numel() == 0:
        return linear(swiglu_gate(linear(x, w_in, b_in)), w_out, b_out)
    _backend._BACKEND.require(x, w_in, w_out)
    return _SwiGLUMLPFunction.apply(x, w_in, b_in, w_out, b_out)


class _GateLogSigmoidFunction(torch.autograd.Function):
    """K12: logsigmoid(x) / normalizer (optionally clamped) and its gradient, one pass each."""

    @staticmethod
    def forward(ctx, x, normalizer, clamp_min):
        be = _backend._BACKEND
        y = torch.empty_like(x)
        cm = float("nan") if clamp_min is None else float(clamp_min)
        _check(be.lib.lina_gate_logsigmoid(_ptr(x), None, _ptr(y), x.numel(), float(normalizer), cm, _dt(x), be.stream(x)))
        ctx.save_for_backward(x)
        ctx.args = (float(normalizer), cm)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        be = _backend._BACKEND
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        _check(be.lib.lina_gate_logsigmoid(_ptr(x), _ptr(dy), _ptr(dx), x.numel(), ctx.args[0], ctx.args[1], _dt(x),
                                           be.stream(x)))
        return dx, None, None


def gate_logsigmoid(x, normalizer: float = 16.0, clamp_min: Optional[float] = None):
    """``logsigmoid(x) / normalizer`` (clamped from below when ``clamp_min`` is given) -- the mixer's gate (reference
    model/gla.py:174-180), differentiable, one pass each way; torch fallback off-device / for other dtypes."""
    if (not fused_ops_available(x) or x.dtype not in (torch.float32, torch.bfloat16) or x.numel() % 4 or x.numel() == 0):
        g = torch.nn.functional.logsigmoid(x) / normalizer
        return g if clamp_min is None else torch.clamp_min(g, clamp_min)
    _backend._BACKEND.require(x)
    return _GateLogSigmoidFunction.apply(x.contiguous(), float(normalizer), clamp_min).view(x.shape)


class _GateLowRankFunction(torch.autograd.Function):
    """K12b: logsigmoid(lr W^T + b) / normalizer in one pass; backward d(lr), dW, db without the [R, C] pre-activation."""

    @staticmethod
    def forward(ctx, lr, w, b, normalizer, clamp_min):
        be = _backend._BACKEND
        C_, L = w.shape
        rows = lr.numel() // L
        lr2 = lr.reshape(rows, L)                              # a view for column slices of a wider row (the slab)
        if lr2.stride(1) != 1:
            lr2 = lr2.contiguous()
        wf = w.detach().float().contiguous()
        bf = None if b is None else b.detach().float().contiguous()
        y = torch.empty(*lr.shape[:-1], C_, dtype=lr.dtype, device=lr.device)
        cm = float("nan") if clamp_min is None else float(clamp_min)
        _check(be.lib.lina_gate_lowrank(_ptr(lr2), lr2.stride(0), _ptr(wf), _ptr(bf), None, _ptr(y), None, rows, C_, L,
                                        float(normalizer), cm, _dt(lr2), be.stream(lr)))
        ctx.save_for_backward(lr2, wf, bf)
        ctx.args = (float(normalizer), cm, lr.shape, w.dtype, None if b is None else b.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        lr2, wf, bf = ctx.saved_tensors
        normalizer, cm, lr_shape, wdt, bdt = ctx.args
        be = _backend._BACKEND
        C_, L = wf.shape
        rows = lr2.shape[0]
        dy2 = dy.to(lr2.dtype).reshape(rows, C_).contiguous()
        dpre = torch.empty_like(dy2)
        part = torch.empty(int(be.lib.lina_gate_lowrank_partials(rows)), C_, L + 1, dtype=torch.float32, device=dy2.device)
        _check(be.lib.lina_gate_lowrank(_ptr(lr2), lr2.stride(0), _ptr(wf), _ptr(bf), _ptr(dy2), _ptr(dpre), _ptr(part),
                                        rows, C_, L, normalizer, cm, _dt(lr2), be.stream(dy2)))
        red = _sum_partials(part)
        dlr = None
        if ctx.needs_input_grad[0]:
            with torch.autocast(dy2.device.type, enabled=False):
                dlr = torch.mm(dpre, wf.to(dpre.dtype)).view(lr_shape)
        dw = red[:, :L].to(wdt) if ctx.needs_input_grad[1] else None
        db = red[:, L].to(bdt) if (bdt is not None and ctx.needs_input_grad[2]) else None
        return dlr, dw, db, None, None


def gate_lowrank(lr, weight, bias=None, normalizer: float = 16.0, clamp_min: Optional[float] = None):
    """``logsigmoid(F.linear(lr, weight, bias)) / normalizer`` (clamped from below when ``clamp_min`` is given): the second
    factor of the mixer's low-rank gate projection fused with the gate (reference model/gla.py:107-109,174-180), K12b.
    ``weight`` [C, L <= 16].  Falls back to the unfused ops where the kernel does not apply."""
    C_, L = weight.shape
    gemm_dt = lr.dtype
    if lr.is_cuda and torch.is_autocast_enabled("cuda"):
        gemm_dt = torch.get_autocast_dtype("cuda")
    if (not fused_ops_available(lr) or gemm_dt not in (torch.float32, torch.bfloat16) or L > 16 or C_ % 4
            or lr.numel() == 0 or lr.numel() // L > 65535 * 128):
        return gate_logsigmoid(linear(lr, weight, bias), normalizer, clamp_min)
    _backend._BACKEND.require(lr, weight, bias)
    return _GateLowRankFunction.apply(lr.to(gemm_dt), weight, bias, float(normalizer), clamp_min)


class _CrossEntropyFunction(torch.autograd.Function):
    """K14: mean cross-entropy over the rows whose target is not ``ignore_index``; rows [N, V] of the logits' own dtype."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        be = _backend._BACKEND
        N, V = logits.shape
        lse = torch.empty(N, dtype=torch.float32, device=logits.device)
        rows = torch.empty(N, dtype=torch.float32, device=logits.device)
        _check(be.lib.lina_cross_entropy(_ptr(logits), _ptr(target), _ptr(lse), _ptr(rows), None, None, N, V, logits.stride(0),
                                         0, int(ignore_index), _dt(logits), be.stream(logits)))
        valid = (target != ignore_index).to(torch.float32)
        count, total = _sum_vector(valid), _sum_vector(rows)
        ctx.save_for_backward(logits, target, lse, count)
        ctx.ignore_index = int(ignore_index)
        return total / count

    @staticmethod
    def backward(ctx, dloss):
        logits, target, lse, count = ctx.saved_tensors
        be = _backend._BACKEND
        N, V = logits.shape
        scale = (dloss.to(torch.float32) / count).reshape(1).contiguous()
        dlogits = torch.empty(N, V, dtype=logits.dtype, device=logits.device)
        _check(be.lib.lina_cross_entropy(_ptr(logits), _ptr(target), _ptr(lse), None, _ptr(scale), _ptr(dlogits), N, V,
                                         logits.stride(0), dlogits.stride(0), ctx.ignore_index, _dt(logits), be.stream(logits)))
        return dlogits, None, None


def cross_entropy(logits, target, ignore_index: int = -100):
    """``F.cross_entropy(logits, target, ignore_index=ignore_index)`` (mean over the rows that count; reference
    modeling_lina.py:106) for logits [N, V], target int64 [N]: K14, one pass over the logits each way in fp32 arithmetic
    from their own dtype.  Falls back to torch where the kernel does not apply."""
    if (not fused_ops_available(logits) or logits.dim() != 2 or logits.dtype not in (torch.float32, torch.bfloat16)
            or target.dtype != torch.int64 or logits.shape[0] == 0 or not 4 <= logits.shape[1] <= 8445):
        return F.cross_entropy(logits, target, ignore_index=ignore_index)
    _backend._BACKEND.require(logits, target)
    lg = logits if (logits.stride(1) == 1 and logits.data_ptr() % 16 == 0) else logits.contiguous()
    return _CrossEntropyFunction.apply(lg, target.contiguous(), ignore_index)


class _EmbedSumFunction(torch.autograd.Function):
    """K6 gather forward; the backward is a scatter-add of the output gradient into the table rows
    (torch index_add_ in fp32 on the device -- plumbing, not a hand-written kernel)."""

    @staticmethod
    def forward(ctx, table, flat, padding_idx=None):
        ctx.save_for_backward(flat)
        ctx.tshape, ctx.tdtype, ctx.padding_idx = table.shape, table.dtype, padding_idx
        return _embed_sum_launch(table, flat)

    @staticmethod
    def backward(ctx, dout):
        (flat,) = ctx.saved_tensors
        Q, n_emb, d = ctx.tshape
        dt = torch.zeros(Q, n_emb, d, dtype=torch.float32, device=dout.device)
        src = dout.float()
        for qi in range(Q):
            dt[qi].index_add_(0, flat[qi], src)
        if ctx.padding_idx is not None:                   # F.embedding(padding_idx=...): that row gets no gradient
            dt[:, ctx.padding_idx].zero_()
        return dt.to(ctx.tdtype), None, None


def embed_sum(table, idx, out=None, padding_idx=None):
    """table [Q,n_emb,d], idx int64 [Q,B,n] -> sum_q table[q, idx[q]] : [B,n,d]
    (MultiEmbedding + reduce over quantizers; reference modeling_lina.py:131,178-179).
    ``out``: optional contiguous [B*n, d] destination (no-grad path).  ``padding_idx``: row that receives no
    gradient (the forward value is gathered like any other row, as in the reference)."""
    be = _backend._BACKEND
    be.require(table, idx)
    Q, n_emb, d = table.shape
    if idx.shape[0] != Q or idx.dtype != torch.int64:
        raise ValueError("idx must be int64 [Q, ...]")
    flat = idx.reshape(Q, -1).contiguous()
    if _needs_grad(table):
        return _EmbedSumFunction.apply(table, flat, padding_idx).view(*idx.shape[1:], d)
    return _embed_sum_launch(table, flat, out).view(*idx.shape[1:], d)
