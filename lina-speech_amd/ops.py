"""Operator-level drop-in surface: the names the reference imports from ``fla``
(/root/reference/model/gla.py:19-23), served by the hand-written HIP kernels through
the C ABI of include/lina_gla.h.

Every function takes/returns torch tensors exactly like the fla operator it replaces
(head-first ``[B,H,T,D]`` views, ``(o, final_state)`` returns) and enqueues on the
CURRENT torch HIP stream without synchronising, so callers can graph-capture.
There is no CPU path: tensors must live on a ROCm device (``Backend.require``).
GLA ops are differentiable (K2b); the other custom ops raise on tensors that require grad
unless stated otherwise.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import BHT, LINA_BF16, LINA_F32


class HipBackend:
    """Default provider of the C ABI: the in-tree HIP library, tensors on a ROCm device."""
    name = "hip"

    def __init__(self):
        self._lib = None

    @property
    def lib(self):
        if self._lib is None:
            self._lib = _lib.load()
        return self._lib

    def require(self, *tensors):
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise RuntimeError("lina_speech_amd ops run on a ROCm GPU only (got a %s tensor); "
                                   "there is no CPU fallback" % t.device)

    def stream(self, ref: torch.Tensor):
        return C.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream)


_BACKEND = HipBackend()


def set_backend(backend) -> None:
    """Install another provider of the same C ABI (object with .lib/.require/.stream)."""
    global _BACKEND
    _BACKEND = backend


def get_backend():
    return _BACKEND


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return LINA_F32
    if t.dtype == torch.bfloat16:
        return LINA_BF16
    raise TypeError(f"unsupported dtype {t.dtype} (float32 or bfloat16)")


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _no_grad(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError("lina_speech_amd: this op has no backward kernel (inference-only: SwiGLU decode "
                                  "epilogue, vocoder K8/K9); call it under torch.no_grad()/inference_mode()")


def _inner_contig(t: torch.Tensor) -> torch.Tensor:
    return t if t.stride(-1) == 1 else t.contiguous()


def _bht(t: torch.Tensor) -> BHT:
    return BHT(t.stride(0), t.stride(1), t.stride(2))


def _check(rc: int):
    if rc != 0:
        raise _lib.LinaError(f"lina C-ABI error {rc}: {_BACKEND.lib.lina_last_error().decode()}")


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
    _BACKEND.require(q, k, v, gk, initial_state)
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


_WORKSPACES = {}


def _workspace(tag: str, nbytes: int, device) -> torch.Tensor:
    """Scratch that is fully written before it is read inside ONE launch sequence on the current stream (segment
    states of the segment-parallel K2): kept per (tag, device, stream) and grown on demand instead of a torch.empty
    per layer per step.  Stream-ordered reuse is safe because consecutive users on one stream serialise."""
    if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        # inside a graph capture the allocation belongs to the graph's private pool: never hand it to eager launches
        return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    key = (tag, device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _WORKSPACES[key] = ws
    return ws


def clear_workspaces() -> None:
    """Release the cached kernel scratch (segment / boundary states of the segment-parallel K2 / K2b): at small B*H with
    many segments it is the size of several activations and would otherwise stay pinned for the life of the process."""
    _WORKSPACES.clear()


def _value_blocks(q, v, gk) -> int:
    """Dv = m * Dk with the L169 key width (``expand_v = 2``: 256 x 512 heads): the recurrence is independent per value
    column, so the call runs as m calls of the full-head 256 x 256 kernel on column blocks of v / o / the states (same q, k,
    g); the backward adds the blocks' dq, dk, dg.  Returns m (1 = no split)."""
    Dk, Dv = q.shape[-1], v.shape[-1]
    if q.dtype == torch.bfloat16 and gk.dtype == torch.bfloat16 and Dk == 256 and Dv > Dk and Dv % Dk == 0:
        return Dv // Dk
    return 1


def _gla_launch(entry: str, q, k, v, gk, scale, initial_state, output_final_state, inplace_state=False, nseg=None,
                keep_seg_states: Optional[list] = None, out: Optional[torch.Tensor] = None):
    """``keep_seg_states``: a list that receives (workspace, nseg) when the segment-parallel kernel ran -- the workspace is
    then a fresh tensor whose head holds the segment start states (the backward's seg_states), not the shared scratch."""
    B, H, T, Dk = q.shape
    Dv = v.shape[-1]
    be = _BACKEND
    m = _value_blocks(q, v, gk) if entry == "lina_gla_chunk_fwd" else 1
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


def chunk_segments(n_heads_total: int, T: int) -> int:
    """Segments for the segment-parallel K2 (lina_gla_chunk_fwd_seg): enough to put ~256 workgroups on the chip
    when B*H is small, at least 256 tokens per segment; 1 = the plain kernel."""
    if n_heads_total >= 128 or T < 1024:
        return 1
    return max(1, min(256 // n_heads_total, T // 256, 16))


def gla_chunk_bwd(q, k, v, gk, d_o, scale, initial_state=None, final_state=None, d_final_state=None,
                  need_dh0=False, nseg=None, path=None, seg_states=None):
    """K2b through the C ABI: returns (dq, dk, dv, dg, dh0).  bf16 tensors with Dk = Dv in {64,128,256} take the
    full-head sweeps (lina_gla_chunk_bwd_full, ``nseg`` sequence segments); everything else -- and ``path="sweeps"`` /
    LINA_K2B=sweeps -- the generic kernel (lina_gla_chunk_bwd).  ``seg_states``: the workspace the segment-parallel forward
    left for the same inputs and ``nseg`` (its head holds the segment start states; skips one pass)."""
    B, H, T, Dk = q.shape
    Dv = v.shape[-1]
    be = _BACKEND
    be.require(q, k, v, gk, d_o, initial_state, final_state, d_final_state)
    m = _value_blocks(q, v, gk) if (path or os.environ.get("LINA_K2B", "full")) == "full" else 1
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
    dq = _head_first_empty(B, H, T, Dk, q.dtype, q.device)
    dk = _head_first_empty(B, H, T, Dk, q.dtype, q.device)
    dv = _head_first_empty(B, H, T, Dv, q.dtype, q.device)
    dg = _head_first_empty(B, H, T, Dk, gk.dtype, q.device)
    dh0 = torch.empty(B, H, Dk, Dv, dtype=torch.float32, device=q.device) if need_dh0 else None
    path = path or os.environ.get("LINA_K2B", "full")
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


# --------------------------------------------------------------------------- short conv (K3 / K4)
def _short_conv_launch(x, w, bias, mask, cache, act):
    B, T, D = x.shape
    W = w.shape[1]
    be = _BACKEND
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
    be = _BACKEND
    out = torch.empty(part.shape[1:], dtype=out_dtype, device=part.device)
    _check(be.lib.lina_sum_partials(_ptr(part), _ptr(out), 1, P, N, _dt(out), be.stream(part)))
    return out


def _sum_partials2(part):
    """``part`` fp32 [O, P, N] -> [O, N] (K13 with an outer axis: the two parameter gradients of the LayerNorm)."""
    O, P, N = part.shape
    if N % 4 or N == 0:
        return part.sum(1)
    be = _BACKEND
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
    be = _BACKEND
    part = torch.empty(int(be.lib.lina_swiglu_bwd_partials(M)), N, dtype=torch.float32, device=x2.device)
    _check(be.lib.lina_colsum(_ptr(x2), _ptr(part), M, N, x2.stride(0), _dt(x2), be.stream(x2)))
    return _sum_partials(part, out_dtype)


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
        be = _BACKEND
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
    be = _BACKEND
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
        be = _BACKEND
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
        be = _BACKEND
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
    return v if v.stride(0) % 4 == 0 else None


def rmsnorm_swish_gate(x, g=None, weight=None, eps: float = 1e-5, n_partial: int = 1, out_dtype=None, out=None,
                       grad_slab=None):
    """FusedRMSNormSwishGate / RMSNorm forward over the last dim.
    ``n_partial`` > 1: ``x`` is [n_partial, ..., D] partial sums (fp32) that are added first.
    A gate ``g`` whose rows are head slices of wider rows ([..., H, D] view of a column slice) is read in place.
    ``grad_slab``: ``(GradSlab, index)`` when ``g`` is column slice ``index`` of a stacked projection (``split_slab``)."""
    be = _BACKEND
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
                and gs.stride(0) % 4 == 0 and gs.stride(1) % 4 == 0):
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
        be = _BACKEND
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
        be = _BACKEND
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


def fused_ops_available(x: torch.Tensor) -> bool:
    """True when the HIP (or emulated) ops can take ``x``: a ROCm tensor, or a CPU tensor under the test emulator."""
    return x.is_cuda if _BACKEND.name == "hip" else not x.is_cuda


def layer_norm(x, weight, bias, eps: float = 1e-5, residual=None, out_dtype=None):
    """K10: ``LayerNorm(x [+ residual])`` over the last dimension (reference model/base_blocks.py:65-69).  Returns ``y``, or
    ``(y, x + residual)`` when a residual branch is given -- the add rides in the norm's pass, forward and backward.
    ``out_dtype``: dtype of ``y`` (default: the CUDA autocast dtype when autocast is on, else x.dtype).  Falls back to
    torch for shapes / dtype combinations the kernel is not built for."""
    be = _BACKEND
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
        be = _BACKEND
        N = u.shape[0]
        y = torch.empty(N, hidden, dtype=u.dtype, device=u.device)
        _check(be.lib.lina_swiglu(_ptr(u), _ptr(y), N, hidden, u.stride(0), y.stride(0), _dt(u), be.stream(u)))
        ctx.save_for_backward(u)
        ctx.hidden = hidden
        return y

    @staticmethod
    def backward(ctx, dy):
        (u,) = ctx.saved_tensors
        be = _BACKEND
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
    _BACKEND.require(u)
    u2 = u.reshape(-1, u.shape[-1]).contiguous()
    return _SwiGLUFunction.apply(u2, hidden).view(*u.shape[:-1], hidden)


# --------------------------------------------------------------------------- projections of the train path
# The GEMMs stay on the vendor library (hipBLASLt through torch); what is ours is how the WEIGHT GRADIENT is posed to it.
# dW = dY^T X reduces over all B T tokens (32768 on config 5) into a small [out, in] tile grid: posed as one GEMM the
# library runs it at 300-580 TFLOP/s (1024x1024 / 1024x1365 / 2730x1024 outputs: 16-44 tiles for 256 CUs, profiles/
# r03_dw_gemm.txt); split over the token axis into a batched GEMM with fp32 partial products + one small sum it runs at
# 830-940 TFLOP/s, and dW comes out in fp32 (the master-weight dtype: no bf16 round trip, no cast kernel).
_LINEAR_SPLIT_MAX_OUT = 3 * 1024 * 1024        # [out, in] up to this many elements: split (above: enough tiles already)
_LINEAR_SPLIT_MIN_ROWS = 2048                  # tokens per split slice, at least


def _linear_split(rows, n_out, n_in):
    if n_out * n_in > _LINEAR_SPLIT_MAX_OUT:
        return 1
    for s in (8, 4, 2):
        if rows % s == 0 and rows // s >= _LINEAR_SPLIT_MIN_ROWS:
            return s
    return 1


def linear_weight_grad(dy2, x2, split=None):
    """dW [out, in] (fp32) = dy2^T x2 for dy2 [rows, out], x2 [rows, in] of one GEMM dtype: token-split batched GEMM with
    fp32 partial products (see above); ``split`` None = by shape."""
    rows, n_out = dy2.shape
    n_in = x2.shape[1]
    S = _linear_split(rows, n_out, n_in) if split is None else split
    f32 = {} if (dy2.dtype == torch.float32 or not dy2.is_cuda) else {"out_dtype": torch.float32}
    if not dy2.is_cuda and dy2.dtype != torch.float32:
        dy2, x2 = dy2.float(), x2.float()          # (CPU: no fp32-output bf16 GEMM; same sum, fp32 operands)
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


def linear(x, weight, bias=None):
    """``F.linear(x, weight, bias)`` for the projections of the train path: same forward GEMM (autocast semantics
    included), weight gradient posed as a token-split batched GEMM in fp32.  Without gradients: F.linear itself."""
    if not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)) or x.dim() < 2:
        return F.linear(x, weight, bias)
    return _LinearFunction.apply(x, weight, bias)


_MLP_PAD = 128          # the hidden dimension of the channel mixer is padded to a multiple of this (+ the bias column)


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
        be = _BACKEND
        cd = x.dtype
        if x.is_cuda and torch.is_autocast_enabled("cuda"):
            cd = torch.get_autocast_dtype("cuda")
        H, d_out, d_in = w_out.shape[1], w_out.shape[0], w_in.shape[1]
        Hp = (H + _MLP_PAD) // _MLP_PAD * _MLP_PAD                    # > H: room for the bias column
        dev = x.device
        x2 = x.reshape(-1, d_in).to(cd).contiguous()
        with torch.autocast(dev.type, enabled=False):
            Wi = torch.empty(2, Hp, d_in, dtype=cd, device=dev)
            Wi[:, H:].zero_()
            Wi[:, :H].copy_(w_in.detach().view(2, H, d_in))
            bi = torch.zeros(2, Hp, dtype=cd, device=dev)
            if b_in is not None:
                bi[:, :H].copy_(b_in.detach().view(2, H))
            Wo = torch.empty(d_out, Hp, dtype=cd, device=dev)
            Wo[:, H:].zero_()
            Wo[:, :H].copy_(w_out.detach())
            if b_out is not None:
                bi[:, H] = _mlp_one(cd, dev)
                Wo[:, H].copy_(b_out.detach())
            u = torch.addmm(bi.view(-1), x2, Wi.view(2 * Hp, d_in).t())
            h = torch.empty(x2.shape[0], Hp, dtype=cd, device=dev)
            _check(be.lib.lina_swiglu(_ptr(u), _ptr(h), x2.shape[0], Hp, u.stride(0), h.stride(0), _dt(u), be.stream(u)))
            y = torch.mm(h, Wo.t())
        ctx.save_for_backward(x2, u, h, Wi, Wo)
        ctx.meta = (x.shape, x.dtype, H, Hp, w_in.dtype, None if b_in is None else b_in.dtype, w_out.dtype,
                    None if b_out is None else b_out.dtype)
        return y.view(*x.shape[:-1], d_out)

    @staticmethod
    def backward(ctx, dy):
        x2, u, h, Wi, Wo = ctx.saved_tensors
        x_shape, xdt, H, Hp, widt, bidt, wodt, bodt = ctx.meta
        be = _BACKEND
        d_out, d_in = Wo.shape[0], x2.shape[1]
        M = x2.shape[0]
        with torch.autocast(x2.device.type, enabled=False):
            dy2 = dy.reshape(M, d_out).to(x2.dtype).contiguous()
            dh = torch.mm(dy2, Wo)
            dWo = linear_weight_grad(dy2, h)                                         # [d_out, Hp] fp32; column H = db_out
            du = torch.empty_like(u)
            part = torch.empty(int(be.lib.lina_swiglu_bwd_partials(M)), 2 * Hp, dtype=torch.float32, device=u.device)
            _check(be.lib.lina_swiglu_bwd_colsum(_ptr(dh), _ptr(u), _ptr(du), _ptr(part), M, Hp, u.stride(0), dh.stride(0),
                                                 du.stride(0), _dt(u), be.stream(u)))
            dx = torch.mm(du, Wi.view(2 * Hp, d_in)).view(x_shape).to(xdt) if ctx.needs_input_grad[0] else None
            dWi = linear_weight_grad(du, x2).view(2, Hp, d_in)
            dw_in = dWi[:, :H].reshape(2 * H, d_in).to(widt)
            db_in = None if bidt is None else _sum_partials(part).view(2, Hp)[:, :H].reshape(2 * H).to(bidt)
            dw_out = dWo[:, :H].to(wodt)
            db_out = None if bodt is None else dWo[:, H].to(bodt)
        return dx, dw_in, db_in, dw_out, db_out


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
    _BACKEND.require(x, w_in, w_out)
    return _SwiGLUMLPFunction.apply(x, w_in, b_in, w_out, b_out)


class _GateLogSigmoidFunction(torch.autograd.Function):
    """K12: logsigmoid(x) / normalizer (optionally clamped) and its gradient, one pass each."""

    @staticmethod
    def forward(ctx, x, normalizer, clamp_min):
        be = _BACKEND
        y = torch.empty_like(x)
        cm = float("nan") if clamp_min is None else float(clamp_min)
        _check(be.lib.lina_gate_logsigmoid(_ptr(x), None, _ptr(y), x.numel(), float(normalizer), cm, _dt(x), be.stream(x)))
        ctx.save_for_backward(x)
        ctx.args = (float(normalizer), cm)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        be = _BACKEND
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
    _BACKEND.require(x)
    return _GateLogSigmoidFunction.apply(x.contiguous(), float(normalizer), clamp_min).view(x.shape)


class _GateLowRankFunction(torch.autograd.Function):
    """K12b: logsigmoid(lr W^T + b) / normalizer in one pass; backward d(lr), dW, db without the [R, C] pre-activation."""

    @staticmethod
    def forward(ctx, lr, w, b, normalizer, clamp_min):
        be = _BACKEND
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
        be = _BACKEND
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
    _BACKEND.require(lr, weight, bias)
    return _GateLowRankFunction.apply(lr.to(gemm_dt), weight, bias, float(normalizer), clamp_min)


def _sum_vector(v):
    """Sum of a long fp32 vector as a 0-dim tensor through K13 (rows of 4 as the "partials") and a 4-element tail -- no
    multi-block torch reduction (see ``column_sum``)."""
    n = v.numel()
    if n < 8 or n % 4 or not fused_ops_available(v) or v.dtype != torch.float32:
        return v.sum()
    w = 256 if n % 256 == 0 else 4
    return _sum_partials(v.contiguous().view(n // w, w)).sum()


class _CrossEntropyFunction(torch.autograd.Function):
    """K14: mean cross-entropy over the rows whose target is not ``ignore_index``; rows [N, V] of the logits' own dtype."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        be = _BACKEND
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
        be = _BACKEND
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
    _BACKEND.require(logits, target)
    lg = logits if (logits.stride(1) == 1 and logits.data_ptr() % 16 == 0) else logits.contiguous()
    return _CrossEntropyFunction.apply(lg, target.contiguous(), ignore_index)


# --------------------------------------------------------------------------- codec head (K6)
def _embed_sum_launch(table, flat, out=None):
    be = _BACKEND
    Q, n_emb, d = table.shape
    N = flat.shape[1]
    if out is None:
        out = torch.empty(N, d, dtype=table.dtype, device=table.device)
    elif tuple(out.shape) != (N, d) or out.dtype != table.dtype or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous {table.dtype} tensor [{N}, {d}]")
    _check(be.lib.lina_embed_sum(_ptr(flat), _ptr(table.contiguous()), _ptr(out), Q, N, n_emb, d, _dt(table),
                                 be.stream(table)))
    return out


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
    be = _BACKEND
    be.require(table, idx)
    Q, n_emb, d = table.shape
    if idx.shape[0] != Q or idx.dtype != torch.int64:
        raise ValueError("idx must be int64 [Q, ...]")
    flat = idx.reshape(Q, -1).contiguous()
    if _needs_grad(table):
        return _EmbedSumFunction.apply(table, flat, padding_idx).view(*idx.shape[1:], d)
    return _embed_sum_launch(table, flat, out).view(*idx.shape[1:], d)


def argmax_rows(logits, out=None):
    """Greedy pick over the last dim, lowest index on ties (topk_sampling(k=1), reference tools.py:38-44)."""
    be = _BACKEND
    be.require(logits)
    lg = _inner_contig(logits)
    n = lg.shape[-1]
    lg2 = lg.reshape(-1, n)
    if out is None:
        out = torch.empty(lg2.shape[0], dtype=torch.int64, device=lg.device)
    _check(be.lib.lina_argmax_rows(_ptr(lg2), _ptr(out), lg2.shape[0], n, lg2.stride(0), _dt(lg2), be.stream(lg2)))
    return out.view(logits.shape[:-1])


def greedy_pick_embed(logits, table, x_out, tok_log, step, counter, x_packed=None):
    """K6d (lina_greedy_pick_embed): arg-max per quantizer of ``logits [B, Q, L]``, the picks logged at
    ``tok_log[step[0]]`` ([max_steps, Q, B] int64), the next input ``x_out [B, d] = sum_q table[q, pick_q]`` and
    ``step[0] += 1`` -- one launch.  ``counter``: int32 [1], zero."""
    be = _BACKEND
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
                                         _ptr(step), _ptr(counter), B, Q, L, n_emb, d, tok_log.shape[0], _dt(table),
                                         be.stream(table)))


def sample_pick_embed(logits, table, x_out, tok_log, step, counter, n_sampled: int, k: int, temp: float = 1.0,
                      seed: int = 0, x_packed=None):
    """K6e (lina_sample_pick_embed): greedy_pick_embed for the reference's default generation mode -- quantizers
    ``q < n_sampled`` are sampled (top-``k``, temperature, the draw of row ``b*Q + q`` of topk_sample_rows at the same
    (seed, step)), the others take the arg-max; token log, next-input embedding and ``step[0] += 1`` in the same launch."""
    be = _BACKEND
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
                                         _ptr(x_packed), _ptr(tok_log), _ptr(step), _ptr(counter), B, Q, L, n_emb, d,
                                         tok_log.shape[0], int(n_sampled), int(k), float(temp),
                                         int(seed) & 0xFFFFFFFFFFFFFFFF, _dt(table), be.stream(table)))


# --------------------------------------------------------------------------- decode-step fusions
def topk_sample_rows(logits, k: int, temp: float = 1.0, u: Optional[torch.Tensor] = None, seed: int = 0,
                     step: Optional[torch.Tensor] = None, out=None):
    """K6c: one top-k / temperature sample per row of ``logits [..., n]`` -> int64 ``[...]`` (reference
    tools.py:38-44 for k > 1).  ``u``: fp32 uniforms [rows] (else hashed from (seed, step[0], row); ``step`` is a
    device int64 tensor)."""
    be = _BACKEND
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
    be = _BACKEND
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
    be = _BACKEND
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
    be = _BACKEND
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
    be = _BACKEND
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
    be = _BACKEND
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
    be = _BACKEND
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
    be = _BACKEND
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
    be = _BACKEND
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
    window).  Call gla_decode_window_flush before anybody else reads ``state``."""
    be = _BACKEND
    be.require(q, k, v, gk, state, gate, norm_weight, og, hist_k, hist_c, hist_v, step, origin)
    B, H, Dk = q.shape
    Dv = v.shape[-1]
    if state.dtype != torch.float32 or not state.is_contiguous() or tuple(state.shape) != (B, H, Dk, Dv):
        raise ValueError("state must be contiguous fp32 [B,H,Dk,Dv]")
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
    _check(be.lib.lina_gla_decode_window(_ptr(q), _ptr(k), _ptr(v), _ptr(gk), _ptr(state), _ptr(gate),
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
    be = _BACKEND
    be.require(state, hist_k, hist_c, hist_v)
    B, H, Dk, Dv = state.shape
    _check(be.lib.lina_gla_decode_window_flush(_ptr(state), _ptr(hist_k), _ptr(hist_c), _ptr(hist_v), int(n_pending),
                                               B, H, Dk, Dv, be.stream(state)))
    return state


def cross_att_step1(q_lin, ln_w, ln_b, ln_eps, kk, pe, att1, xp, scale):
    """Blind cross-attention step 1 (see lina_gla.h).  att1: [B,T_txt] view (row stride free), written in place."""
    be = _BACKEND
    be.require(q_lin, ln_w, ln_b, kk, pe, att1, xp)
    B, d = q_lin.shape
    Tn = kk.shape[1]
    _check(be.lib.lina_cross_att_step1(_ptr(q_lin), _ptr(ln_w), _ptr(ln_b), float(ln_eps), _ptr(kk), _ptr(pe),
                                       _ptr(att1), att1.stride(0), _ptr(xp), B, Tn, d, float(scale), _dt(q_lin),
                                       be.stream(q_lin)))


def cross_att_step2(xp, pe, vv, att2, x, scale):
    """Blind cross-attention step 2: x += softmax(xp . pe^T * scale) . vv   (see lina_gla.h)."""
    be = _BACKEND
    be.require(xp, pe, vv, att2, x)
    B, d = xp.shape
    Tn = vv.shape[1]
    _check(be.lib.lina_cross_att_step2(_ptr(xp), _ptr(pe), _ptr(vv), _ptr(att2), att2.stride(0), _ptr(x), B, Tn, d,
                                       float(scale), _dt(xp), be.stream(xp)))


def cross_scores(q_lin, ln_w, ln_b, ln_eps, kk, scores, scale):
    """scores[b,t] = scale * <LayerNorm(q_lin[b]), kk[b,t,:]> (fp32 [B,T_txt]); see lina_gla.h."""
    be = _BACKEND
    be.require(q_lin, ln_w, ln_b, kk, scores)
    B, d = q_lin.shape
    _check(be.lib.lina_cross_scores(_ptr(q_lin), _ptr(ln_w), _ptr(ln_b), float(ln_eps), _ptr(kk), _ptr(scores), B,
                                    kk.shape[1], d, float(scale), _dt(q_lin), be.stream(q_lin)))


def cross_scores_softmax(q_lin, ln_w, ln_b, ln_eps, kk, att, attc, scale):
    """att[b,:Tn] = softmax(scale * <LayerNorm(q_lin[b]), kk[b,t,:]>) into the strided ``att`` rows and the contiguous
    zero-padded copy attc [B,Tp] -- cross_scores + softmax_rows in one launch (lina_cross_scores_softmax)."""
    be = _BACKEND
    be.require(q_lin, ln_w, ln_b, kk, att, attc)
    B, d = q_lin.shape
    Tn = kk.shape[1]
    _check(be.lib.lina_cross_scores_softmax(_ptr(q_lin), _ptr(ln_w), _ptr(ln_b), float(ln_eps), _ptr(kk), _ptr(att),
                                            att.stride(0), _ptr(attc), B, Tn, attc.shape[1], d, float(scale), _dt(q_lin),
                                            be.stream(q_lin)))


def softmax_weighted_rows_add(scores, scale, att, vv, x, x_packed=None):
    """att[b,:Tn] = softmax(scores[b,:Tn] * scale);  x[b,:] += att[b,:] . vv[b]  -- softmax_rows + weighted_rows_add in
    one launch.  With ``x_packed`` the residual stream is the fragment-major buffer (``x`` is not touched)."""
    be = _BACKEND
    be.require(scores, att, vv, x, x_packed)
    B, Tn, d = vv.shape
    if scores.dtype != vv.dtype or att.dtype != vv.dtype:
        raise TypeError("scores / att / vv must share the model dtype")
    _check(be.lib.lina_softmax_weighted_rows_add(_ptr(scores), scores.stride(0), float(scale), _ptr(att), att.stride(0),
                                                 _ptr(vv), _ptr(x), _ptr(x_packed), B, Tn, d, _dt(vv), be.stream(vv)))


def softmax_pe_rows(scores, att, pe, xp, xp_packed=None):
    """att[b,:Tn] = softmax(scores[b,:Tn]) (fp32 scores, already scaled);  xp[b,:] = att[b,:] . pe[:Tn,:] -- one launch;
    ``xp_packed``: also the fragment-major copy of xp (the A operand of the next projection)."""
    be = _BACKEND
    be.require(scores, att, pe, xp, xp_packed)
    B, Tn = scores.shape
    d = pe.shape[1]
    if scores.dtype != torch.float32 or scores.stride(1) != 1:
        raise TypeError("scores must be fp32 [B, Tn] with contiguous rows")
    if att.dtype != pe.dtype or xp.dtype != pe.dtype or pe.shape[0] < Tn or not pe.is_contiguous() or not xp.is_contiguous():
        raise TypeError("att / pe / xp must share the model dtype; pe [>= Tn, d] and xp [B, d] contiguous")
    if xp_packed is not None and xp_packed.numel() < packed_numel(B, d):
        raise ValueError("packed xp buffer is too small")
    _check(be.lib.lina_softmax_pe_rows(_ptr(scores), scores.stride(0), _ptr(att), att.stride(0), _ptr(pe), _ptr(xp),
                                       _ptr(xp_packed), B, Tn, d, _dt(pe), be.stream(pe)))


def softmax_rows(x, scale, att, attc, Tn):
    """att[b,:Tn] = softmax(x[b,:Tn]*scale) into the strided `att` rows and the contiguous padded copy attc [B,Tp]."""
    be = _BACKEND
    be.require(x, att, attc)
    B = x.shape[0]
    _check(be.lib.lina_softmax_rows(_ptr(x), x.stride(0), _dt(x), float(scale), _ptr(att), att.stride(0), _ptr(attc), B,
                                    Tn, attc.shape[1], _dt(attc), be.stream(x)))


def weighted_rows_add(attc, vv, x, x_packed=None):
    """x[b,:] += sum_t attc[b,t] * vv[b,t,:].  With ``x_packed`` the residual stream is the fragment-major buffer: it is
    updated in place there and ``x`` is not touched."""
    be = _BACKEND
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
    be = _BACKEND
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
    be = _BACKEND
    be.require(frames, window)
    B, T, win = frames.shape
    frames = frames.float().contiguous()
    window = window.float().contiguous()
    pad = (win - hop) // 2
    y = torch.empty(B, (T - 1) * hop + win - 2 * pad, dtype=torch.float32, device=frames.device)
    _check(be.lib.lina_istft_ola(_ptr(frames), _ptr(window), _ptr(y), B, T, win, int(hop), be.stream(frames)))
    return y
