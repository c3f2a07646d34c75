"""Residual block glue around the mixer: MixingBlock / SwiGLU (reference
model/base_blocks.py:42-69) and the small non-causal text encoder that runs once per
utterance before the decode loop (reference model/encoder.py:14-43, base_blocks.py:9-40;
plain torch SDPA -- out of the hot path, kept only so a full model can be assembled).
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.nn as nn
import torch.nn.functional as F


class SwiGLU(nn.Module):
    """p_out(silu(a) * b), (a, b) = p_in(x).chunk(2); hidden = d*4//3, both linears biased."""

    def __init__(self, d_model: int):
        super().__init__()
        self.hidden = d_model * 4 // 3
        self.p_in = nn.Linear(d_model, self.hidden * 2)
        self.p_out = nn.Linear(self.hidden, d_model)

    def forward(self, x):
        from . import ops
        # train path: one node on padded operands (K11 / K11c gate kernels, token-split weight gradients, the biases
        # riding in the GEMMs); without gradients / off-device: the three ops
        return ops.swiglu_mlp(x, self.p_in.weight, self.p_in.bias, self.p_out.weight, self.p_out.bias)


class MixingBlock(nn.Module):
    """Pre-norm residual block: x += tmix(norm1(x), **kw); x += cmix(norm2(x)); dropout."""

    def __init__(self, tmix: Callable, cmix: Callable, norm: Callable, dropout: float = 0.0):
        super().__init__()
        self.tmix, self.cmix = tmix(), cmix()
        self.norm1, self.norm2 = norm(), norm()
        self.drop = nn.Dropout(dropout)

    def can_defer(self, x) -> bool:
        """True when this block can take / hand on a PENDING branch (``forward(..., _pending=, _defer=)``): both norms are
        LayerNorms on the fused kernel and the block's dropout is the identity."""
        from . import ops
        return (isinstance(self.norm1, nn.LayerNorm) and isinstance(self.norm2, nn.LayerNorm) and ops.fused_ops_available(x)
                and not (self.training and self.drop.p > 0))

    def forward(self, x, _pending=None, _defer: bool = False, **kwargs):
        """``x += tmix(norm1(x)); x += cmix(norm2(x)); dropout`` (reference base_blocks.py:65-69).
        A stack of blocks may chain them without materialising the stream in between: ``_pending`` is a branch output the
        previous block has NOT yet added to ``x`` (it is added inside this block's norm1 pass, K10), and with ``_defer``
        the block returns ``(x, branch)`` with its own last branch still to be added -- ``x + branch`` is the block's
        output.  Both default to the plain form."""
        from . import ops
        ln = isinstance(self.norm1, nn.LayerNorm) and isinstance(self.norm2, nn.LayerNorm) and ops.fused_ops_available(x)
        if not ln:
            if _pending is not None:
                x = x + _pending
            y = self.tmix(self.norm1(x), **kwargs)
            x = (y[0] if type(y) is tuple else y) + x
            x = self.drop(self.cmix(self.norm2(x)) + x)
            return (x, None) if _defer else x
        # K10: the norms run on the HIP kernel (fp32 stream in, GEMM-dtype operand out) and the residual adds ride in
        # their passes -- same arithmetic as the lines above
        n1, n2 = self.norm1, self.norm2
        if _pending is not None:
            h, x = ops.layer_norm(x, n1.weight, n1.bias, n1.eps, residual=_pending)
        else:
            h = ops.layer_norm(x, n1.weight, n1.bias, n1.eps)
        y = self.tmix(h, **kwargs)
        y = y[0] if type(y) is tuple else y
        h, x = ops.layer_norm(x, n2.weight, n2.bias, n2.eps, residual=y)
        c = self.cmix(h)
        if _defer and self.can_defer(x):
            return x, c
        x = self.drop(c + x)
        return (x, None) if _defer else x


class RotaryEmbedding(nn.Module):
    """Rotary position embedding with the parameter name and arithmetic of ``rotary_embedding_torch.RotaryEmbedding``
    (lucidrains; the reference imports it at model/base_blocks.py:6 and lists it un-pinned in requirements.txt:4; the
    package is absent from this image, so this restates its published defaults: ``freqs = theta^(-2i/dim)``, theta
    10000, kept as a non-trainable Parameter ``freqs`` -> state-dict key ``...rotary.freqs``; angles interleaved
    pairwise; only the first ``dim`` channels of a head are rotated).  Runs once per utterance in the text encoder --
    plain torch, not a kernel."""

    def __init__(self, dim: int, theta: float = 10000.0):
        super().__init__()
        self.freqs = nn.Parameter(1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim)),
                                  requires_grad=False)

    def forward(self, pos):
        """positions [...] -> angles [..., dim] (each frequency repeated for its channel pair)."""
        return (pos.to(self.freqs.dtype)[..., None] * self.freqs).repeat_interleave(2, dim=-1)

    def rotate_queries_or_keys(self, t, offset: int = 0):
        pos = torch.arange(t.shape[-2], device=t.device) + offset
        return apply_rotary_emb(self.forward(pos), t)


def apply_rotary_emb(angles, t):
    rot = angles.shape[-1]
    a = angles.to(t.dtype)
    head, rest = t[..., :rot], t[..., rot:]
    pair = head.reshape(*head.shape[:-1], rot // 2, 2)
    half = torch.stack((-pair[..., 1], pair[..., 0]), dim=-1).reshape(head.shape)
    return torch.cat((head * a.cos() + half * a.sin(), rest), dim=-1)


class SelfAttention(nn.Module):
    """Bidirectional multi-head self-attention of the text encoder (reference model/base_blocks.py:9-40; rotary on
    q and k by default like the reference)."""

    def __init__(self, dim: int, heads: int, rotary: bool = True, is_causal: bool = False):
        super().__init__()
        assert dim % heads == 0
        self.qkv = nn.Linear(dim, 3 * dim)
        self.heads, self.is_causal = heads, is_causal
        self.rotary = RotaryEmbedding((dim // heads) // 2) if rotary else None

    def forward(self, x, mask=None, pos=None, time_step: int = 0, **kwargs):
        B, N, D = x.shape
        from . import ops
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias)      # nn.Linear's forward; bias gradient as K13a / K13
        q, k, v = qkv.view(B, N, 3, self.heads, D // self.heads).permute(2, 0, 3, 1, 4)
        if self.rotary is not None:
            if pos is not None:
                ang = self.rotary(pos).unsqueeze(1)
                q, k = apply_rotary_emb(ang, q), apply_rotary_emb(ang, k)
            else:
                q = self.rotary.rotate_queries_or_keys(q, offset=time_step)
                k = self.rotary.rotate_queries_or_keys(k)
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, is_causal=self.is_causal)
        return y.transpose(1, 2).reshape(B, N, D)


class TextEncoder(nn.Module):
    def __init__(self, dim: int, heads: int, n_layers: int = 4, dropout: float = 0.1, rotary: bool = True):
        super().__init__()
        self.sa = nn.ModuleList([
            MixingBlock(lambda: SelfAttention(dim, heads, rotary=rotary), lambda: SwiGLU(dim),
                        lambda: nn.LayerNorm(dim), dropout) for _ in range(n_layers)])

    def forward(self, x, mask=None, pos=None):
        if mask is not None:  # [b,n,m] -> [b,1,n,m], keep the diagonal attendable
            eye = torch.eye(mask.shape[-1], device=x.device, dtype=torch.bool)
            mask = (mask.bool() | eye)[:, None]
        for blk in self.sa:
            x = blk(x, mask=mask, pos=pos)
        return x
