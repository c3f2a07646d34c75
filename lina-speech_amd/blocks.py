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
        a, b = self.p_in(x).chunk(2, dim=-1)
        return self.p_out(F.silu(a) * b)


class MixingBlock(nn.Module):
    """Pre-norm residual block: x += tmix(norm1(x), **kw); x += cmix(norm2(x)); dropout."""

    def __init__(self, tmix: Callable, cmix: Callable, norm: Callable, dropout: float = 0.0):
        super().__init__()
        self.tmix, self.cmix = tmix(), cmix()
        self.norm1, self.norm2 = norm(), norm()
        self.drop = nn.Dropout(dropout)

    def forward(self, x, **kwargs):
        y = self.tmix(self.norm1(x), **kwargs)
        x = (y[0] if type(y) is tuple else y) + x
        x = self.cmix(self.norm2(x)) + x
        return self.drop(x)


class SelfAttention(nn.Module):
    """Bidirectional multi-head self-attention of the text encoder (no rotary: the decode path of
    this package never needs it; ask for rotary and it raises)."""

    def __init__(self, dim: int, heads: int, rotary: bool = False, is_causal: bool = False):
        super().__init__()
        if rotary:
            raise NotImplementedError("rotary text encoder is outside the generation hot path (SURVEY 2, #11)")
        assert dim % heads == 0
        self.qkv = nn.Linear(dim, 3 * dim)
        self.heads, self.is_causal = heads, is_causal

    def forward(self, x, mask=None, pos=None, **kwargs):
        B, N, D = x.shape
        q, k, v = self.qkv(x).view(B, N, 3, self.heads, D // self.heads).permute(2, 0, 3, 1, 4)
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, is_causal=self.is_causal)
        return y.transpose(1, 2).reshape(B, N, D)


class TextEncoder(nn.Module):
    def __init__(self, dim: int, heads: int, n_layers: int = 4, dropout: float = 0.1, rotary: bool = False):
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
