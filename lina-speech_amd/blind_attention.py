"""Position-only ("blind") cross-attention between the codec stream and the text
(reference model/crossatt.py:76-155 with ConvPos :21-32 / SinPos :35-48).

SURVEY.md 8(f) f-1: adjacent to the hot path -- called every decode step.  The text-side
projections, LayerNorms and the positional table do not depend on the step, so
``prepare(ctx)`` computes them ONCE per utterance and ``forward`` reuses them (the
reference recomputes them each step, crossatt.py:114-116,125-127; results are identical).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops


def attention_with_weights(q, k, v, mask=None):
    """softmax(q k^T / sqrt(d)) v, also returning the weights (reference crossatt.py:13-19)."""
    w = q @ k.transpose(-2, -1) * (1.0 / math.sqrt(q.size(-1)))
    if mask is not None:
        w = w.masked_fill(~mask, -torch.finfo(w.dtype).max)
    w = torch.softmax(w, dim=-1)
    return w @ v, w


def train_attention(q, k, v, mask=None, dropout_p: float = 0.0):
    """``F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=p)`` (what the reference's training branch calls,
    crossatt.py:141-144) for THIS shape -- one head of width d (1024: above every fused kernel's head size, so torch takes its
    math path), thousands of queries against a few dozen text positions.  The math path scales q AND k by d^-1/4 before the
    product and expands a shared operand per batch row: six passes over [B, 1, T, d] tensors for products whose results are
    [B, 1, T, Ttxt].  Here the scale is applied to the small score tensor and shared operands stay shared (a [1, 1, Ttxt, d]
    operand folds the batch into ONE GEMM); mask, softmax and dropout as in the math path (bool mask -> -inf)."""
    w = torch.matmul(q, k.transpose(-2, -1)) * (1.0 / math.sqrt(q.size(-1)))
    if mask is not None:
        w = w.masked_fill(~mask, float("-inf")) if mask.dtype == torch.bool else w + mask
    w = torch.softmax(w, dim=-1)
    if dropout_p > 0.0:
        w = torch.dropout(w, dropout_p, True)
    return torch.matmul(w, v)        # (under autocast the fp32 softmax output is cast with v; otherwise the dtypes agree)


class ConvPos(nn.Module):
    def __init__(self, dim: int, max_seq_len: int = 2000, kernel_size: int = 31):
        super().__init__()
        self.embed = nn.Embedding(max_seq_len, dim)
        self.dw_conv = nn.Conv1d(dim, dim, kernel_size, groups=dim, padding="same")

    def forward(self, pos):
        return self.dw_conv(self.embed(pos).transpose(1, 2)).transpose(1, 2)


class SinPos(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, pos):
        e = 2 * torch.arange(self.dim // 2, device=pos.device) / self.dim
        p = pos.unsqueeze(-1) * torch.pow(10000, -e)[None, None, :]
        return torch.sin(torch.cat((p, p + math.pi / 2), dim=2))


class BlindCrossAttention(nn.Module):
    def __init__(self, q_dim, k_dim, att_dim, heads, pos_net, dropout=0.1, pos_dim=64, rotary=False,
                 pos_type="sinusoidal"):
        super().__init__()
        if rotary:
            raise NotImplementedError("rotary cross-attention is not part of the 'convblind' architecture")
        self.q, self.k, self.v = nn.Linear(q_dim, att_dim), nn.Linear(k_dim, att_dim), nn.Linear(k_dim, att_dim)
        self.pos_net = pos_net
        self.pos_embed = ConvPos(pos_dim) if pos_type == "convolutional" else SinPos(pos_dim)
        assert att_dim % heads == 0
        self.ln_q, self.ln_k, self.ln_v = nn.LayerNorm(att_dim), nn.LayerNorm(att_dim), nn.LayerNorm(att_dim)
        self.rotary = None
        self.dropout_att = nn.Dropout(dropout)
        self._prepared = None

    @staticmethod
    def _norm(ln: nn.LayerNorm, x):
        """``ln(x)`` on the fused LayerNorm (K10) where the ops run: under autocast nn.LayerNorm casts its bf16 input to fp32,
        normalises, and the attention product casts the fp32 result back -- two passes over [B, T, d] each way that K10's
        bf16-in / bf16-out form (fp32 statistics inside, the same rounding point) does not make."""
        if ops.fused_ops_available(x) and ln.weight is not None and ln.bias is not None:
            return ops.layer_norm(x, ln.weight, ln.bias, ln.eps)
        return ln(x)

    def prepare(self, ctx, pos=None):
        """Step-invariant text side: (k, v, pos_emb), each [B|1, 1, Ttxt, d]."""
        k = self._norm(self.ln_k, ops.linear(ctx, self.k.weight, self.k.bias)).unsqueeze(1)   # (ops.linear: nn.Linear's forward;
        v = self._norm(self.ln_v, ops.linear(ctx, self.v.weight, self.v.bias)).unsqueeze(1)   #  bias gradient as K13a / K13)
        if pos is None:
            pos = torch.arange(ctx.shape[1], device=ctx.device).unsqueeze(0)
        return k, v, self.pos_embed(pos).unsqueeze(1)

    def forward(self, q, k, mask=None, time_step=None, pos=None, prepared=None, **kwargs):
        kk, vv, pe = prepared if prepared is not None else self.prepare(k, pos)
        qq = self._norm(self.ln_q, ops.linear(q, self.q.weight, self.q.bias)).unsqueeze(1)
        if mask is not None:
            mask = mask.unsqueeze(1)
        if self.training:
            sdpa = lambda a, b, c: (train_attention(a, b, c, mask, self.dropout_att.p), None)
        else:
            sdpa = lambda a, b, c: attention_with_weights(a, b, c, mask=mask)
        x, att1 = sdpa(qq, kk, pe)
        x = self.pos_net(x.squeeze(1), **kwargs)
        x = (x[0] if type(x) is tuple else x).unsqueeze(1)
        x, att2 = sdpa(x, pe, vv)
        att = torch.cat((att1, att2), dim=1) if att1 is not None else None
        return x.squeeze(1), att
