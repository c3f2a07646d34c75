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


def _mm_f32(a, b):
    """a [.., m, k] @ b [.., k, n] with an fp32 RESULT for operands of one GEMM dtype (bf16 operands: the matrix core's fp32
    accumulators written out unrounded -- ``out_dtype=`` of recent torch; elsewhere the same sums on upcast operands).  ``b`` may
    have a leading batch of 1: the batch then folds into ONE GEMM."""
    from .autograd import _mm_has_out_dtype
    kw = {}
    if a.dtype != torch.float32:
        if a.is_cuda and _mm_has_out_dtype(a.device):
            kw = {"out_dtype": torch.float32}
        else:
            a, b = a.float(), b.float()
    if b.shape[0] == 1:
        return torch.mm(a.reshape(-1, a.shape[-1]), b[0], **kw).view(*a.shape[:-1], b.shape[-1])
    return torch.bmm(a, b, **kw)


class _TrainAttentionFunction(torch.autograd.Function):
    """One head of softmax(q k^T / sqrt(d) + mask) v for q [B, T, d], k [B|1, S, d], v [B|1, S, dv], S small (the text) -- the
    arithmetic of torch's SDPA math path on low-precision inputs (scores and probabilities in fp32) without its passes over
    [B, T, d]: the score GEMM writes its fp32 accumulators, the probabilities are rounded to the GEMM dtype only as the operand of
    the second product (as every fused attention kernel does); backward the same way round (dP in fp32, dS rounded as an operand)."""

    @staticmethod
    def forward(ctx, q, k, v, mask):
        scale = 1.0 / math.sqrt(q.shape[-1])
        s = _mm_f32(q, k.transpose(-2, -1)) * scale                       # [B, T, S] fp32
        if mask is not None:
            s = s.masked_fill(~mask, float("-inf"))
        w = torch.softmax(s, dim=-1)
        wb = w.to(v.dtype)
        out = torch.matmul(wb, v)                                         # [B, T, dv] (v [1, S, dv]: one GEMM over B T rows)
        ctx.save_for_backward(q, k, v, w)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, w = ctx.saved_tensors
        dout = dout.to(v.dtype).contiguous()
        B, T, S = w.shape
        dq = dk = dv = None
        wb = w.to(v.dtype)
        # a SHARED text-side operand ([1, S, d]: the positional table) gathers its gradient over all B T rows into an [S, d]
        # result -- as one GEMM the library runs that [64, 32768] x [32768, 1024] product in 115 us; token-split (the form of
        # every weight gradient of the train path, ops.linear_weight_grad) in 28
        from .autograd import linear_weight_grad
        if ctx.needs_input_grad[2]:
            dv = (linear_weight_grad(wb.reshape(B * T, S), dout.reshape(B * T, -1)).to(v.dtype).unsqueeze(0) if v.shape[0] == 1
                  else torch.bmm(wb.transpose(1, 2), dout))
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dw = _mm_f32(dout, v.transpose(-2, -1))                       # [B, T, S] fp32
            ds = (w * (dw - (dw * w).sum(-1, keepdim=True)) * ctx.scale).to(q.dtype)
            if ctx.needs_input_grad[0]:
                dq = torch.matmul(ds, k)
            if ctx.needs_input_grad[1]:
                dk = (linear_weight_grad(ds.reshape(B * T, S), q.reshape(B * T, -1)).to(k.dtype).unsqueeze(0) if k.shape[0] == 1
                      else torch.bmm(ds.transpose(1, 2), q))
        return dq, dk, dv, None


def train_attention(q, k, v, mask=None, dropout_p: float = 0.0):
    """``F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=p)`` (what the reference's training branch calls,
    crossatt.py:141-144) for THIS shape -- [B, 1, T, d]: one head of width d (1024: above every fused kernel's head size, so torch
    takes its math path), thousands of queries against a few dozen text positions.  The math path upcasts q, k, v to fp32, scales q
    AND k by d^-1/4 and expands a shared operand per batch row: six passes over [B, 1, T, d] tensors for products whose results are
    [B, 1, T, Ttxt].  ``_TrainAttentionFunction`` keeps the math path's precision where it matters (fp32 scores and
    probabilities) on the low-precision operands as they are.  Dropout on the weights (not used by the L169 configuration) and
    non-boolean masks take the written-out fp32 form."""
    cd = q.dtype
    if q.is_cuda and torch.is_autocast_enabled("cuda"):
        cd = torch.get_autocast_dtype("cuda")
    ok = (dropout_p == 0.0 and q.dim() == 4 and q.shape[1] == 1 and k.shape[1] == 1 and v.shape[1] == 1
          and (mask is None or mask.dtype == torch.bool) and k.shape[0] in (1, q.shape[0]) and v.shape[0] in (1, q.shape[0]))
    if ok:
        m3 = None
        if mask is not None:
            m3 = mask.expand(q.shape[0], 1, q.shape[2], k.shape[2])[:, 0]
        with torch.autocast(q.device.type, enabled=False):
            # (squeeze, not q[:, 0]: a select's backward zero-fills a full-size tensor and copies the gradient into it)
            return _TrainAttentionFunction.apply(q.squeeze(1).to(cd), k.squeeze(1).to(cd), v.squeeze(1).to(cd), m3).unsqueeze(1)
    with torch.autocast(q.device.type, enabled=False):
        qf, kf, vf = q.float(), k.float(), v.float()
        w = torch.matmul(qf, kf.transpose(-2, -1)) * (1.0 / math.sqrt(q.size(-1)))
        if mask is not None:
            w = w.masked_fill(~mask, float("-inf")) if mask.dtype == torch.bool else w + mask
        w = torch.softmax(w, dim=-1)
        if dropout_p > 0.0:
            w = torch.dropout(w, dropout_p, True)
        return torch.matmul(w, vf).to(cd)


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
