"""Codec-token side of the model: stacked embedding tables, the logits head and the
sampling / RVQ-delay helpers (reference model/multiembed.py:7-23, the EinMix head of
model/modeling_lina.py:51-57, model/tools.py:38-67).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class MultiEmbedding(nn.Module):
    """``n_level`` embedding tables of identical shape in one parameter ``weight [q, n_emb, d]``.
    ``forward(idx[q, ...]) -> [q, ..., d]`` (per-level gather, like the reference);
    ``embed_sum(idx) -> [..., d]`` is the fused gather + sum over levels the decode loop uses (K6a).
    NB: ``padding_idx`` does not zero a row (reference initialises with normal_ afterwards), but -- as in the
    reference's ``F.embedding(padding_idx=...)`` under vmap (model/multiembed.py:21-23) -- that row receives NO
    gradient."""

    def __init__(self, n_level: int, n_emb: int, d_emb: int, padding_idx=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_level, n_emb, d_emb))
        self.n_level, self.padding_idx = n_level, padding_idx
        nn.init.normal_(self.weight)

    def forward(self, idx):
        if self.n_level == 1:                   # one level: the gather itself, seen as [1, ...] (no stacking copy)
            return nn.functional.embedding(idx[0], self.weight[0], padding_idx=self.padding_idx).unsqueeze(0)
        return torch.stack([nn.functional.embedding(idx[q], self.weight[q], padding_idx=self.padding_idx)
                            for q in range(self.n_level)], dim=0)

    def embed_sum(self, idx):
        return ops.embed_sum(self.weight, idx, padding_idx=self.padding_idx)


class CodecHead(nn.Module):
    """logits[b,n,q,l] = sum_d y[b,n,d] * weight[q,l,d]   (EinMix 'b n d -> b n q l', no bias)."""

    def __init__(self, n_quant: int, n_vocab: int, d_model: int):
        super().__init__()
        bound = (3.0 / d_model) ** 0.5  # einops EinMix default: U(-sqrt(3/fan_in), +)
        self.weight = nn.Parameter(torch.empty(n_quant, n_vocab, d_model).uniform_(-bound, bound))

    def forward(self, y):
        q, l, d = self.weight.shape
        return (y @ self.weight.reshape(q * l, d).t()).view(*y.shape[:-1], q, l)


def topk_sampling(seq, k: int = 1, temp: float = 1.0, generator=None):
    """Top-k / temperature sampling over the last dim of ``seq [rows, vocab]`` -> ``[rows, 1]``
    (reference model/tools.py:38-44).  k == 1 is the greedy pick (device-side arg-max, K6b); k > 1 runs the
    device-side sampler K6c on one uniform number per row drawn from torch's generator (same distribution as
    the reference's softmax + multinomial; torch.multinomial's own random stream is not reproduced)."""
    if k == 1:
        return ops.argmax_rows(seq).unsqueeze(-1)
    u = torch.rand(seq.shape[0], device=seq.device, generator=generator)
    return ops.topk_sample_rows(seq, k, temp, u=u).unsqueeze(-1)


def delay_rvq(code, head_token: int = -2, tail_token: int = -3):
    """Stagger quantizer i by i+1 frames, padding with head/tail tokens: [q,n] -> [q,n+q+1]."""
    q, n = code.shape
    out = torch.empty(q, n + q + 1, dtype=torch.long)
    for i in range(q):
        out[i, :i + 1] = head_token
        out[i, i + 1:i + 1 + n] = code[i]
        out[i, i + 1 + n:] = tail_token
    return out


def undelay_rvq(extended_code):
    """Inverse stagger on [q,b,n+q+1] -> [q,b,n]."""
    q, _, n = extended_code.shape
    return torch.stack([extended_code[i, :, i + 1:i + 1 + n - (q + 1)] for i in range(q)], dim=0)


def sequence_mask(lengths, max_len=None, device=None):
    max_len = int(lengths.max()) if max_len is None else max_len
    return torch.arange(max_len, device=device or lengths.device)[None, :] < lengths.to(device or lengths.device)[:, None]
