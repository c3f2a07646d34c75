"""AttentiveGLA: encoder GLA stack -> blind cross-attention -> decoder GLA stack
(reference model/gla.py:252-365, abstract base model/attentive_rnn.py:6-17).

Same constructor, ``forward`` / ``init_state`` / ``step`` / ``to_mode`` contracts and
layer numbering (decoder = n_layer+i, pos_net = 2*n_layer).
"""
from __future__ import annotations

import os
from abc import abstractmethod
from typing import Optional

import torch
import torch.nn as nn
import torch.utils.checkpoint

from .blind_attention import BlindCrossAttention
from .blocks import MixingBlock, SwiGLU
from .mixer import GatedLinearAttention
from .modules import Cache


def _maybe_grad_ckpt(blk):
    """Activation checkpointing switch of the reference (model/gla.py:26-33,290-291,297-298): with ``GRAD_CKPT`` in
    the environment every encoder / decoder block is recomputed in backward (non-reentrant checkpoint) while the
    module is training.  Read at call time (the reference reads it at import time)."""
    if "GRAD_CKPT" not in os.environ:
        return blk
    return lambda *a, **kw: torch.utils.checkpoint.checkpoint(blk, *a, **kw, use_reentrant=False)


class AttentiveRNN(nn.Module):
    @abstractmethod
    def forward(self, x, ctx, x_mask, ctx_mask):
        ...

    @abstractmethod
    def init_state(self):
        ...

    @abstractmethod
    def step(self, x, ctx, crossatt_mask):
        ...


class AttentiveGLA(AttentiveRNN):
    def __init__(self, d_model: int, n_layer: int, heads: int, dropout_att: float = 0.0, dropout: float = 0.0,
                 d_blind: Optional[int] = None, blind: bool = False, cross_att_pp: bool = False,
                 rotary: bool = False, use_short_conv: bool = False, expand_k: float = 1.0,
                 expand_v: float = 2.0, pos_type: str = "sinusoidal"):
        super().__init__()
        if not blind or cross_att_pp:
            raise NotImplementedError("only the blind cross-attention ('convblind') stacking is on the path "
                                      "(SURVEY 2, #3)")

        def block(d, h, idx):
            return MixingBlock(lambda: GatedLinearAttention(hidden_size=d, num_heads=h, use_short_conv=use_short_conv,
                                                            expand_k=expand_k, expand_v=expand_v, layer_idx=idx),
                               lambda: SwiGLU(d), lambda: nn.LayerNorm(d), dropout=dropout)

        self.n_layer = n_layer
        self.encoder = nn.ModuleList([block(d_model, heads, i) for i in range(n_layer)])
        self.decoder = nn.ModuleList([block(d_model, heads, n_layer + i) for i in range(n_layer)])
        d_blind = d_model if d_blind is None else d_blind
        self.cross_att = BlindCrossAttention(d_model, d_model, d_model, 1, block(d_blind, heads, 2 * n_layer),
                                             dropout_att, pos_dim=d_blind, rotary=rotary, pos_type=pos_type)

    # teacher-forced / prefill
    def forward(self, x, ctx, mask=None, pos=None, reset_mask=None, attention_only=None, forced_attention=None,
                init_state=None, crossatt_pos=None):
        kw = dict(use_cache=init_state is not None, past_key_values=init_state)
        # the blocks are chained on (stream, pending branch): a block's last residual add rides in the next block's first
        # norm pass instead of a pass of its own (MixingBlock.forward); the values are those of the plain loop
        pend = None
        for blk in self.encoder:
            x, pend = (_maybe_grad_ckpt(blk) if self.training else blk)(x, _pending=pend, _defer=True, **kw)
        if pend is not None:
            x = x + pend
        v, att = self.cross_att(x, ctx, mask=mask, reset_mask=reset_mask, pos=crossatt_pos)
        pend = v                                                # x + v: added by the first decoder block's norm
        for blk in self.decoder:
            x, pend = (_maybe_grad_ckpt(blk) if self.training else blk)(x, _pending=pend, _defer=True, **kw)
        if pend is not None:
            x = x + pend
        return x, att

    def init_state(self, max_seqlen: int = 1000, batch_size: int = 16) -> Cache:
        cache = Cache()
        blocks = list(self.encoder) + list(self.decoder)
        for i, blk in enumerate(blocks):
            cache.update(blk.tmix.init_state(batch_size), i, offset=0)
        # the pos_net slot reuses the last decoder block's shapes (reference gla.py:310-311)
        cache.update(blocks[-1].tmix.init_state(batch_size), len(blocks), offset=0)
        return cache

    def get_state_from_params(self, params, batch_size, scale=0.02):
        cache = self.init_state(batch_size=batch_size)
        for i, x in enumerate(params):
            if len(x) == 2:   # rank-r factorisation: k [1,r,h,k,1], v [1,r,h,1,v]
                state = (x[0] * x[1]).sum(1) * scale
            else:
                state = x[0]
            state = state.expand(batch_size, *state.shape[1:]).clone()
            cache.states[i] = cache.states[i][:-1] + (state,)
        return cache

    def get_init_state_tuning_params(self, lora: Optional[int] = None, scale: float = 0.02, device=None):
        params = []
        for blk in list(self.encoder) + list(self.decoder):
            m = blk.tmix
            if lora is not None:
                params.append((nn.Parameter(torch.randn(1, lora, m.num_heads, m.head_qk_dim, 1, device=device)),
                               nn.Parameter(torch.randn(1, lora, m.num_heads, 1, m.head_v_dim, device=device) * scale)))
            else:
                params.append(nn.Parameter(torch.randn(1, m.num_heads, m.head_qk_dim, m.head_v_dim,
                                                       device=device) * scale))
        return params

    def to_mode(self, mode: str):
        for blk in list(self.encoder) + list(self.decoder):
            blk.tmix.mode = mode
        self.cross_att.pos_net.mode = mode  # attribute on the block, exactly as the reference sets it (gla.py:333)

    # single-token decode
    def step(self, y_embd, x_enc, time_step, cache, prepared=None):
        kw = dict(past_key_values=cache, use_cache=True)
        for blk in self.encoder:
            y_embd = blk(y_embd, **kw)
        v, att = self.cross_att(y_embd, x_enc, time_step=time_step, prepared=prepared, **kw)
        y_embd = y_embd + v
        for blk in self.decoder:
            y_embd = blk(y_embd, **kw)
        return y_embd, att, cache
