"""Config 1 of BASELINE.json: the scalar-gate ("simple") GLA stack.

``SimpleGatedLinearAttention`` stands where the reference imports ``fla.layers.simple_gla.SimpleGatedLinearAttention``
(model/simple_gla.py:16,135; an external layer: scalar log-gate per head, expand_k = expand_v = 1, output
RMSNorm (x) swish gate -- SURVEY A.7) and runs on the chunk kernel K2 with the gate broadcast over Dk;
``AttentiveSimpleGLA`` mirrors the reference's wrapper (model/simple_gla.py:116-165): same constructor, layer
numbering (decoder = n_layer + 1 + i, pos_net = n_layer) and ``forward(x, ctx, mask, pos, reset_mask, ...)``.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import modules, ops
from .attentive import AttentiveRNN, _maybe_grad_ckpt
from .blind_attention import BlindCrossAttention
from .blocks import MixingBlock, SwiGLU


class SimpleGatedLinearAttention(nn.Module):
    """Returns the (o, attentions, past_key_values) triple fla layers return."""

    def __init__(self, mode="chunk", hidden_size=1024, expand_k=1.0, expand_v=1.0, num_heads=4,
                 use_short_conv=False, conv_size=4, gate_logit_normalizer=16, layer_idx=None, **kw):
        super().__init__()
        self.mode, self.num_heads, self.layer_idx = mode, num_heads, layer_idx
        self.key_dim, self.value_dim = int(hidden_size * expand_k), int(hidden_size * expand_v)
        self.head_v_dim = self.value_dim // num_heads
        self.use_short_conv, self.gate_logit_normalizer = use_short_conv, gate_logit_normalizer
        self.q_proj = nn.Linear(hidden_size, self.key_dim, bias=False)
        self.k_proj = nn.Linear(hidden_size, self.key_dim, bias=False)
        self.v_proj = nn.Linear(hidden_size, self.value_dim, bias=False)
        self.g_proj = nn.Linear(hidden_size, self.value_dim, bias=False)
        self.gk_proj = nn.Linear(hidden_size, num_heads, bias=True)
        if use_short_conv:
            self.q_conv1d = modules.ShortConvolution(self.key_dim, conv_size)
            self.k_conv1d = modules.ShortConvolution(self.key_dim, conv_size)
            self.v_conv1d = modules.ShortConvolution(self.value_dim, conv_size)
        self.g_norm_swish_gate = modules.FusedRMSNormSwishGate(self.head_v_dim)
        self.o_proj = nn.Linear(self.value_dim, hidden_size, bias=False)

    def forward(self, hidden_states, attention_mask=None, past_key_values=None, use_cache=False,
                output_attentions=False, **kw):
        B, T, _ = hidden_states.shape
        H = self.num_heads
        # one pass over the activations for the four wide projections and the per-head gate logits
        w = torch.cat([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.g_proj.weight,
                       self.gk_proj.weight], dim=0)
        q, k, v, gate, glog = F.linear(hidden_states, w).split(
            [self.key_dim, self.key_dim, self.value_dim, self.value_dim, H], dim=-1)
        if self.use_short_conv:
            q, k, v = self.q_conv1d(q), self.k_conv1d(k), self.v_conv1d(v)
        heads = lambda t: t.view(B, T, H, -1).transpose(1, 2)
        g = (F.logsigmoid((glog + self.gk_proj.bias).float()) / self.gate_logit_normalizer).transpose(1, 2)
        if q.dtype == torch.bfloat16:
            g = g.to(q.dtype)      # the bf16 full-head / segment-parallel K2 and K2b take a bf16 gate (as mixer.py passes it)
        o, _ = ops.chunk_simple_gla(heads(q), heads(k), heads(v), g)
        o = self.g_norm_swish_gate(o.transpose(1, 2), gate.view(B, T, H, -1)).reshape(B, T, -1)
        return self.o_proj(o), None, past_key_values


class AttentiveSimpleGLA(AttentiveRNN):
    def __init__(self, d_model: int, n_layer: int, heads: int, dropout_att: float = 0.0,
                 d_blind: Optional[int] = None, blind: bool = False, cross_att_pp: bool = False, rotary: bool = False,
                 use_short_conv: bool = False, pos_type: str = "sinusoidal", dropout: float = 0.0):
        super().__init__()
        if not blind or cross_att_pp:
            raise NotImplementedError("only the blind cross-attention stacking is on the path (SURVEY 2, #3)")

        def block(d, h, idx):
            return MixingBlock(lambda: SimpleGatedLinearAttention(hidden_size=d, num_heads=h,
                                                                  use_short_conv=use_short_conv, layer_idx=idx),
                               lambda: SwiGLU(d), lambda: nn.LayerNorm(d), dropout=dropout)

        self.encoder = nn.ModuleList([block(d_model, heads, i) for i in range(n_layer)])
        self.decoder = nn.ModuleList([block(d_model, heads, i + n_layer + 1) for i in range(n_layer)])
        d_blind = d_model if d_blind is None else d_blind
        self.cross_att = BlindCrossAttention(d_model, d_model, d_model, 1, block(d_blind, heads, n_layer), dropout_att,
                                             pos_dim=d_blind, rotary=rotary, pos_type=pos_type)

    def forward(self, x, ctx, mask=None, pos=None, reset_mask=None, forced_attention=None, attention_only=None):
        for blk in self.encoder:
            x = (_maybe_grad_ckpt(blk) if self.training else blk)(x)
        v, att = self.cross_att(x, ctx, mask=mask, pos=pos, reset_mask=reset_mask)
        x = x + v
        for blk in self.decoder:
            x = (_maybe_grad_ckpt(blk) if self.training else blk)(x)
        return x, att

    def init_state(self, max_seqlen=1000, state=None):     # the reference leaves these unimplemented (:167-171)
        pass

    def get_state(self):
        pass

    def step(self, y_embd, x_enc, time_step):
        raise NotImplementedError("the reference's AttentiveSimpleGLA.step unpacks a block triple that MixingBlock "
                                  "does not return (model/simple_gla.py:173-180): config 1 is forward-only")
