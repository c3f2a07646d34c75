"""Stand-in ``fla`` / ``rotary_embedding_torch`` namespaces backed by the CPU oracle.

TEST INFRASTRUCTURE ONLY.  The reference imports its arithmetic from the absent
``fla`` package (model/gla.py:19-23, model/simple_gla.py:16-20) and, at import
time only, ``rotary_embedding_torch`` (model/base_blocks.py:6).  ``install()``
registers module objects under those names so that the reference's in-tree
``model/*.py`` can be imported *in the build container* for golden capture
(tests/golden/make_golden.py).  Semantics follow SURVEY.md Appendix B
[EXT-UNVERIFIED: upstream fla ~Oct 2024].
"""
from __future__ import annotations

import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import gla_oracle as O


class ShortConvolution(nn.Conv1d):
    """fla.modules.ShortConvolution(hidden_size, kernel_size, bias=False, activation='silu')."""

    def __init__(self, hidden_size, kernel_size, bias=False, activation="silu", use_fast_conv1d=True):
        super().__init__(hidden_size, hidden_size, kernel_size, groups=hidden_size, bias=bias,
                         padding=kernel_size - 1)
        self.hidden_size = hidden_size
        self.activation = activation

    def forward(self, x, mask=None, cache=None):
        return O.short_conv(x, self.weight, mask, cache, self.activation, self.bias)

    @property
    def state_size(self):
        return self.hidden_size * self.kernel_size[0]


class FusedRMSNormSwishGate(nn.Module):
    def __init__(self, hidden_size, elementwise_affine=True, eps=1e-5):
        super().__init__()
        self.hidden_size, self.eps = hidden_size, eps
        self.weight = nn.Parameter(torch.ones(hidden_size)) if elementwise_affine else None

    def forward(self, x, o, residual=None, prenorm=False, residual_in_fp32=False):
        return O.rmsnorm_swish_gate(x, o, self.weight, self.eps)


class RMSNorm(nn.Module):
    def __init__(self, hidden_size, elementwise_affine=True, eps=1e-5):
        super().__init__()
        self.hidden_size, self.eps = hidden_size, eps
        self.weight = nn.Parameter(torch.ones(hidden_size)) if elementwise_affine else None

    def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
        if residual is not None:
            x = x + residual
        y = O.rmsnorm(x, self.weight, self.eps)
        return (y, x) if prenorm else y


class Cache:
    """fla.models.utils.Cache: list of per-layer state tuples; first update per layer
    appends, later updates copy_ element-wise into the stored tensors
    (usage: model/gla.py:145,213,303-311,323)."""

    def __init__(self, seen_tokens: int = 0):
        self.states = []
        self._seen_tokens = seen_tokens

    def __getitem__(self, layer_idx):
        return self.states[layer_idx]

    def __iter__(self):
        return iter(self.states)

    def __len__(self):
        return len(self.states)

    def update(self, state, layer_idx, offset=1):
        if isinstance(state, torch.Tensor):
            state = (state,)
        if len(self.states) <= layer_idx:
            self.states.append(tuple(state))
        else:
            for old, new in zip(self.states[layer_idx], state):
                old.copy_(new)
            if layer_idx == len(self.states) - 1:
                self._seen_tokens += offset
        return state

    def get_seq_length(self, layer_idx=0):
        return self._seen_tokens


class SimpleGatedLinearAttention(nn.Module):
    """fla.layers.simple_gla.SimpleGatedLinearAttention [EXT-UNVERIFIED]: scalar gate per head,
    expand_k = expand_v = 1 defaults, output RMSNorm (x) swish gate."""

    def __init__(self, mode="chunk", hidden_size=1024, expand_k=1.0, expand_v=1.0, num_heads=4,
                 use_short_conv=False, conv_size=4, gate_logit_normalizer=16, layer_idx=None, **kw):
        super().__init__()
        self.num_heads = num_heads
        self.key_dim, self.value_dim = int(hidden_size * expand_k), int(hidden_size * expand_v)
        self.head_v_dim = self.value_dim // num_heads
        self.use_short_conv = use_short_conv
        self.gate_logit_normalizer = gate_logit_normalizer
        self.layer_idx = layer_idx
        self.q_proj = nn.Linear(hidden_size, self.key_dim, bias=False)
        self.k_proj = nn.Linear(hidden_size, self.key_dim, bias=False)
        self.v_proj = nn.Linear(hidden_size, self.value_dim, bias=False)
        self.g_proj = nn.Linear(hidden_size, self.value_dim, bias=False)
        self.gk_proj = nn.Linear(hidden_size, num_heads, bias=True)
        if use_short_conv:
            self.q_conv1d = ShortConvolution(self.key_dim, conv_size)
            self.k_conv1d = ShortConvolution(self.key_dim, conv_size)
            self.v_conv1d = ShortConvolution(self.value_dim, conv_size)
        self.g_norm_swish_gate = FusedRMSNormSwishGate(self.head_v_dim)
        self.o_proj = nn.Linear(self.value_dim, hidden_size, bias=False)

    def forward(self, hidden_states, attention_mask=None, past_key_values=None, use_cache=False,
                output_attentions=False, **kw):
        B, T, _ = hidden_states.shape
        H = self.num_heads
        q, k, v = self.q_proj(hidden_states), self.k_proj(hidden_states), self.v_proj(hidden_states)
        if self.use_short_conv:
            q, k, v = self.q_conv1d(q), self.k_conv1d(k), self.v_conv1d(v)
        q, k, v = (x.view(B, T, H, -1).transpose(1, 2) for x in (q, k, v))
        g = F.logsigmoid(self.gk_proj(hidden_states).float()).transpose(1, 2) / self.gate_logit_normalizer
        o, _ = O.simple_gla_recurrent(q, k, v, g)
        o = o.transpose(1, 2)
        gate = self.g_proj(hidden_states).view(B, T, H, -1)
        o = self.g_norm_swish_gate(o, gate).reshape(B, T, -1)
        return self.o_proj(o), None, past_key_values


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def build_modules():
    """Return {module name: module object} for every fla.* / rotary name the reference imports."""
    mods = {}
    mods["fla"] = _mod("fla")
    mods["fla.modules"] = _mod("fla.modules", FusedRMSNormSwishGate=FusedRMSNormSwishGate, RMSNorm=RMSNorm,
                               ShortConvolution=ShortConvolution, FusedCrossEntropyLoss=nn.CrossEntropyLoss)
    mods["fla.modules.activations"] = _mod("fla.modules.activations",
                                           swiglu_linear=lambda x, y, w, b: F.linear(F.silu(x) * y, w, b))
    mods["fla.ops"] = _mod("fla.ops")
    mods["fla.ops.gla"] = _mod("fla.ops.gla", chunk_gla=O.chunk_gla, fused_chunk_gla=O.fused_chunk_gla,
                               fused_recurrent_gla=O.fused_recurrent_gla)
    mods["fla.ops.gla.naive"] = _mod("fla.ops.gla.naive", naive_recurrent_gla=O.naive_recurrent_gla)
    mods["fla.ops.simple_gla"] = _mod("fla.ops.simple_gla", chunk_simple_gla=O.chunk_simple_gla)
    mods["fla.models"] = _mod("fla.models")
    mods["fla.models.utils"] = _mod("fla.models.utils", Cache=Cache)
    mods["fla.models.gla"] = _mod("fla.models.gla")
    mods["fla.models.gla.configuration_gla"] = _mod("fla.models.gla.configuration_gla", GLAConfig=object)
    mods["fla.layers"] = _mod("fla.layers")
    mods["fla.layers.simple_gla"] = _mod("fla.layers.simple_gla",
                                         SimpleGatedLinearAttention=SimpleGatedLinearAttention)

    class _NoRotary:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            raise RuntimeError("rotary stand-in invoked; run the reference with rotary=False")

        rotate_queries_or_keys = __call__

    def _no_apply(*a, **k):
        raise RuntimeError("rotary stand-in invoked; run the reference with rotary=False")

    mods["rotary_embedding_torch"] = _mod("rotary_embedding_torch", RotaryEmbedding=_NoRotary,
                                          apply_rotary_emb=_no_apply)
    return mods


def install():
    """Register the stand-ins in sys.modules (golden capture / reference-side tests only)."""
    for name, m in build_modules().items():
        sys.modules.setdefault(name, m)
