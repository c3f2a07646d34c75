"""Serve the HIP-backed operators under the ``fla.*`` names the reference imports
(/root/reference/model/gla.py:19-23, model/simple_gla.py:16-20), so that the reference's
UNMODIFIED model files run on this package:

    import lina_speech_amd.fla_compat as fc; fc.install()
    from model.gla import AttentiveGLA          # the reference's own module, now on MI355X kernels

``install()`` only registers module objects in ``sys.modules``; it does not touch the
reference tree.  A real ``fla`` installation, if importable, is left alone unless
``override=True``.
"""
from __future__ import annotations

import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import modules, ops


class SimpleGatedLinearAttention(nn.Module):
    """fla.layers.simple_gla.SimpleGatedLinearAttention (scalar log-gate per head)
    on the chunk kernel K2; returns the (o, attentions, past_key_values) triple fla layers return."""

    def __init__(self, mode="chunk", hidden_size=1024, expand_k=1.0, expand_v=1.0, num_heads=4,
                 use_short_conv=False, conv_size=4, gate_logit_normalizer=16, layer_idx=None, **kw):
        super().__init__()
        self.num_heads, self.layer_idx = num_heads, layer_idx
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
        q, k, v = self.q_proj(hidden_states), self.k_proj(hidden_states), self.v_proj(hidden_states)
        if self.use_short_conv:
            q, k, v = self.q_conv1d(q), self.k_conv1d(k), self.v_conv1d(v)
        q, k, v = (x.view(B, T, H, -1).transpose(1, 2) for x in (q, k, v))
        g = (F.logsigmoid(self.gk_proj(hidden_states).float()) / self.gate_logit_normalizer).transpose(1, 2)
        o, _ = ops.chunk_simple_gla(q, k, v, g.to(q.dtype))
        gate = self.g_proj(hidden_states).view(B, T, H, -1)
        o = self.g_norm_swish_gate(o.transpose(1, 2), gate).reshape(B, T, -1)
        return self.o_proj(o), None, past_key_values


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__dict__["__lina_speech_amd__"] = True
    return m


def build_modules():
    return {
        "fla": _mod("fla"),
        "fla.modules": _mod("fla.modules", FusedRMSNormSwishGate=modules.FusedRMSNormSwishGate,
                            RMSNorm=modules.RMSNorm, ShortConvolution=modules.ShortConvolution,
                            FusedCrossEntropyLoss=nn.CrossEntropyLoss),
        "fla.modules.activations": _mod("fla.modules.activations",
                                        swiglu_linear=lambda x, y, w, b: F.linear(F.silu(x) * y, w, b)),
        "fla.ops": _mod("fla.ops"),
        "fla.ops.gla": _mod("fla.ops.gla", chunk_gla=ops.chunk_gla, fused_chunk_gla=ops.fused_chunk_gla,
                            fused_recurrent_gla=ops.fused_recurrent_gla),
        "fla.ops.gla.naive": _mod("fla.ops.gla.naive", naive_recurrent_gla=ops.naive_recurrent_gla),
        "fla.ops.simple_gla": _mod("fla.ops.simple_gla", chunk_simple_gla=ops.chunk_simple_gla),
        "fla.models": _mod("fla.models"),
        "fla.models.utils": _mod("fla.models.utils", Cache=modules.Cache),
        "fla.models.gla": _mod("fla.models.gla"),
        "fla.models.gla.configuration_gla": _mod("fla.models.gla.configuration_gla", GLAConfig=object),
        "fla.layers": _mod("fla.layers"),
        "fla.layers.simple_gla": _mod("fla.layers.simple_gla",
                                      SimpleGatedLinearAttention=SimpleGatedLinearAttention),
    }


def install(override: bool = False) -> None:
    for name, m in build_modules().items():
        if override or name not in sys.modules:
            sys.modules[name] = m
