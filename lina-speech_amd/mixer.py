"""GatedLinearAttention token mixer -- module-level drop-in for the reference's
``model/gla.py:44-247`` (same constructor arguments, forward signature, ``mode`` switch,
``init_state`` layout and state-dict keys), with the fla operators replaced by the HIP
kernels in ``ops`` (K1 recurrent step, K2 chunk scan, K3/K4 conv, K5 norm-gate).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .modules import Cache, FusedRMSNormSwishGate, RMSNorm, ShortConvolution

_MODES = ("chunk", "fused_recurrent", "fused_chunk", "naive")
_ACT = {"swish": F.silu, "silu": F.silu, "gelu": F.gelu, "relu": F.relu, "sigmoid": torch.sigmoid,
        "tanh": torch.tanh}


class GatedLinearAttention(nn.Module):
    def __init__(self, mode: str = "fused_chunk", hidden_size: int = 1024, expand_k: float = 1.0,
                 expand_v: float = 2.0, num_heads: int = 4, use_short_conv: bool = False, conv_size: int = 4,
                 conv_bias: bool = False, share_conv_kernel: bool = False, gate_fn: str = "swish",
                 layernorm_eps: float = 1e-5, gate_logit_normalizer: int = 16, gate_low_rank_dim: int = 16,
                 clamp_min: Optional[float] = None, fuse_norm: bool = True, layer_idx: Optional[int] = None,
                 **kwargs):
        super().__init__()
        if mode not in _MODES:
            raise AssertionError(f"Not suppoerted mode `{mode}`.")
        self.mode, self.hidden_size, self.num_heads = mode, hidden_size, num_heads
        self.expand_k, self.expand_v = expand_k, expand_v
        self.use_short_conv, self.conv_size, self.conv_bias = use_short_conv, conv_size, conv_bias
        self.share_conv_kernel = share_conv_kernel
        self.key_dim, self.value_dim = int(hidden_size * expand_k), int(hidden_size * expand_v)
        self.clamp_min, self.layer_idx = clamp_min, layer_idx
        self.gate_logit_normalizer = gate_logit_normalizer
        assert self.key_dim % num_heads == 0, f"key dim must be divisible by num_heads of {num_heads}"
        assert self.value_dim % num_heads == 0, f"value dim must be divisible by num_heads of {num_heads}"
        self.head_qk_dim, self.head_v_dim = self.key_dim // num_heads, self.value_dim // num_heads
        self.state = None  # read by the 'inference' / 'naive' / 'init_state_tuning' modes (gla.py:190,197,200)

        self.q_proj = nn.Linear(hidden_size, self.key_dim, bias=False)
        self.k_proj = nn.Linear(hidden_size, self.key_dim, bias=False)
        self.v_proj = nn.Linear(hidden_size, self.value_dim, bias=False)
        self.g_proj = nn.Linear(hidden_size, self.value_dim, bias=False)
        self.gk_proj = nn.Sequential(nn.Linear(hidden_size, gate_low_rank_dim, bias=False),
                                     nn.Linear(gate_low_rank_dim, self.key_dim, bias=True))
        self.o_proj = nn.Linear(self.value_dim, hidden_size, bias=False)
        if use_short_conv:
            if share_conv_kernel:
                self.h_conv1d = ShortConvolution(hidden_size, conv_size, bias=conv_bias, activation="silu")
            else:
                self.q_conv1d = ShortConvolution(self.key_dim, conv_size, bias=conv_bias, activation="silu")
                self.k_conv1d = ShortConvolution(self.key_dim, conv_size, bias=conv_bias, activation="silu")
                self.v_conv1d = ShortConvolution(self.value_dim, conv_size, bias=conv_bias, activation="silu")
        self.fuse_norm_and_gate = gate_fn == "swish" and fuse_norm
        if self.fuse_norm_and_gate:
            self.g_norm_swish_gate = FusedRMSNormSwishGate(self.head_v_dim, eps=layernorm_eps)
        else:
            self.g_norm = RMSNorm(self.head_v_dim, eps=layernorm_eps)
            self.gate_fn = _ACT[gate_fn]
        for m in self.modules():  # reference initialiser: xavier-uniform, gain 2^-2.5 (gla.py:122-129)
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight, gain=2 ** -2.5)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    # ------------------------------------------------------------------ helpers
    def _heads(self, x):
        B, T, _ = x.shape
        return x.view(B, T, self.num_heads, -1).transpose(1, 2)  # 'b l (h d) -> b h l d' as a view

    def _gates(self, hidden_states, reset_mask, reset_val, low_rank=None):
        if low_rank is None:
            low_rank = self.gk_proj[0](hidden_states)
        # K12b: the 16-wide second projection, logsigmoid / normalizer (+ clamp) in one pass each way (the head split is a
        # view of the result)
        gk = self._heads(ops.gate_lowrank(low_rank, self.gk_proj[1].weight, self.gk_proj[1].bias,
                                          self.gate_logit_normalizer, self.clamp_min))
        if reset_mask is not None:
            gk = gk.masked_fill(reset_mask.unsqueeze(1).unsqueeze(3), reset_val)
        return gk

    def _recurrence(self, mode, q, k, v, gk, recurrent_state, use_cache):
        # A cached single-token call is the decode step: always the in-place recurrent kernel K1,
        # whatever chunk mode the module is in (the reference dispatches 'fused_chunk' with T = 1
        # here and notes it is slow, model/gla.py:142; the arithmetic is the same recurrence).
        if (q.shape[2] == 1 and use_cache and recurrent_state is not None and not self.training
                and mode in ("fused_recurrent", "fused_chunk", "chunk")):
            return ops.fused_recurrent_gla(q, k, v, gk, initial_state=recurrent_state, output_final_state=True,
                                           inplace_state=recurrent_state.dtype == torch.float32)
        if mode == "fused_recurrent":
            inplace = (recurrent_state is not None and not self.training and recurrent_state.dtype == torch.float32)
            return ops.fused_recurrent_gla(q, k, v, gk, initial_state=recurrent_state,
                                           output_final_state=use_cache, inplace_state=inplace and use_cache)
        if mode == "inference":
            o, st = ops.fused_recurrent_gla(q, k, v, gk, initial_state=self.state, output_final_state=True)
            self.state = st
            return o, st
        if mode == "fused_chunk":
            return ops.fused_chunk_gla(q, k, v, gk, initial_state=recurrent_state, output_final_state=use_cache)
        if mode == "chunk":
            return ops.chunk_gla(q, k, v, gk, initial_state=recurrent_state, output_final_state=use_cache)
        if mode == "naive":
            return ops.naive_recurrent_gla(q, k, v, gk, initial_state=self.state, output_final_state=use_cache)
        if mode == "init_state_tuning":
            init = self.state.expand(q.shape[0], *self.state.shape[1:]).contiguous()
            return ops.fused_recurrent_gla(q, k, v, gk, initial_state=init, output_final_state=True)
        raise NotImplementedError(f"Not supported mode `{mode}`.")

    # ------------------------------------------------------------------ forward
    def forward(self, hidden_states: torch.Tensor, reset_mask: Optional[torch.Tensor] = None,
                attention_mask: Optional[torch.Tensor] = None, reset_val: float = -20,
                past_key_values: Optional[Cache] = None, use_cache: Optional[bool] = False,
                output_attentions: Optional[bool] = False, **kwargs) -> torch.Tensor:
        last_state = past_key_values[self.layer_idx] if use_cache else None
        conv_states: Tuple = ()
        if hidden_states.is_cuda and torch.is_autocast_enabled():
            # under autocast every one of the five projections below would cast this (fp32 LayerNorm output) tensor
            # to the autocast dtype on its own: do it once (same values, 4 fewer passes over [B,T,d])
            hidden_states = hidden_states.to(torch.get_autocast_dtype("cuda"))
        g_pre = lr_pre = slab = None                            # outputs of the fused projection, when it ran
        if self.use_short_conv and self.share_conv_kernel:
            conv_states = (last_state[0] if use_cache else None,)
            hidden_states = self.h_conv1d(hidden_states, attention_mask, conv_states[0])
            q, k, v = self.q_proj(hidden_states), self.k_proj(hidden_states), self.v_proj(hidden_states)
        else:
            if hidden_states.shape[1] > 1 and self.q_proj.bias is None and self.g_proj.bias is None:
                # prefill / training: the five projections of the block input as ONE GEMM over the stacked weights
                # (same columns, one pass over the activations, a 5136-wide GEMM instead of five narrow ones);
                # the outputs are strided views of its result
                parts = [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.g_proj.weight,
                         self.gk_proj[0].weight]
                sizes = [self.key_dim, self.key_dim, self.value_dim, self.value_dim, self.gk_proj[0].weight.shape[0]]
                # rows of the stacked output padded to a multiple of 128 bytes (L169: 4112 -> 4160 columns, zero weight rows):
                # with 8224-byte rows every 512-byte wave access of the slices' consumers (conv, norm-gate, gate kernels: they
                # read q | k | v | g | lr in place) straddles one more cache line -- the fused convolution ran 89 us forward /
                # 171 us backward on such rows against 76 / 150 on aligned ones (profiles/r06_conv_ab.txt); the GEMM works on
                # 17 column tiles of 256 either way
                pad = (-sum(sizes)) % 64
                if pad:
                    sizes.append(pad)
                # the stacked GEMM-dtype operand in ONE pass over the master weights (K16; zero pad rows included) and the
                # blocks' gradients as row ranges of dW -- no torch.cat of fp32 weights, no cast, no split in backward
                # (training: the consumers of the slices write their input gradients into ONE slab -- no concat pass)
                (q, k, v, g_pre, lr_pre, *_), slab = ops.split_slab(ops.stacked_linear(hidden_states, parts, pad), sizes)
            else:
                q, k, v = self.q_proj(hidden_states), self.k_proj(hidden_states), self.v_proj(hidden_states)
            if self.use_short_conv:
                conv_states = tuple(last_state[i] if use_cache else None for i in range(3))
                gs = [None] * 3 if slab is None else [(slab, 0), (slab, 1), (slab, 2)]
                fused = None
                if slab is not None and not use_cache:
                    # training: the three depthwise convolutions over adjacent column slices of the projection in ONE launch
                    # each way (autograd.short_conv3); same values, 6 KB contiguous per token instead of 3 x 2 KB
                    cs = (self.q_conv1d, self.k_conv1d, self.v_conv1d)
                    if len({c.activation for c in cs}) == 1:
                        fused = ops.short_conv3((q, k, v), [c.weight for c in cs], [c.bias for c in cs], attention_mask,
                                                cs[0].activation, grad_slab=(slab, 0))
                if fused is not None:
                    q, k, v = fused
                else:
                    q = self.q_conv1d(q, attention_mask, conv_states[0], grad_slab=gs[0])
                    k = self.k_conv1d(k, attention_mask, conv_states[1], grad_slab=gs[1])
                    v = self.v_conv1d(v, attention_mask, conv_states[2], grad_slab=gs[2])
        if attention_mask is not None:  # left padding
            v = v * attention_mask.unsqueeze(-1).to(v.dtype)
        q, k, v = self._heads(q), self._heads(k), self._heads(v)
        gk = self._gates(hidden_states, reset_mask, reset_val, lr_pre)

        recurrent_state = last_state[-1] if use_cache else None
        o, recurrent_state = self._recurrence(self.mode, q, k, v, gk, recurrent_state, use_cache)

        if past_key_values is not None and not self.training:
            past_key_values.update(conv_states + (recurrent_state,), self.layer_idx, q.shape[2])

        B, H, T, Dv = o.shape
        o = o.transpose(1, 2)                                   # [B,T,H,Dv] (contiguous by construction)
        g = self.g_proj(hidden_states) if g_pre is None else g_pre
        if self.fuse_norm_and_gate:
            o = self.g_norm_swish_gate(o, g.view(B, T, H, Dv),
                                       grad_slab=None if slab is None else (slab, 3)).reshape(B, T, H * Dv)
        else:
            o = self.g_norm(o).reshape(B, T, H * Dv) * self.gate_fn(g)
        return ops.linear(o, self.o_proj.weight, self.o_proj.bias)

    # ------------------------------------------------------------------ state
    def init_state(self, batch_size: int) -> Tuple[torch.Tensor, ...]:
        """(conv_q[B,Kd,W], conv_k[B,Kd,W], conv_v[B,Vd,W], S[B,H,Dk,Dv]) -- layout of reference
        model/gla.py:229-240, except that S is always fp32 (the reference uses the model dtype)."""
        p = next(self.parameters())
        st: Tuple[torch.Tensor, ...] = ()
        if self.use_short_conv:
            if self.share_conv_kernel:
                st += (p.new_zeros(batch_size, self.hidden_size, self.conv_size),)
            else:
                st += (p.new_zeros(batch_size, self.key_dim, self.conv_size),
                       p.new_zeros(batch_size, self.key_dim, self.conv_size),
                       p.new_zeros(batch_size, self.value_dim, self.conv_size))
        st += (torch.zeros(batch_size, self.num_heads, self.head_qk_dim, self.head_v_dim, dtype=torch.float32,
                           device=p.device),)
        return st

    def state_size(self, **kwargs) -> int:
        n = self.key_dim * self.head_v_dim
        for m in self.children():
            if isinstance(m, ShortConvolution):
                n += m.state_size
        return n
