"""fla-shaped building blocks served by the HIP kernels: ShortConvolution,
FusedRMSNormSwishGate, RMSNorm and the per-layer state Cache.

These are the classes the reference instantiates from ``fla.modules`` /
``fla.models.utils`` (/root/reference/model/gla.py:19,23,104-115,303); parameter
names and shapes are kept (``weight [D,1,W]``, ``weight [hidden]``) so reference
checkpoints load unchanged.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops


class ShortConvolution(nn.Module):
    """Depthwise causal conv (kernel W, no padding leak) + activation; K3 (prefill) / K4 (step).

    ``forward(x[B,T,D], mask[B,T]|None, cache[B,D,W]|None) -> [B,T,D]``; the cache is
    updated IN PLACE (reference model/gla.py:149,161-163).
    """

    def __init__(self, hidden_size: int, kernel_size: int, bias: bool = False,
                 activation: Optional[str] = "silu", use_fast_conv1d: bool = True):
        super().__init__()
        self.hidden_size = hidden_size
        self.kernel_size = (kernel_size,)
        self.activation = activation
        self.weight = nn.Parameter(torch.empty(hidden_size, 1, kernel_size))
        self.bias = nn.Parameter(torch.empty(hidden_size)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # nn.Conv1d(groups=D) default: kaiming_uniform(a=sqrt 5) over fan_in = kernel_size
        bound = (1.0 / self.kernel_size[0]) ** 0.5
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)

    @property
    def state_size(self) -> int:
        return self.hidden_size * self.kernel_size[0]

    def forward(self, x, mask=None, cache=None, *, grad_slab=None):
        return ops.short_conv(x, self.weight, self.bias, mask, cache, self.activation, grad_slab=grad_slab)

    def extra_repr(self):
        return f"{self.hidden_size}, kernel_size={self.kernel_size[0]}, activation={self.activation}"


class FusedRMSNormSwishGate(nn.Module):
    """y = rmsnorm(x) * weight * g * sigmoid(g) over the last dim -- K5 (reference model/gla.py:111,219)."""

    def __init__(self, hidden_size: int, elementwise_affine: bool = True, eps: float = 1e-5):
        super().__init__()
        self.hidden_size, self.eps = hidden_size, eps
        self.weight = nn.Parameter(torch.ones(hidden_size)) if elementwise_affine else None

    def forward(self, x, o, residual=None, prenorm=False, residual_in_fp32=False, *, grad_slab=None):
        if residual is not None or prenorm:
            raise NotImplementedError("residual/prenorm form is not used on the Lina path")
        return ops.rmsnorm_swish_gate(x, o, self.weight, self.eps, grad_slab=grad_slab)


class RMSNorm(nn.Module):
    """fla.modules.RMSNorm incl. the (x, residual, prenorm=True) -> (y, residual) call form
    (reference model/simple_gla.py:109)."""

    def __init__(self, hidden_size: int, elementwise_affine: bool = True, eps: float = 1e-5):
        super().__init__()
        self.hidden_size, self.eps = hidden_size, eps
        self.weight = nn.Parameter(torch.ones(hidden_size)) if elementwise_affine else None

    def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
        if residual is not None:
            x = x + residual
        y = ops.rmsnorm(x, self.weight, self.eps)
        return (y, x) if prenorm else y


class Cache:
    """Per-layer recurrent/conv state store with fla.models.utils.Cache semantics: the first
    ``update`` for a layer stores the tuple, later ones ``copy_`` into the stored tensors (a
    tensor that already IS the stored one -- the in-place HIP decode update -- is skipped).
    Usage in the reference: model/gla.py:145,213,303-311,323."""

    def __init__(self, seen_tokens: int = 0):
        self.states = []
        self._seen_tokens = seen_tokens

    def __getitem__(self, layer_idx: int):
        if layer_idx >= len(self.states):
            raise KeyError(f"cache holds {len(self.states)} layers, asked for {layer_idx}")
        return self.states[layer_idx]

    def __iter__(self):
        return iter(self.states)

    def __len__(self):
        return len(self.states)

    def update(self, state, layer_idx: int, offset: int = 1):
        if isinstance(state, torch.Tensor):
            state = (state,)
        if len(self.states) <= layer_idx:
            self.states.append(tuple(state))
        else:
            for kept, new in zip(self.states[layer_idx], state):
                if kept is not new and kept.data_ptr() != new.data_ptr():
                    kept.copy_(new)
            if layer_idx == len(self.states) - 1:
                self._seen_tokens += offset
        return state

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self._seen_tokens

    def to_legacy_cache(self):
        return tuple(self.states)
