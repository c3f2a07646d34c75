"""Pinned synthetic configurations (the reference's training yaml is not in its repo;
SURVEY.md fact 6 / Appendix C.1 infer the shape from the parameter count)."""
from __future__ import annotations

import torch

from .attentive import AttentiveGLA
from .blocks import TextEncoder
from .lina_model import LinaModel


def l169(heads: int = 4, expand_v: float = 1.0, txt_layers: int = 4, n_layer: int = 6) -> LinaModel:
    """"169M d1024 x l12": d_model 1024, 6 encoder + 6 decoder GLA blocks + the pos_net GLA block of the
    blind cross-attention, key_dim = value_dim = 1024, H = 4 (Dk = Dv = 256), conv W = 4, codebook 4096 + 3
    specials, 1 quantizer, text vocab 256, 4-layer text encoder  ->  166.7 M parameters.  (``n_layer`` / ``txt_layers`` smaller:
    a slice of the same width for tests.)"""
    rnn = AttentiveGLA(d_model=1024, n_layer=n_layer, heads=heads, blind=True, use_short_conv=True, expand_k=1.0,
                       expand_v=expand_v, pos_type="convolutional")
    txt = TextEncoder(1024, 4, n_layers=txt_layers, dropout=0.0, rotary=False)
    return LinaModel(rnn, d_model=1024, n_quant=1, n_codebook=4096, n_special_token_in=3, n_special_token_out=3,
                     n_txt_vocab=256, txt_encoder=txt)


def tiny(d: int = 64, heads: int = 1, n_layer: int = 1, n_codebook: int = 253) -> LinaModel:
    rnn = AttentiveGLA(d_model=d, n_layer=n_layer, heads=heads, blind=True, use_short_conv=True, expand_k=1.0,
                       expand_v=1.0, pos_type="convolutional")
    txt = TextEncoder(d, heads, n_layers=1, dropout=0.0, rotary=False)
    return LinaModel(rnn, d_model=d, n_quant=1, n_codebook=n_codebook, n_special_token_in=3, n_special_token_out=3,
                     n_txt_vocab=256, txt_encoder=txt)


def n_params(model: torch.nn.Module) -> int:
    return sum(p.numel() for p in model.parameters())
