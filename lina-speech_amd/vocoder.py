"""Codes -> 24 kHz waveform: the WavTokenizer decoder that consumes the generation path's output
(reference 3rdparty/decoder/pretrained.py:193-239, models.py:152-235, modules.py:8-82,
heads.py:24-67, spectral_ops.py:7-75).  Same sub-module names and parameter shapes as the reference
(``backbone.embed / pos_net.{0..5} / norm / convnext.{i}.{dwconv,norm,pwconv1,pwconv2,gamma} / final_layer_norm``,
``head.out``, ``head.istft.window``), so a WavTokenizer checkpoint's ``backbone.*`` / ``head.*`` entries load as-is.

MI355X-first: activations stay channels-last ``[B, L, C]`` (the reference transposes between ``[B,C,L]`` convolutions
and ``[B,L,C]`` norms / linears); the depthwise conv + (Ada)LayerNorm of every ConvNeXt block is ONE HIP kernel (K8);
the ISTFT's windowing, overlap-add, envelope division and trimming are ONE gather kernel (K9) after rocFFT's irfft;
the 1x1 / k=3 convolutions, GroupNorm, the single attention block and the linears are library calls through torch.
Inference only.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class AdaLayerNorm(nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps, self.dim = eps, embedding_dim
        self.scale = nn.Embedding(num_embeddings, embedding_dim)
        self.shift = nn.Embedding(num_embeddings, embedding_dim)
        nn.init.ones_(self.scale.weight)
        nn.init.zeros_(self.shift.weight)

    def rows(self, cond_embedding_id):
        return self.scale(cond_embedding_id), self.shift(cond_embedding_id)

    def forward(self, x, cond_embedding_id):
        scale, shift = self.rows(cond_embedding_id)
        return F.layer_norm(x, (self.dim,), eps=self.eps) * scale + shift


class ConvNeXtBlock(nn.Module):
    """x [B,L,C] -> x + gamma * pwconv2(gelu(pwconv1(norm(dwconv(x)))))   (reference modules.py:8-60)."""

    def __init__(self, dim: int, intermediate_dim: int, layer_scale_init_value: float,
                 adanorm_num_embeddings: Optional[int] = None):
        super().__init__()
        self.dwconv = nn.Conv1d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.adanorm = adanorm_num_embeddings is not None
        self.norm = AdaLayerNorm(adanorm_num_embeddings, dim, eps=1e-6) if self.adanorm else nn.LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, intermediate_dim)
        self.pwconv2 = nn.Linear(intermediate_dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim)) if layer_scale_init_value > 0 else None

    def forward(self, x, cond_embedding_id=None):
        if self.adanorm:
            scale, shift = self.norm.rows(cond_embedding_id)             # [B,C] (or [1,C])
        else:
            scale, shift = self.norm.weight, self.norm.bias
        h = ops.dwconv7_ln(x, self.dwconv.weight, self.dwconv.bias, scale, shift, self.norm.eps)   # K8
        h = self.pwconv2(F.gelu(self.pwconv1(h)))
        return torch.addcmul(x, h, self.gamma) if self.gamma is not None else x + h


def _group_norm(c):
    return nn.GroupNorm(num_groups=32, num_channels=c, eps=1e-6, affine=True)


class ResnetBlock(nn.Module):
    """GroupNorm-swish-conv3 twice + skip, channel-first inside (reference models.py:20-80, temb unused)."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, dropout: float = 0.0):
        super().__init__()
        out_channels = out_channels or in_channels
        self.norm1 = _group_norm(in_channels)
        self.conv1 = nn.Conv1d(in_channels, out_channels, 3, padding=1)
        self.norm2 = _group_norm(out_channels)
        self.conv2 = nn.Conv1d(out_channels, out_channels, 3, padding=1)
        self.nin_shortcut = nn.Conv1d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x):                                                # [B,C,L]
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.nin_shortcut is None else self.nin_shortcut(x)) + h


class AttnBlock(nn.Module):
    """Single-head softmax attention over the sequence with 1x1-conv projections (reference models.py:82-125)."""

    def __init__(self, c: int):
        super().__init__()
        self.norm = _group_norm(c)
        self.q, self.k, self.v, self.proj_out = (nn.Conv1d(c, c, 1) for _ in range(4))

    def forward(self, x):                                                # [B,C,L]
        h = self.norm(x)
        q, k, v = (m(h).transpose(1, 2).unsqueeze(1) for m in (self.q, self.k, self.v))      # [B,1,L,C]
        o = F.scaled_dot_product_attention(q, k, v)                      # scale = C^-0.5, as the reference
        return x + self.proj_out(o.squeeze(1).transpose(1, 2))


class VocosBackbone(nn.Module):
    def __init__(self, input_channels: int, dim: int, intermediate_dim: int, num_layers: int,
                 layer_scale_init_value: Optional[float] = None, adanorm_num_embeddings: Optional[int] = None):
        super().__init__()
        self.input_channels = input_channels
        self.embed = nn.Conv1d(input_channels, dim, kernel_size=7, padding=3)
        self.adanorm = adanorm_num_embeddings is not None
        self.norm = AdaLayerNorm(adanorm_num_embeddings, dim, eps=1e-6) if self.adanorm else nn.LayerNorm(dim, eps=1e-6)
        lsv = layer_scale_init_value or 1 / num_layers
        self.convnext = nn.ModuleList([ConvNeXtBlock(dim, intermediate_dim, lsv, adanorm_num_embeddings)
                                       for _ in range(num_layers)])
        self.final_layer_norm = nn.LayerNorm(dim, eps=1e-6)
        self.pos_net = nn.Sequential(ResnetBlock(dim), ResnetBlock(dim), AttnBlock(dim), ResnetBlock(dim),
                                     ResnetBlock(dim), _group_norm(dim))

    def forward(self, x, bandwidth_id=None):
        """x [B, C_in, L] (the reference's feature layout) -> [B, L, dim]."""
        x = self.pos_net(self.embed(x)).transpose(1, 2)                  # channels-last from here on
        x = self.norm(x, bandwidth_id) if self.adanorm else self.norm(x)
        for blk in self.convnext:
            x = blk(x, bandwidth_id)
        return self.final_layer_norm(x)


class ISTFT(nn.Module):
    def __init__(self, n_fft: int, hop_length: int, win_length: int, padding: str = "same"):
        super().__init__()
        if padding != "same":
            raise NotImplementedError("only the 'same' padding the WavTokenizer head uses is built")
        if (win_length - hop_length) % 2:
            raise ValueError("win_length - hop_length must be even")
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self.register_buffer("window", torch.hann_window(win_length))

    def forward(self, spec):
        """spec complex [B, T, n_fft/2+1] (frame-major) -> audio [B, T*hop]."""
        frames = torch.fft.irfft(spec, self.n_fft, dim=-1, norm="backward")      # rocFFT, [B,T,n_fft]
        return ops.istft_ola(frames, self.window, self.hop_length)              # K9


class ISTFTHead(nn.Module):
    def __init__(self, dim: int, n_fft: int, hop_length: int, padding: str = "same"):
        super().__init__()
        self.out = nn.Linear(dim, n_fft + 2)
        self.istft = ISTFT(n_fft=n_fft, hop_length=hop_length, win_length=n_fft, padding=padding)

    def forward(self, x):                                                # [B,L,dim]
        mag, phase = self.out(x).float().chunk(2, dim=-1)
        return self.istft(torch.polar(torch.exp(mag).clamp(max=1e2), phase))


class WavTokenizerDecoder(nn.Module):
    """codes [K, B, L] (or [K, L]) -> audio [B, L*hop]; ``decode(features)`` as the reference's ``WavTokenizer.decode``."""

    def __init__(self, n_codes: int = 4096, n_codebooks: int = 1, codebook_dim: int = 512, dim: int = 768,
                 intermediate_dim: int = 2304, num_layers: int = 12, adanorm_num_embeddings: Optional[int] = 4,
                 n_fft: int = 1280, hop_length: int = 320):
        super().__init__()
        self.n_codes = n_codes
        # [K, n_codes, C]: the K residual codebooks (reference: the rows of quantizer.vq.layers[k].codebook stacked)
        self.codebook = nn.Parameter(torch.randn(n_codebooks, n_codes, codebook_dim), requires_grad=False)
        self.backbone = VocosBackbone(codebook_dim, dim, intermediate_dim, num_layers,
                                      adanorm_num_embeddings=adanorm_num_embeddings)
        self.head = ISTFTHead(dim, n_fft, hop_length)

    @torch.inference_mode()
    def codes_to_features(self, codes):
        """codes [K,B,L] (or [K,L]) -> sum over codebooks of the code vectors, [B, C, L]
        (reference pretrained.py:208-239; the gather is K6a)."""
        if codes.dim() == 2:
            codes = codes.unsqueeze(1)
        return ops.embed_sum(self.codebook, codes).transpose(1, 2)

    @torch.inference_mode()
    def decode(self, features, bandwidth_id=None):
        return self.head(self.backbone(features, bandwidth_id=bandwidth_id))

    @torch.inference_mode()
    def forward(self, codes, bandwidth_id=None):
        return self.decode(self.codes_to_features(codes), bandwidth_id=bandwidth_id)
