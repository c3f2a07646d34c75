"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's codes -> waveform step (WavTokenizer decoder),
the checker for lina_speech_amd/vocoder.py and the K8 / K9 kernels.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline may import this; the product never does.

Every function cites the reference lines it restates (3rdparty/decoder/*).  Pinned by goldens captured from the
reference's own modules (VocosBackbone, ISTFTHead) with seeded weights: tests/golden/vocoder_small.npz
(tests/golden/make_golden.py::golden_vocoder).  Plain torch ops, any float dtype (fp64 in the tests).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def dwconv7_ln(x, w, bias=None, scale=None, shift=None, eps=1e-6):
    """ConvNeXtBlock.dwconv + (Ada)LayerNorm, channels-last x [B,L,C] (modules.py:44-50, 62-82)."""
    C = x.shape[-1]
    z = F.conv1d(x.transpose(1, 2), w.reshape(C, 1, 7), bias, padding=3, groups=C).transpose(1, 2)
    z = F.layer_norm(z, (C,), eps=eps)
    if scale is not None:
        z = z * (scale if scale.dim() == 1 else scale.unsqueeze(1))
    if shift is not None:
        z = z + (shift if shift.dim() == 1 else shift.unsqueeze(1))
    return z


def istft_same(frames, window, hop):
    """Windowed overlap-add / envelope / trim of spectral_ops.py:56-75, frames [B,T,win] (after irfft)."""
    B, T, win = frames.shape
    pad = (win - hop) // 2
    out_size = (T - 1) * hop + win
    fr = (frames * window).transpose(1, 2)                                           # [B,win,T]
    y = F.fold(fr, output_size=(1, out_size), kernel_size=(1, win), stride=(1, hop))[:, 0, 0, pad:out_size - pad]
    wsq = window.square().expand(1, T, -1).transpose(1, 2)
    env = F.fold(wsq, output_size=(1, out_size), kernel_size=(1, win), stride=(1, hop)).reshape(-1)[pad:out_size - pad]
    return y / env


class OracleVocoder:
    """Functional decode over a reference-keyed state dict ``backbone.* / head.*``."""

    def __init__(self, sd: dict, num_layers: int, n_fft: int, hop: int, adanorm: bool = True, dtype=torch.float64):
        self.sd = {k: v.to(dtype) if v.is_floating_point() else v for k, v in sd.items()}
        self.n, self.n_fft, self.hop, self.adanorm = num_layers, n_fft, hop, adanorm

    def _gn(self, x, p):                                              # models.py:14-15
        return F.group_norm(x, 32, self.sd[p + ".weight"], self.sd[p + ".bias"], eps=1e-6)

    def _conv(self, x, p, pad):
        return F.conv1d(x, self.sd[p + ".weight"], self.sd[p + ".bias"], padding=pad)

    def _resnet(self, x, p):                                          # models.py:59-80
        h = self._conv(F.silu(self._gn(x, p + ".norm1")), p + ".conv1", 1)
        h = self._conv(F.silu(self._gn(h, p + ".norm2")), p + ".conv2", 1)
        return x + h

    def _attn(self, x, p):                                            # models.py:107-125
        h = self._gn(x, p + ".norm")
        q, k, v = (self._conv(h, f"{p}.{n}", 0) for n in "qkv")
        w = torch.softmax(torch.bmm(q.transpose(1, 2), k) * q.shape[1] ** -0.5, dim=2)
        return x + self._conv(torch.bmm(v, w.transpose(1, 2)), p + ".proj_out", 0)

    def _ada(self, p, bw):                                            # modules.py:62-82
        if self.adanorm:
            return self.sd[p + ".scale.weight"][bw], self.sd[p + ".shift.weight"][bw]
        return self.sd[p + ".weight"], self.sd[p + ".bias"]

    def backbone(self, feats, bw=None):                               # models.py:217-229
        sd = self.sd
        x = self._conv(feats.to(next(iter(sd.values())).dtype), "backbone.embed", 3)
        for i in (0, 1):
            x = self._resnet(x, f"backbone.pos_net.{i}")
        x = self._attn(x, "backbone.pos_net.2")
        for i in (3, 4):
            x = self._resnet(x, f"backbone.pos_net.{i}")
        x = self._gn(x, "backbone.pos_net.5").transpose(1, 2)
        sc, sh = self._ada("backbone.norm", bw)
        x = F.layer_norm(x, (x.shape[-1],), eps=1e-6)
        x = x * (sc if sc.dim() == 1 else sc.unsqueeze(1)) + (sh if sh.dim() == 1 else sh.unsqueeze(1))
        for i in range(self.n):                                       # modules.py:42-60
            p = f"backbone.convnext.{i}"
            sc, sh = self._ada(p + ".norm", bw)
            h = dwconv7_ln(x, sd[p + ".dwconv.weight"], sd[p + ".dwconv.bias"], sc, sh)
            h = F.linear(F.gelu(F.linear(h, sd[p + ".pwconv1.weight"], sd[p + ".pwconv1.bias"])),
                         sd[p + ".pwconv2.weight"], sd[p + ".pwconv2.bias"])
            x = x + sd[p + ".gamma"] * h
        return F.layer_norm(x, (x.shape[-1],), sd["backbone.final_layer_norm.weight"],
                            sd["backbone.final_layer_norm.bias"], eps=1e-6)

    def head(self, x):                                                # heads.py:44-67
        o = F.linear(x, self.sd["head.out.weight"], self.sd["head.out.bias"])
        mag, ph = o.chunk(2, dim=-1)
        spec = torch.polar(torch.exp(mag).clamp(max=1e2), ph)
        frames = torch.fft.irfft(spec, self.n_fft, dim=-1, norm="backward")
        return istft_same(frames, self.sd["head.istft.window"], self.hop)

    def decode(self, feats, bw=None):                                 # pretrained.py:193-205
        return self.head(self.backbone(feats, bw))
