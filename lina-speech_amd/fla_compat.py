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

import torch.nn as nn
import torch.nn.functional as F

from . import modules, ops
from .simple_gla import SimpleGatedLinearAttention  # noqa: F401  (fla.layers.simple_gla name)


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
