"""The provider of the C ABI behind every operator (the in-tree HIP library on a ROCm device; the test-side wave64
emulator installs another one with ``set_backend``), the tensor -> ABI argument helpers and the scratch-workspace cache.
There is no CPU path: ``HipBackend.require`` raises on tensors that do not live on a ROCm device."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import BHT, LINA_BF16, LINA_F32


class HipBackend:
    """Default provider of the C ABI: the in-tree HIP library, tensors on a ROCm device."""
    name = "hip"

    def __init__(self):
        self._lib = None

    @property
    def lib(self):
        if self._lib is None:
            self._lib = _lib.load()
        return self._lib

    def require(self, *tensors):
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise RuntimeError("lina_speech_amd ops run on a ROCm GPU only (got a %s tensor); "
                                   "there is no CPU fallback" % t.device)

    def stream(self, ref: torch.Tensor):
        return C.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream)


_BACKEND = HipBackend()


def set_backend(backend) -> None:
    """Install another provider of the same C ABI (object with .lib/.require/.stream)."""
    global _BACKEND
    _BACKEND = backend


def get_backend():
    return _BACKEND


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return LINA_F32
    if t.dtype == torch.bfloat16:
        return LINA_BF16
    raise TypeError(f"unsupported dtype {t.dtype} (float32 or bfloat16)")


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _no_grad(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError("lina_speech_amd: this op has no backward kernel (inference-only: SwiGLU decode "
                                  "epilogue, vocoder K8/K9); call it under torch.no_grad()/inference_mode()")


def _inner_contig(t: torch.Tensor) -> torch.Tensor:
    return t if t.stride(-1) == 1 else t.contiguous()


def _bht(t: torch.Tensor) -> BHT:
    return BHT(t.stride(0), t.stride(1), t.stride(2))


def _check(rc: int):
    if rc != 0:
        raise _lib.LinaError(f"lina C-ABI error {rc}: {_BACKEND.lib.lina_last_error().decode()}")


_WORKSPACES = {}


def _workspace(tag: str, nbytes: int, device) -> torch.Tensor:
    """Scratch that is fully written before it is read inside ONE launch sequence on the current stream (segment
    states of the segment-parallel K2): kept per (tag, device, stream) and grown on demand instead of a torch.empty
    per layer per step.  Stream-ordered reuse is safe because consecutive users on one stream serialise."""
    if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        # inside a graph capture the allocation belongs to the graph's private pool: never hand it to eager launches
        return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    key = (tag, device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _WORKSPACES[key] = ws
    return ws


def clear_workspaces() -> None:
    """Release the cached kernel scratch (segment / boundary states of the segment-parallel K2 / K2b): at small B*H with
    many segments it is the size of several activations and would otherwise stay pinned for the life of the process."""
    _WORKSPACES.clear()
    from .autograd import clear_mlp_pack          # (the padded SwiGLU weight cache: see its note on ``param.data`` writes)
    clear_mlp_pack()


def fused_ops_available(x: torch.Tensor) -> bool:
    """True when the HIP (or emulated) ops can take ``x``: a ROCm tensor, or a CPU tensor under the test emulator."""
    return x.is_cuda if _BACKEND.name == "hip" else not x.is_cuda
