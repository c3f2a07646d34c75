"""ctypes binding of the C-ABI library (include/lina_gla.h).

The HIP library is the ONLY compute backend of this package: ``load()`` raises if
``csrc/liblina_gla.so`` is missing (run ``python -m lina_speech_amd.build`` or
``__graft_entry__.build()``) -- there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LINA_GLA_LIB") or os.path.join(_HERE, "csrc", "liblina_gla.so")   # override: perf-analysis builds

LINA_F32, LINA_BF16 = 0, 1
CONV_BWD_TT = 64          # LINA_CONV_BWD_TT in include/lina_gla.h
LOOP_CTL_ROWS = 4         # LINA_LOOP_CTL_ROWS: int32 words in front of the per-row stop flags of a loop-control block


class BHT(C.Structure):
    """lina_bht_strides: element strides of a head-first [B,H,T,D] view."""
    _fields_ = [("b", C.c_int64), ("h", C.c_int64), ("t", C.c_int64)]


_p, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

_GLA_SIG = [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, BHT, BHT, BHT, BHT, BHT, _i, _i, _f, _p]

PROTOTYPES = {
    "lina_version": (C.c_int, []),
    "lina_last_error": (C.c_char_p, []),
    "lina_gla_recurrent_fwd": (C.c_int, _GLA_SIG),
    "lina_gla_chunk_fwd": (C.c_int, _GLA_SIG),
    "lina_gla_chunk_fwd_seg_workspace": (C.c_int64, [_i, _i, _i, _i, _i]),
    "lina_gla_chunk_fwd_seg": (C.c_int, [_p] * 8 + [_i] * 6 + [BHT] * 5 + [_i, _i, _f, _p]),
    "lina_gla_chunk_bwd_workspace": (C.c_int64, [_i, _i, _i, _i, _i]),
    "lina_gla_chunk_bwd": (C.c_int, [_p] * 14 + [_i] * 5 + [BHT] * 9 + [_i, _i, _f, _p]),
    "lina_gla_chunk_bwd_full_workspace": (C.c_int64, [_i, _i, _i, _i, _i, _i]),
    "lina_gla_chunk_bwd_full": (C.c_int, [_p] * 15 + [_i] * 6 + [BHT] * 9 + [_i, _i, _f, _p]),
    "lina_short_conv_fwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _i, _p]),
    "lina_short_conv_step": (C.c_int, [_p, _p, _p, _p, _p, _i, _i, _i, _i64, _i64, _i, _i, _p]),
    "lina_short_conv_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64,
                                      _i, _i, _p]),
    "lina_rmsnorm_gate_bwd_partials": (C.c_int, [_i64]),
    "lina_rmsnorm_gate_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i64, _i64, _i64, _i64, _f, _i, _p]),
    "lina_rmsnorm_gate_fwd": (C.c_int, [_p, _p, _p, _p, _i64, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _i, _i64,
                                        _f, _i, _i, _p]),
    "lina_embed_sum": (C.c_int, [_p, _p, _p, _i, _i64, _i, _i, _i, _p]),
    "lina_greedy_pick_embed": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "lina_sample_pick_embed": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, C.c_uint64,
                                         _i, _p]),
    "lina_argmax_rows": (C.c_int, [_p, _p, _i64, _i, _i64, _i, _p]),
    "lina_topk_sample_rows": (C.c_int, [_p, _p, _i64, _i, _i64, _i, _f, _p, C.c_uint64, _p, _i, _p]),
    "lina_dwconv7_ln": (C.c_int, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i64, _f, _i, _p]),
    "lina_istft_ola": (C.c_int, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "lina_gla_decode_prologue": (C.c_int, [_p, _i64, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                           _i, _i, _i, _i, _i, _f, _f, _i, _p]),
    "lina_swiglu": (C.c_int, [_p, _p, _i64, _i, _i64, _i64, _i, _p]),
    "lina_swiglu_bwd": (C.c_int, [_p, _p, _p, _i64, _i, _i64, _i64, _i64, _i, _p]),
    "lina_cross_entropy": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i, _i64, _i64, _i64, _i, _p]),
    "lina_colsum": (C.c_int, [_p, _p, _i64, _i, _i64, _i, _p]),
    "lina_sum_partials": (C.c_int, [_p, _p, _i, _i, _i64, _i, _p]),
    "lina_mlp_pack": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "lina_stack_rows": (C.c_int, [_p, _p, _i, _i, _i, _p, _i, _p]),
    "lina_adamw_multi": (C.c_int, [_p, _p, _p, _p, _p, _i] + [C.c_double] * 7 + [_p]),
    "lina_adamw_multi_max": (C.c_int, []),
    "lina_swiglu_bwd_partials": (C.c_int, [_i64]),
    "lina_swiglu_bwd_colsum": (C.c_int, [_p, _p, _p, _p, _i64, _i, _i64, _i64, _i64, _i, _p]),
    "lina_gate_lowrank_partials": (C.c_int, [_i64]),
    "lina_gate_lowrank": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _i64, _i, _i, _f, _f, _i, _p]),
    "lina_gate_logsigmoid": (C.c_int, [_p, _p, _p, _i64, _f, _f, _i, _p]),
    "lina_layernorm_fwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _f, _i, _i, _i, _p]),
    "lina_layernorm_bwd_partials": (C.c_int, [_i64]),
    "lina_layernorm_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _p]),
    "lina_gla_decode_update": (C.c_int, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64,
                                         _i64, _i64, _i, _i, _f, _p]),
    "lina_gla_decode_update_norm": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i,
                                              _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f,
                                              _i, _i, _f, _p]),
    "lina_gla_decode_window_max": (C.c_int, []),
    "lina_gla_decode_window": (C.c_int, [_p] * 15 + [_i] * 5 + [_i64] * 10 + [_f, _i, _i, _i, _f, _p]),
    "lina_gla_decode_window_flush": (C.c_int, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "lina_gla_decode_window_s": (C.c_int, [_p] * 5 + [_i] + [_p] * 10 + [_i] * 5 + [_i64] * 10 + [_f, _i, _i, _i, _f, _p]),
    "lina_gla_decode_window_flush_s": (C.c_int, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "lina_linear_skinny_ex": (C.c_int, [_p, _i64, _p, _i64, _i, _i, _p, _p, _p, _i64, _p, _i64, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "lina_gla_decode_inproj_packed": (C.c_int, [_p] * 15 + [_i] * 6 + [_f, _f, _f, _i, _i, _p]),
    "lina_weighted_rows_add_packed": (C.c_int, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "lina_cross_att_step1": (C.c_int, [_p, _p, _p, _f, _p, _p, _p, _i64, _p, _i, _i, _i, _f, _i, _p]),
    "lina_cross_att_step2": (C.c_int, [_p, _p, _p, _p, _i64, _p, _i, _i, _i, _f, _i, _p]),
    "lina_cross_scores_softmax": (C.c_int, [_p, _p, _p, _f, _p, _p, _i64, _p, _i, _i, _i, _i, _f, _i, _p]),
    "lina_softmax_weighted_rows_add": (C.c_int, [_p, _i64, _f, _p, _i64, _p, _p, _p, _i, _i, _i, _i, _p]),
    "lina_pe_softmax_weighted_rows_add": (C.c_int, [_p, _i, _p, _f, _p, _i64, _p, _i64, _i64, _p, _p, _p, _i, _i, _i, _i, _p]),
    "lina_softmax_pe_rows": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _p, _p, _p, _i, _i, _i, _i, _p]),
    "lina_cross_scores": (C.c_int, [_p, _p, _p, _f, _p, _p, _i, _i, _i, _f, _i, _p]),
    "lina_softmax_rows": (C.c_int, [_p, _i64, _i, _f, _p, _i64, _p, _i, _i, _i, _i, _p]),
    "lina_weighted_rows_add": (C.c_int, [_p, _i, _p, _p, _i, _i, _i, _i, _p]),
    "lina_gla_decode_inproj": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                         _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _p]),
    "lina_linear_skinny": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p, _i64, _p, _i64, _i, _i, _i, _i, _i, _f, _i, _p]),
}


def _preload_hip_runtime() -> None:
    """The library is linked without a DT_NEEDED for libamdhip64 (build.py): it must bind to the ONE
    HIP runtime of the process -- torch's bundled copy -- or its launches could not use torch's
    streams.  Import torch and put that runtime's symbols in the global scope before dlopen."""
    import torch  # noqa: F401  (loads torch/lib/libamdhip64.so)
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"),
             "/opt/rocm/lib/libamdhip64.so"]
    for c in cands:
        if os.path.exists(c):
            C.CDLL(c, mode=C.RTLD_GLOBAL)
            return
    raise RuntimeError("no libamdhip64.so found (torch/lib or /opt/rocm/lib)")


def bind(path: str, hip_runtime: bool = True) -> C.CDLL:
    """dlopen `path` and attach the prototypes of include/lina_gla.h (raises if a symbol is missing)."""
    if hip_runtime:
        _preload_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"lina_speech_amd: HIP library not built ({LIB_PATH} missing). Run `python -m lina_speech_amd.build`."
                " There is no CPU fallback.")
        _lib = bind(LIB_PATH)
    return _lib


class LinaError(RuntimeError):
    pass


def check(rc: int, lib=None) -> None:
    if rc != 0:
        lib = lib or load()
        raise LinaError(f"lina C-ABI error {rc}: {lib.lina_last_error().decode()}")
