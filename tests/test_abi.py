"""The C-ABI shared library: it is built in-tree, loads, exports every symbol include/lina_gla.h
declares, and rejects bad arguments with error codes -- none of which needs a GPU."""
import ctypes
import os
import re

import pytest

from lina_speech_amd import _lib, build as hip_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    hip_build.build(verbose=False)
    return _lib.bind(_lib.LIB_PATH)


def test_header_and_library_agree(lib):
    hdr = open(os.path.join(ROOT, "include", "lina_gla.h")).read()
    declared = set(re.findall(r"\b(lina_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} not exported"
    assert lib.lina_version() == 100


def test_argument_errors_are_codes_not_crashes(lib):
    z = ctypes.c_void_p(0)
    st = _lib.BHT(0, 0, 0)
    rc = lib.lina_gla_recurrent_fwd(z, z, z, z, z, z, z, 1, 1, 1, 64, 64, st, st, st, st, st, 0, 0, 1.0, z)
    assert rc == -1 and b"null" in lib.lina_last_error()
    one = ctypes.c_void_p(16)
    rc = lib.lina_gla_chunk_fwd(one, one, one, one, one, z, z, 1, 1, 1, 48, 64, st, st, st, st, st, 0, 0, 1.0, z)
    assert rc == -2 and b"Dk=48" in lib.lina_last_error()
    rc = lib.lina_rmsnorm_gate_fwd(one, z, z, one, 4, 1, 6, 8, 0, 8, 0, 8, 0, 1, 0, 1e-5, 0, 0, z)
    assert rc == -1 and b"multiple of 4" in lib.lina_last_error()
    rc = lib.lina_gla_decode_prologue(one, 64, 0, 0, 0, 0, one, one, one, one, one, one, one, one, one, one,
                                      1, 64, 64, 3, 16, 16.0, float("nan"), 0, z)
    assert rc == -2 and b"W=3" in lib.lina_last_error()


def test_product_ops_refuse_cpu_tensors():
    import torch
    from lina_speech_amd import ops
    assert ops.get_backend().name == "hip"
    x = torch.randn(1, 1, 1, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.chunk_gla(x, x, x, x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.rmsnorm(torch.randn(4, 64))
