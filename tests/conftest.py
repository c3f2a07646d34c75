"""pytest configuration: markers, import path, and the CPU wave64-emulator backend.

`-m "not gpu"` tests run the product's REAL host code and kernel SOURCES on the CPU by
binding ops to the emulated build of the same C ABI (tests/emu).  `-m gpu` tests run the
HIP library on a real MI355X.  The oracle (oracle/) is only ever the checker.
"""
import ctypes
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class EmuBackend:
    """Provider of the C ABI backed by tests/_emu_build/liblina_gla_emu.so (host pointers)."""
    name = "emu"

    def __init__(self, lib):
        self.lib = lib

    def require(self, *tensors):
        for t in tensors:
            if t is not None and t.is_cuda:
                raise RuntimeError("emulator backend takes CPU tensors")

    def stream(self, ref):
        return ctypes.c_void_p(0)


@pytest.fixture(scope="session")
def emu_lib():
    from emu import build_emu
    from lina_speech_amd import _lib
    return _lib.bind(build_emu.build(), hip_runtime=False)


@pytest.fixture()
def emu(emu_lib):
    """Route lina_speech_amd.ops through the CPU emulator for the duration of a test."""
    from lina_speech_amd import ops
    prev = ops.get_backend()
    ops.set_backend(EmuBackend(emu_lib))
    yield emu_lib
    ops.set_backend(prev)


@pytest.fixture()
def hip():
    """Make sure ops use the real HIP library on cuda:0 (gpu-marked tests)."""
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    from lina_speech_amd import ops
    prev = ops.get_backend()
    be = ops.HipBackend()
    be.lib  # load now: a missing .so must fail the test loudly
    ops.set_backend(be)
    yield be
    ops.set_backend(prev)


def golden(name):
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", name))
