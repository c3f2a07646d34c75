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
    # LINA_EMU_VARIANT="tag:-DX=1,-DY=2:file.hip,other.hip": run the session on an opt-in build of the kernel sources (e.g. the
    # C = 32 forward kernel behind -DLINA_K2_NOPIPE=1), built beside the default library
    var = os.environ.get("LINA_EMU_VARIANT")
    if var:
        tag, defs, only = (var.split(":") + ["", ""])[:3]
        return _lib.bind(build_emu.build(defs=tuple(d for d in defs.split(",") if d), tag=tag,
                                         only=[f for f in only.split(",") if f] or None), hip_runtime=False)
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



# ----------------------------------------------------------------------------- achieved errors of a GPU session
# Every assert_close / record_parity call (tests/kernel_cases.py) logs the error it ACHIEVED next to its tolerance; after a
# session that ran on a GPU the log is written to gpurun_out/parity_<tag>.json (LINA_PARITY_TAG, default "gpu"): worst case per
# (test function, label).  The committed copy lives under profiles/ (r03_parity.json): a reader can tell a 2e-2 budget from a
# 2e-3 result.
@pytest.fixture(autouse=True)
def _tag_parity_entries(request):
    import kernel_cases
    n0 = len(kernel_cases.PARITY_LOG)
    yield
    for e in kernel_cases.PARITY_LOG[n0:]:
        e.setdefault("test", request.node.name)
        e.setdefault("file", os.path.basename(str(request.node.fspath)))
        e.setdefault("gpu", request.node.get_closest_marker("gpu") is not None)


def pytest_sessionfinish(session, exitstatus):
    import json
    try:
        import kernel_cases
    except Exception:
        return
    rows = [e for e in kernel_cases.PARITY_LOG if e.get("gpu")]
    if not rows or not torch.cuda.is_available():
        return
    worst = {}
    for e in rows:
        fn = e["test"].split("[")[0]
        key = (e["file"], fn, e["what"])
        w = worst.get(key)
        if w is None or e["achieved"] > w["achieved"]:
            worst[key] = {**{k: v for k, v in e.items() if k not in ("test", "gpu")}, "test": fn, "cases": 0}
    for e in rows:
        worst[(e["file"], e["test"].split("[")[0], e["what"])]["cases"] += 1
    out = {"device": torch.cuda.get_device_name(0), "exit_status": int(exitstatus), "n_checks": len(rows),
           "what": "worst achieved error per (test, label) of this pytest -m gpu session; 'achieved' and 'tolerance' are max "
                   "|got - ref| / max |ref| unless the entry says otherwise",
           "entries": sorted(worst.values(), key=lambda e: (e["file"], e["test"], e["what"]))}
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"parity_{os.environ.get('LINA_PARITY_TAG', 'gpu')}.json"), "w") as f:
        json.dump(out, f, indent=1)
