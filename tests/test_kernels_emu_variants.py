"""Opt-in kernel variants that the default build does not ship yet, parity-tested on the emulator so that the next GPU
session can go straight to measuring them.  LINA_K2_TR: K2 / K2b without transposed operand tiles (ds_read_b64_tr_b16)."""
import pytest
import torch

from kernel_cases import check_chunk, check_chunk_bwd_full, check_chunk_segmented

DEV = "cpu"


@pytest.fixture(scope="module")
def emu_tr():
    from conftest import EmuBackend
    from emu import build_emu
    from lina_speech_amd import _lib, ops
    prev = ops.get_backend()
    ops.set_backend(EmuBackend(_lib.bind(build_emu.build(defs=("-DLINA_K2_TR=1",), tag="tr", only=("gla_chunk_full.hip",)), hip_runtime=False)))
    yield
    ops.set_backend(prev)


@pytest.mark.parametrize("T,resets", [(1, False), (33, False), (70, True), (200, False)])
def test_tr_variant_chunk_forward(emu_tr, T, resets):
    check_chunk(DEV, B=1, H=1, T=T, Dk=256, Dv=256, dtype=torch.bfloat16, resets=resets)


def test_tr_variant_head_groups_and_segments(emu_tr):
    check_chunk(DEV, B=1, H=4, T=70, Dk=64, Dv=64, dtype=torch.bfloat16, resets=True)
    check_chunk_segmented(DEV, 1, 2, 100, 3, resets=True, D=128)


@pytest.mark.parametrize("T,nseg,D,H", [(40, 1, 256, 1), (100, 3, 256, 1), (70, 2, 128, 2)])
def test_tr_variant_backward_sweeps(emu_tr, T, nseg, D, H):
    check_chunk_bwd_full(DEV, 1, H, T, D, nseg, resets=True)


@pytest.fixture(scope="module")
def emu_w32():
    from conftest import EmuBackend
    from emu import build_emu
    from lina_speech_amd import _lib, ops
    prev = ops.get_backend()
    lib = build_emu.build(defs=("-DLINA_K2_TR=1", "-DLINA_K2_W32=1"), tag="w32", only=("gla_chunk_full.hip",))
    ops.set_backend(EmuBackend(_lib.bind(lib, hip_runtime=False)))
    yield
    ops.set_backend(prev)


# LINA_K2_W32: a wave owns 128 state rows x 32 columns (every sweep at the L169 head shape)
@pytest.mark.parametrize("T,resets", [(1, False), (17, False), (33, False), (70, True), (200, False)])
def test_w32_variant_chunk_forward(emu_w32, T, resets):
    check_chunk(DEV, B=1, H=1, T=T, Dk=256, Dv=256, dtype=torch.bfloat16, resets=resets)


def test_w32_variant_segments_and_other_shapes(emu_w32):
    check_chunk_segmented(DEV, 1, 1, 100, 3, resets=True, D=256)
    check_chunk(DEV, B=1, H=2, T=40, Dk=128, Dv=128, dtype=torch.bfloat16, resets=True)    # G = 2 keeps the 256 x 16 form


@pytest.mark.parametrize("T,nseg,h0,dht", [(40, 1, True, True), (100, 3, True, True), (33, 1, False, False), (200, 2, True, False)])
def test_w32_variant_backward_sweeps(emu_w32, T, nseg, h0, dht):
    check_chunk_bwd_full(DEV, 1, 1, T, 256, nseg, resets=True, with_h0=h0, with_dht=dht)


@pytest.mark.parametrize("defs", [("-DLINA_K2_TR=1",), ("-DLINA_K2_TR=1", "-DLINA_K2_W32=1")])
def test_variants_cross_compile_for_gfx950_without_scratch(defs, tmp_path):
    """The variants wait for their DMA by counting vector-memory operations like the default build: no kernel may spill."""
    import os
    import subprocess
    from lina_speech_amd import build
    if not os.path.exists(build.HIPCC):
        pytest.skip("no hipcc")
    src = os.path.join(build.CSRC, "gla_chunk_full.hip")
    cmd = [build.HIPCC, *[f for f in build.FLAGS if f != "-shared"], *defs, "-I", build.CSRC, "-c", src,
           "-o", str(tmp_path / "v.o"), "-Rpass-analysis=kernel-resource-usage"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    build._check_no_scratch(src, p.stdout + p.stderr)         # raises if a kernel uses scratch
