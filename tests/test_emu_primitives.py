"""The emulator's model of device primitives whose semantics were MEASURED on the hardware: checked against the committed
probe output (profiles/), so that kernels written on top of them can be developed on the CPU."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _tables(text):
    """{pattern: {lane: [(row, col) x 4]}} from either output format."""
    out, pat = {}, None
    for m in re.finditer(r"pattern (\d+)|lane\s+(\d+):((?:\s*\(\s*\d+,\s*\d+\))+)", text):
        if m.group(1) is not None:
            pat = int(m.group(1))
            out[pat] = {}
        else:
            out[pat][int(m.group(2))] = [(int(a), int(b)) for a, b in re.findall(r"\(\s*(\d+),\s*(\d+)\)", m.group(3))]
    return out


def test_transposing_lds_read_matches_the_hardware_probe(tmp_path):
    exe = str(tmp_path / "tr_read_check")
    emu = os.path.join(HERE, "emu")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", emu, "-I", os.path.join(ROOT, "lina-speech_amd", "csrc"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(emu, "tr_read_check.cpp"),
                    os.path.join(emu, "emu_runtime.cpp"), "-o", exe], check=True)
    got = _tables(subprocess.run([exe], check=True, capture_output=True, text=True).stdout)
    ref = _tables(open(os.path.join(ROOT, "profiles", "r02_tr_read_probe.txt")).read())
    assert set(ref) == {0, 1, 2} and all(len(ref[p]) == 64 for p in ref)
    assert got == ref
