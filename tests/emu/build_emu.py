"""TEST INFRASTRUCTURE: compile the kernel sources of lina-speech_amd/csrc for the CPU
wave64 emulator (tests/emu/lina_dev.h shadows csrc/lina_dev.h) -> tests/_emu_build/liblina_gla_emu.so.
The emulated library exposes the same C ABI (include/lina_gla.h); pointers are host pointers."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "lina-speech_amd", "csrc")
OUT = os.path.join(ROOT, "tests", "_emu_build")
LIB = os.path.join(OUT, "liblina_gla_emu.so")
CXX = os.environ.get("CXX", "g++")
# -fno-gnu-unique: the kernels' `__shared__` arrays are function-local statics of template functions; with GNU unique symbols
# two emulator libraries loaded into one process (the default build and an opt-in variant) would SHARE them, sized by
# whichever was loaded first
FLAGS = ["-O2", "-std=c++17", "-fPIC", "-fno-gnu-unique", "-ffp-contract=off", "-fno-strict-aliasing", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         "-Wno-unknown-pragmas", "-Wno-attributes", "-Wno-sign-compare"]


def stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "*.cpp")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, defs=(), tag: str = "", only=None) -> str:
    """``defs`` / ``tag``: an opt-in kernel variant (e.g. defs=("-DSOME_EXPERIMENT=1",), tag="exp") built beside the default
    library as liblina_gla_emu_<tag>.so, so that variants the product does not ship yet are still parity-tested.
    ``only``: the source files the definitions affect -- the others are linked from the default build's objects."""
    lib = LIB if not tag else LIB[:-3] + f"_{tag}.so"
    if tag and only:
        build(force)                                        # the default objects must exist and be current
    if not force and not stale(lib):
        return lib
    os.makedirs(OUT, exist_ok=True)
    procs, objs = [], []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(HERE, "emu_runtime.cpp")]:
        if tag and only and os.path.basename(src) not in only:
            objs.append(os.path.join(OUT, os.path.basename(src) + ".o"))
            continue
        obj = os.path.join(OUT, os.path.basename(src) + (f".{tag}" if tag else "") + ".o")
        cmd = [CXX, *FLAGS, *defs, "-I", HERE, "-I", CSRC, "-x", "c++", "-c", src, "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    bad = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip():
            sys.stderr.write(f"--- {CXX} {os.path.basename(src)}\n{out}\n")
        bad |= p.returncode != 0
    if bad:
        raise RuntimeError("emulator build failed")
    subprocess.run([CXX, "-shared", "-fPIC", *objs, "-o", lib], check=True)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
