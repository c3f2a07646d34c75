"""Build the C-ABI shared library (hand-written HIP, gfx950 only) in-tree.

    python -m lina_speech_amd.build          # or: __graft_entry__.build()

Output: lina-speech_amd/csrc/liblina_gla.so (git-ignored; travels to the GPU box).
hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "liblina_gla.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-strict-aliasing",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-inline-asm"]


# Kernels that wait for their global->LDS DMA by COUNTING vector-memory operations (wait_vmem_but<N>: "all but the last N
# loads"): a register spill the compiler adds would put scratch loads into that count, so these files must compile to
# kernels without scratch -- checked from the compiler's own resource remarks at build time.
NO_SCRATCH = {"gla_chunk_full.hip", "gla_decode_window.hip"}   # K1w: a spilled tile register waits for its load mid-issue


def _check_no_scratch(src: str, out: str) -> str:
    """Fail the build if a kernel of ``src`` uses scratch; return the compiler output without the resource remarks."""
    import re
    name, keep, in_remark, n_seen = None, [], False, 0
    for line in out.splitlines():
        if in_remark and re.match(r"^\s*(\d+\s*)?\|", line):      # the source snippet / caret under a remark
            continue
        in_remark = False
        if "-Rpass-analysis=kernel-resource-usage" in line or "remark:" in line:
            in_remark = True
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            n_seen += m is not None
            if m and int(m.group(1)) != 0:
                raise RuntimeError(f"{os.path.basename(src)}: kernel {name} spills to scratch ({m.group(1)} bytes/lane); "
                                   "its DMA wait counts vector-memory operations -- reduce register pressure")
            continue
        keep.append(line)
    if n_seen == 0:       # a changed remark format must not turn the check into a no-op
        raise RuntimeError(f"{os.path.basename(src)}: no ScratchSize remark found in the compiler output -- the "
                           "no-scratch check of its DMA-counting kernels did not run")
    return "\n".join(keep)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        cmd = [HIPCC, *[f for f in FLAGS if f != "-shared"], "-I", CSRC, "-c", src, "-o", obj]
        if os.path.basename(src) in NO_SCRATCH:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    bad = False
    for src, p in procs:
        out, _ = p.communicate()
        if os.path.basename(src) in NO_SCRATCH:
            out = _check_no_scratch(src, out)
        if p.returncode != 0 or (verbose and out.strip()):
            sys.stderr.write(f"--- hipcc {os.path.basename(src)}\n{out}\n")
        bad |= p.returncode != 0
    if bad:
        raise RuntimeError("hipcc failed (see messages above)")
    # Link WITHOUT a DT_NEEDED on a particular libamdhip64: the HIP symbols resolve at load time
    # against the runtime the host process already holds (torch bundles its own copy with no SONAME;
    # a second runtime in one process cannot see torch's streams/devices).  See _lib.bind().
    cmd = [os.environ.get("CXX", "g++"), "-shared", "-fPIC", *objs, "-o", LIB]
    subprocess.run(cmd, check=True)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
