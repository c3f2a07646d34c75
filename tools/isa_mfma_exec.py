#!/usr/bin/env python
"""Audit: MFMA instructions inside EXEC-masked regions that are not skipped when EXEC is empty.  An MFMA issued under EXEC = 0
still executes -- on whatever its (unwritten, because masked) operand registers hold: found in round 3 as non-finite sums of
the 16-wave projection kernels on the hardware (the CPU emulator cannot show it).  Safe forms: a scalar branch around the
region (wave-uniform condition in an SGPR, `wave_uniform()`), or `s_cbranch_execz` right behind the `s_and_saveexec`.

    python tools/isa_mfma_exec.py [file.hip ...]      (exit status 1 if anything is flagged)
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "lina-speech_amd", "csrc")


def scan(lines):
    """{kernel symbol: number of MFMAs inside an EXEC-masked region that has no s_cbranch_execz behind its saveexec}.
    One forward pass; the only control flow it knows: behind an unconditional ``s_branch`` (or ``s_endpgm``) the fall-through is
    dead, so the block at the next label is entered by jumps only and starts with no open region (round 5: the gate path of the
    tall in-projection ends in a masked block that jumps back to its loop header -- without this the whole second half of the
    kernel, reached by a scalar branch from the top, was reported as masked)."""
    name, regions, flagged, pending, dead = None, [], {}, {}, False
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, regions, pending, dead = m.group(1), [], {}, False
            continue
        if name is None:
            continue
        t = line.strip()
        ml = re.match(r"^(\.LBB\d+_\d+):", t)
        if ml:                           # a block boundary: fall-through state (unless dead) + what earlier jumps carried here
            if dead:
                regions, dead = [], False
            for r in pending.pop(ml.group(1), []):
                if r not in regions:
                    regions.append(r)
            continue
        if re.match(r"s_branch\s", t):
            dead = True
            continue
        ms = re.match(r"s_(and|andn2|or)_saveexec_b64\s+(s\[\d+:\d+\])", t)
        mr = re.match(r"s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\])", t)
        if ms:
            nxt = next((l.strip() for l in lines[i + 1:i + 6] if l.strip().startswith("s_cbranch")), "")
            if ms.group(1) == "or" and regions:      # `s_or_saveexec sX, sX`: the else-flip of the if / else just opened:
                regions.pop()                        # it REPLACES that region (the join restores through this one's register)
            regions.append((ms.group(2), not nxt.startswith("s_cbranch_execz")))
        elif mr:                         # restoring a saved mask ends that region and every region opened inside it
            regs = [r for r, _ in regions]
            if mr.group(1) in regs:
                del regions[regs.index(mr.group(1)):]
            elif regions:
                regions.pop()
        elif t.startswith("v_mfma") and not dead and any(u for _, u in regions):
            flagged[name] = flagged.get(name, 0) + 1
        elif t.startswith("s_endpgm"):
            dead = True                  # (a kernel may end in several places: keep scanning, with a dead fall-through)
    return flagged


def main(files):
    bad = 0
    for f in files:
        out = "/tmp/_isa_%s.s" % os.path.basename(f)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                        "-fno-strict-aliasing", "-Wno-inline-asm", "-I", CS, "-I", os.path.join(ROOT, "include"),
                        "--cuda-device-only", "-S", f, "-o", out], stderr=subprocess.DEVNULL)
        if not os.path.exists(out):
            continue
        flagged = scan(open(out).read().splitlines())
        if flagged:
            names = subprocess.run(["c++filt"], input="\n".join(flagged), stdout=subprocess.PIPE, text=True).stdout.splitlines()
            for n, k in zip(names, flagged.values()):
                print(f"{os.path.basename(f):22s} {n.split('(')[0].replace('void lina::', '')[:90]:90s} "
                      f"MFMAs under an unskipped EXEC mask: {k}")
                bad += 1
    print("flagged kernels:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or sorted(glob.glob(os.path.join(CS, "*.hip")))))
