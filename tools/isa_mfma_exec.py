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
    """{kernel symbol: number of MFMAs inside an EXEC-masked region that has no s_cbranch_execz behind its saveexec}."""
    name, regions, flagged = None, [], {}
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, regions = m.group(1), []
            continue
        if name is None:
            continue
        t = line.strip()
        ms = re.match(r"s_(and|andn2|or)_saveexec_b64\s+(s\[\d+:\d+\])", t)
        mr = re.match(r"s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\])", t)
        if ms:
            nxt = next((l.strip() for l in lines[i + 1:i + 4] if l.strip() and not l.strip().startswith(";")), "")
            regions.append((ms.group(2), not nxt.startswith("s_cbranch_execz")))
        elif mr:                         # restoring a saved mask ends that region and every region opened inside it
            regs = [r for r, _ in regions]
            if mr.group(1) in regs:
                del regions[regs.index(mr.group(1)):]
            elif regions:
                regions.pop()
        elif t.startswith("v_mfma") and any(u for _, u in regions):
            flagged[name] = flagged.get(name, 0) + 1
        elif t.startswith("s_endpgm"):
            name = None
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
