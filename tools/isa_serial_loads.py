#!/usr/bin/env python
"""Audit: kernels whose global loads are waited for one at a time.  A predicated load (`ok ? ld(p) : 0`) compiles to an
EXEC-masked region with `s_waitcnt vmcnt(0)` behind it, so N such loads cost N memory round trips (found in round 3 in the
decode step's epilogue preloads and the cross-attention scores kernel).  For every kernel of every csrc/*.hip: the number of
global loads and how many of them are followed by `s_waitcnt vmcnt(0)` before the next load is issued.

    python tools/isa_serial_loads.py [file.hip ...]
"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "lina-speech_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(CS, "*.hip")))
for f in files:
    out = "/tmp/_isa_%s.s" % os.path.basename(f)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    "-fno-strict-aliasing", "-Wno-inline-asm", "-I", CS, "-I", os.path.join(ROOT, "include"),
                    "--cuda-device-only", "-S", f, "-o", out], stderr=subprocess.DEVNULL)
    if not os.path.exists(out):
        continue
    name, rows = None, []
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            cur = {"name": name, "loads": 0, "serial": 0, "pending": False}
            rows.append(cur)
            continue
        if name is None:
            continue
        t = line.strip()
        if re.match(r"(global|buffer|flat)_load", t):
            cur["loads"] += 1
            cur["pending"] = True
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            if cur["pending"]:
                cur["serial"] += 1
            cur["pending"] = False
        elif t.startswith("s_endpgm"):
            name = None
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        if r["serial"] >= 4:
            n = n.split("(")[0].replace("void lina::", "")
            print(f"{os.path.basename(f):22s} {n[:84]:84s} loads={r['loads']:3d} waited-for-one-by-one={r['serial']:3d}")
