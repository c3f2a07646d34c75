#!/usr/bin/env python
"""Register / scratch / LDS use of every kernel of one csrc file (compiler remarks), demangled:
    python tools/kres.py gla_decode_window [-DMACRO=1 ...] [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "lina-speech_amd", "csrc")
name = sys.argv[1]
defs = [a for a in sys.argv[2:] if a.startswith("-")]
flt = [a for a in sys.argv[2:] if not a.startswith("-")]
src = name if os.path.exists(name) else os.path.join(CS, name + ".hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-strict-aliasing",
       "-Wno-inline-asm", "-I", CS, "-I", os.path.join(ROOT, "include"), *defs, "-c", src, "-o", "/tmp/kres.o",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur, rows = None, []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
if not rows:
    print(out)
    sys.exit(1)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), stdout=subprocess.PIPE,
                       text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.split("(")[0].replace("void lina::", "")
    if all(f in n for f in flt):
        print(f"{n[:90]:90s} vgpr={r.get('vgpr')} agpr={r.get('agpr')} sgpr={r.get('sgpr')} scratch={r.get('scratch')} lds={r.get('lds')} occ={r.get('occ')}")
