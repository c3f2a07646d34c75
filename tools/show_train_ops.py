"""Compact view of gpurun_out/train_ops.txt (tools/prof_train_ops.py): ops only (kernels dropped), ms per step."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/train_ops.txt"
n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 60
txt = open(path).read()
steps = int(re.search(r"# (\d+) steps", txt).group(1))
lines = txt.split("# per op")[0].split("\n")


def us(s):
    m = re.match(r"([\d.]+)(us|ms|s)$", s)
    return float(m.group(1)) * {"us": 1, "ms": 1e3, "s": 1e6}[m.group(2)]


hdr, n = None, 0
for ln in lines:
    if ln.startswith("---") or not ln.strip():
        continue
    cols = re.split(r"\s{2,}", ln.strip())
    if cols[0] == "Name":
        hdr = cols
        idx = [hdr.index(c) for c in ("Name", "Self CUDA", "# of Calls", "Input Shapes")]
        continue
    if hdr is None or len(cols) < len(hdr) - 1:
        continue
    name = cols[idx[0]]
    kernel = name.startswith(("Cijk", "Custom_Cijk", "void "))
    if kernel and "-k" not in sys.argv:
        continue
    try:
        t, calls = us(cols[idx[1]]), int(cols[idx[2]])
    except (AttributeError, ValueError, IndexError):
        continue
    shapes = cols[idx[3]] if len(cols) > idx[3] else ""
    print(f"{t / steps / 1e3:7.3f} ms/step {calls // steps:4d}x {t / calls:8.1f} us  {name[:44]:44s} {shapes[:90]}")
    n += 1
    if n >= n_rows:
        break
print(re.search(r"Self CUDA time total: [\d.]+\w+", txt).group(0), f"over {steps} steps")
