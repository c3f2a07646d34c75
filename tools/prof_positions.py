#!/usr/bin/env python
"""Per-window-position durations of the K1w launches in a rocprofv3 (rocpd sqlite) kernel trace of the decode loop:
launches are grouped by their order modulo (window x blocks), printed as mean microseconds per window position.

    python tools/prof_positions.py <results.db> [window=8] [blocks=13]
"""
import sqlite3
import sys


def main(db, window=8, blocks=13):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select start, end-start from kernels where name like '%gla_decode_window_kernel%' order by start").fetchall()
    d = [r[1] / 1e3 for r in rows]
    sd = sorted(d)
    pct = lambda q: sd[min(len(sd) - 1, int(q * len(sd)))]
    print("percentiles us: " + " ".join(f"p{int(q * 100)}={pct(q):.1f}" for q in (0.05, 0.25, 0.5, 0.75, 0.85, 0.9, 0.95, 0.99)))
    edges = [0, 10, 12, 14, 16, 18, 20, 25, 30, 40, 50, 1e9]
    print("histogram: " + " ".join(f"[{edges[i]:g},{edges[i + 1]:g}):{sum(edges[i] <= x < edges[i + 1] for x in d)}"
                                   for i in range(len(edges) - 1)))
    n = len(d) // (window * blocks) * (window * blocks)
    d = d[len(d) - n:]                      # the tail is the steady-state graph loop
    per = [[] for _ in range(window)]
    for i, x in enumerate(d):
        per[(i // blocks) % window].append(x)
    means = [sum(p) / max(len(p), 1) for p in per]
    k = means.index(max(means))             # the write position is the slowest: rotate it to the end
    means = means[k + 1:] + means[:k + 1]
    print(f"{len(d)} K1w launches; mean us per window position (write position last):",
          " ".join(f"{m:.1f}" for m in means), f"| all {sum(d) / len(d):.2f}")


if __name__ == "__main__":
    main(sys.argv[1], *[int(a) for a in sys.argv[2:]])
