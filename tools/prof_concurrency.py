#!/usr/bin/env python
"""From a rocprofv3 results.db (kernel trace): how much of the traced interval has 0 / 1 / >= 2 kernels in flight, per
stream (queue) totals, and the busy time -- to see whether work put on a second HIP stream really runs beside the first.
    python tools/prof_concurrency.py results.db [skip_first_fraction]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = cur.execute(f"select start, end, {qcol or 0}, name from kernels order by start").fetchall()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip
rows = [r for r in rows if r[0] >= t_lo]
ev = []
for s, e, q, n in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d
    last = t
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} kernels over {span / 1e6:.2f} ms; sum of durations {sum(e - s for s, e, _, _ in rows) / 1e6:.2f} ms")
for k in sorted(hist):
    print(f"  {k} kernels in flight: {hist[k] / 1e6:8.2f} ms = {100 * hist[k] / span:5.1f} %")
per_q = {}
for s, e, q, n in rows:
    a = per_q.setdefault(q, [0, 0])
    a[0] += 1
    a[1] += e - s
for q, (n, d) in sorted(per_q.items(), key=lambda kv: -kv[1][1]):
    print(f"  queue {q}: {n} kernels, {d / 1e6:.2f} ms")
