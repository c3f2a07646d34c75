#!/usr/bin/env python
"""One kernel of a rocprofv3 (rocpd sqlite) trace broken out by launch grid: a process that runs the same kernel at several
problem sizes (K1w at 64 / 128 / 256 / 512 rows in bench.py) pools them in `--stats`; the bench line's in-situ figure is for ONE
of them.    python tools/prof_by_grid.py <results.db> <name substring> [out.txt]"""
import sqlite3
import sys

db, pat = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, grid_x, grid_y, workgroup_x, count(*), avg(end-start), min(end-start), max(end-start) from kernels "
                   "where name like ? group by name, grid_x, grid_y, workgroup_x order by 1, 2", (f"%{pat}%",)).fetchall()
lines = [f"{'kernel':78s} {'workgroups':>10s} {'launches':>9s} {'mean us':>9s} {'min us':>8s} {'max us':>8s}"]
for name, gx, gy, wx, n, avg, mn, mx in rows:
    short = name.replace("void lina::", "").split("(")[0][:78]
    lines.append(f"{short:78s} {gx // max(wx, 1) * max(gy, 1):10d} {n:9d} {avg / 1e3:9.2f} {mn / 1e3:8.2f} {mx / 1e3:8.2f}")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
