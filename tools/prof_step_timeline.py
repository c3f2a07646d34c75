#!/usr/bin/env python
"""Timeline of ONE decode step from a rocprofv3 (rocpd sqlite) kernel trace of the bench loop: the trace's tail is the
steady-state graph loop, periodic with `period` launches per token; prints, per position in the step, the kernel, its mean
/ min duration and the mean gap to the next launch's start (start[i+1] - end[i]).

    python tools/prof_step_timeline.py <results.db> [out.csv]
"""
import csv, sqlite3, sys


def short(n):
    n = n.replace("void lina::", "").split("(")[0]
    return n[:70]


def main(db, out=None):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    names = [r[0] for r in rows]
    # period = distance between consecutive greedy_pick_embed launches in the tail
    picks = [i for i, n in enumerate(names) if "greedy_pick_embed" in n]
    period = picks[-1] - picks[-2]
    n_steps = 0
    i = len(picks) - 1
    while i > 0 and picks[i] - picks[i - 1] == period:
        n_steps += 1
        i -= 1
    n_steps = min(n_steps, 600)
    end_idx = picks[-1] + 1
    beg_idx = end_idx - n_steps * period
    dur = [[] for _ in range(period)]
    gap = [[] for _ in range(period)]
    for s in range(n_steps):
        for p in range(period):
            k = beg_idx + s * period + p
            dur[p].append(rows[k][2] - rows[k][1])
            if k + 1 < len(rows):
                gap[p].append(rows[k + 1][1] - rows[k][2])
    tot = 0.0
    table = []
    for p in range(period):
        d = sum(dur[p]) / len(dur[p]) / 1e3
        g = sum(gap[p]) / max(len(gap[p]), 1) / 1e3
        tot += d + g
        table.append((p, short(names[beg_idx + p]), d, min(dur[p]) / 1e3, g))
    print(f"{n_steps} steps of {period} launches; sum(mean duration + mean gap) = {tot:.1f} us per step")
    for p, n, d, mn, g in table:
        print(f"{p:3d} {n:70s} {d:7.2f} (min {mn:6.2f})  gap {g:6.2f}")
    if out:
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["position", "kernel", "mean_us", "min_us", "gap_to_next_us"])
            for r in table:
                w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.3f}"])


if __name__ == "__main__":
    main(*sys.argv[1:])
