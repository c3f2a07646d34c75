#!/usr/bin/env python
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite) results.db into the per-kernel summary that is
committed under profiles/ -- same content as `rocprofv3 --stats` kernel_stats (name, calls, total,
average, min, max, percentage), plus VGPR/LDS/grid of the first dispatch.

    python tools/prof_summary.py gpurun_out/<tag>_prof/<tag>_results.db profiles/<tag>_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "min(vgpr_count), min(accum_vgpr_count), min(lds_size), min(grid_x), min(grid_y), min(workgroup_x) "
        "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage",
                    "VGPR", "AGPR", "LDS", "GridX", "GridY", "WorkgroupX"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", r[4], r[5], f"{100.0 * r[2] / total:.2f}", *r[6:]])
    print(f"{len(rows)} kernels, {sum(r[1] for r in rows)} dispatches, {total / 1e6:.2f} ms -> {out_path}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
