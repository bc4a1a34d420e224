#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace --stats`) as a small
text table: per kernel launches / total / average / min / max duration (µs), registers, LDS.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_x.kernel_stats.txt
"""
import sqlite3
import sys


def main(path, skip=0):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# source: %s" % path)
    print("%-78s %7s %12s %10s %10s %10s %6s %5s %5s %7s %9s %5s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "grid", "wg"))
    for name, calls, tot, avg, mn, mx, vg, ag, sg, lds, grid, wg in rows:
        print("%-78s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %5d %7d %9d %5d" % (
            name[:78], calls, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, (vg or 0) + (ag or 0),
            sg or 0, lds or 0, grid or 0, wg or 0))


if __name__ == "__main__":
    main(sys.argv[1])
