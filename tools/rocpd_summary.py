#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace --stats`) as a small
text table: per kernel launches / total / average / min / max duration (µs), registers, LDS.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--skip-fraction 0.3333] > profiles/r01_x.kernel_stats.txt
"""
import sqlite3
import sys


def main(path, skip_fraction=0.0):
    """skip_fraction: drop that share of every kernel's dispatches, earliest first -- the warm-up steps of a
    `bench.py --steps K --warmup W` run are W / (K + W) of a per-step kernel's dispatches (the committed summary
    averaged the warm-up dispatches in, so bench.py's live `frac`, which drops them, did not follow from it)."""
    db = sqlite3.connect(path)
    print("# source: %s" % path)
    if skip_fraction > 0:
        print("# the first %.4f of every kernel's dispatches (by start time: the warm-up steps) are excluded" % skip_fraction)
        per = {}
        for name, start, dur, vg, ag, sg, lds, grid, wg in db.execute(
                "select name, start, duration, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x from kernels order by start"):
            per.setdefault(name, []).append((dur, vg, ag, sg, lds, grid, wg))
        rows = []
        for name, ds in per.items():
            ds = ds[int(len(ds) * skip_fraction):] or ds
            dur = [d[0] for d in ds]
            rows.append((name, len(ds), sum(dur), sum(dur) / len(ds), min(dur), max(dur)) + tuple(max((d[i] or 0) for d in ds) for i in range(1, 7)))
        rows.sort(key=lambda r: -r[2])
    else:
        rows = db.execute(
            "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
            "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
            "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-78s %7s %12s %10s %10s %10s %6s %5s %5s %7s %9s %5s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "grid", "wg"))
    for name, calls, tot, avg, mn, mx, vg, ag, sg, lds, grid, wg in rows:
        print("%-78s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %5d %7d %9d %5d" % (
            name[:78], calls, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, (vg or 0) + (ag or 0),
            sg or 0, lds or 0, grid or 0, wg or 0))


if __name__ == "__main__":
    frac = 0.0
    args = [a for a in sys.argv[1:]]
    if "--skip-fraction" in args:
        i = args.index("--skip-fraction")
        frac = float(args[i + 1])
        del args[i:i + 2]
    main(args[0], frac)
