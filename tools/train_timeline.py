#!/usr/bin/env python3
"""Per-stream timeline summary of the LAST `k_adamw_hf`-delimited training step in a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace`): wall span, busy time per queue/stream, idle gaps on the busiest one, top kernels.

    python tools/train_timeline.py gpurun_out/x/prof_kd/..._results.db
"""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    print("# columns:", cols)
    rows = db.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
    ends = [i for i, r in enumerate(rows) if "k_adamw_hf" in r[0]]
    if len(ends) < 2:
        print("fewer than two optimizer steps in the trace")
        return
    a, b = ends[-2] + 1, ends[-1] + 1
    step = rows[a:b]
    t0, t1 = step[0][1], max(r[2] for r in step)
    print("step: %d kernels, wall %.3f ms" % (len(step), (t1 - t0) / 1e6))
    per_q = defaultdict(list)
    for n, s, e, q in step:
        per_q[q].append((s, e, n))
    for q, ks in sorted(per_q.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        busy = sum(e - s for s, e, _ in ks)
        gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
        pos = [g for g in gaps if g > 0]
        print("queue %s: %4d kernels, busy %.3f ms, span %.3f ms, positive gaps %d (sum %.3f ms, median %.2f us)" % (
            q, len(ks), busy / 1e6, (ks[-1][1] - ks[0][0]) / 1e6, len(pos), sum(pos) / 1e6,
            (sorted(pos)[len(pos) // 2] / 1e3) if pos else 0.0))
    prev_end = rows[ends[-2]][2]
    print("bubble between the previous optimizer kernel and this step's first kernel: %.3f ms" % ((t0 - prev_end) / 1e6))
    allk = sorted(step, key=lambda r: r[1])
    # device idle: time inside the step's wall span when no kernel of any queue is running
    idle, cur_end = 0, allk[0][1]
    biggest = []
    for n, s_, e_, q in allk:
        if s_ > cur_end:
            idle += s_ - cur_end
            biggest.append((s_ - cur_end, n[:60]))
        cur_end = max(cur_end, e_)
    print("device idle inside the step: %.3f ms" % (idle / 1e6))
    for g, n in sorted(biggest, reverse=True)[:12]:
        print("   idle %8.1f us before %s" % (g / 1e3, n))
    agg = defaultdict(lambda: [0, 0])
    for n, s, e, q in step:
        agg[n[:70]][0] += 1
        agg[n[:70]][1] += e - s
    print("top kernels of the step:")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print("  %-70s %4d %9.1f us  avg %7.1f" % (n, c, t / 1e3, t / 1e3 / c))


def dispatches(path):
    """Every kernel of the last step in start order: start offset, duration, queue, grid, name (one line each)."""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    rows = db.execute("select name, start, end, %s, grid_x, workgroup_x from kernels order by start" % qcol).fetchall()
    ends = [i for i, r in enumerate(rows) if "k_adamw_hf" in r[0]]
    if len(ends) < 2:
        return
    step = rows[ends[-2] + 1:ends[-1] + 1]
    t0 = step[0][1]
    for n, s, e, q, gx, wx in step:
        print("%9.1f us  +%7.1f us  q%-2s wg %5d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, (gx or 0) // max(wx or 1, 1),
                                                       n.replace("convdr::", "").replace("void ", "")[:90]))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--dispatches":
        dispatches(sys.argv[1])
    else:
        main(sys.argv[1])
