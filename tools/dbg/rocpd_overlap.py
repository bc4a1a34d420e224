"""Do consecutive dispatches of a rocprofv3 kernel trace overlap in time?  (start of kernel i+1 vs end of kernel i)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, start, end from kernels order by start").fetchall()
tot_d = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
ov = [(rows[i + 1][1] - rows[i][2]) for i in range(len(rows) - 1)]
neg = [x for x in ov if x < 0]
print("dispatches %d  sum of durations %.3f ms  first-start..last-end %.3f ms" % (len(rows), tot_d / 1e6, span / 1e6))
print("gaps start[i+1] - end[i]: %d negative (overlap), median %.2f us, min %.2f us, sum of overlaps %.3f ms" % (
    len(neg), sorted(ov)[len(ov) // 2] / 1e3, min(ov) / 1e3, -sum(neg) / 1e6))
big = [(rows[i][0][:50], rows[i + 1][0][:50], ov[i] / 1e3) for i in range(len(ov)) if ov[i] < -20000][:8]
for b in big:
    print("  %-50s -> %-50s gap %.1f us" % b)
import collections
d = collections.defaultdict(list)
for n, s, e in rows:
    d[n[:60]].append((e - s) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:7]:
    v = sorted(v)
    print("  %-60s n=%3d  median %.1f us  min %.1f  max %.1f" % (n, len(v), v[len(v) // 2], v[0], v[-1]))
