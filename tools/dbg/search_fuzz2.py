"""Second fuzz axis: large k (up to 2048), many queries, many small add() calls, tiny / degenerate blocks."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402

from oracle import search as OS  # noqa: E402
from convdr_amd.search import FlatIPIndex  # noqa: E402

bad = 0
for c in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    rs = np.random.RandomState(5000 + c)
    d = int(rs.choice([64, 768, 320]))
    n = int(rs.choice([2, 100, 2049, 8191, 8193, 20000, 40000]))
    nq = int(rs.choice([1, 5, 128, 129, 700]))
    k = int(rs.choice([1, 2, 100, 1000, 2048]))
    kind = rs.randint(4)
    P = rs.randn(n, d).astype(np.float32)
    if kind == 1:
        P[:] = P[rs.randint(0, n, size=n) % max(1, n // 50)]          # ~50 copies of each distinct row: heavy ties
    elif kind == 2:
        P[n // 2:] = 0.0                                               # half the block is zero vectors
    elif kind == 3:
        P = (P * 1e-4).astype(np.float32)                              # tiny scores
    Q = rs.randn(nq, d).astype(np.float32)
    idx = FlatIPIndex(d)
    pieces = int(rs.choice([1, 2, 7]))
    cuts = sorted(set([0, n] + list(rs.randint(0, n + 1, size=pieces - 1))))
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            idx.add(P[a:b])
    try:
        D, I = idx.search(Q, k)
        Dr, Ir = OS.flat_ip_search(Q, P, k)
        ok = np.array_equal(I, Ir) and np.array_equal(D, Dr)
        msg = ""
    except Exception as e:  # noqa: BLE001
        ok, msg = False, "%s: %s" % (type(e).__name__, str(e)[:160])
    print("[%s] case %d: n=%d nq=%d k=%d d=%d kind=%d adds=%d stats=%s %s" % ("ok" if ok else "FAIL", c, n, nq, k, d, kind, len(cuts) - 1,
                                                                                idx.stats, msg), flush=True)
    bad += not ok
print("mismatches:", bad)
