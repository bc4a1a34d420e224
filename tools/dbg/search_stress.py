"""Repeat the randomised search cases of tests/test_ip_search_gpu.py many times in one process and describe any mismatch."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402

from oracle import search as OS  # noqa: E402
from convdr_amd.search import FlatIPIndex  # noqa: E402


def make(c):
    rs = np.random.RandomState(1000 + c)
    d = int(rs.choice([64, 768, 768, 40, 200]))
    n = int(rs.choice([1, 64, 300, 4097, 9000, 33000]))
    nq, k, kind = int(rs.choice([1, 3, 130])), int(rs.choice([1, 10, 100, 333])), c % 5
    P = rs.randn(n, d).astype(np.float32)
    if kind == 1:
        P[rs.randint(0, n, size=n // 2 + 1)] = P[rs.randint(0, n, size=n // 2 + 1)]
    elif kind == 2:
        P = (0.05 * P + rs.randn(1, d).astype(np.float32) * 3).astype(np.float32)
    elif kind == 3:
        P *= np.exp(rs.randn(n, 1) * 2).astype(np.float32)
    elif kind == 4:
        P = np.round(P * 2) / 2
    Q = rs.randn(nq, d).astype(np.float32)
    if kind == 4:
        Q = np.round(Q * 2) / 2
    cut = int(rs.randint(0, n + 1))
    return P, Q, k, cut, kind


def main(reps):
    cases = [make(c) for c in range(14)]
    refs = [OS.flat_ip_search(Q, P, k) for P, Q, k, cut, kind in cases]
    bad = 0
    for r in range(reps):
        for c, (P, Q, k, cut, kind) in enumerate(cases):
            n, d = P.shape
            idx = FlatIPIndex(d)
            for a, b in ((0, cut), (cut, n)):
                if b > a:
                    idx.add(P[a:b])
            D, I = idx.search(Q, k)
            Dr, Ir = refs[c]
            if not (np.array_equal(I, Ir) and np.array_equal(D, Dr)):
                bad += 1
                rows = np.nonzero((I != Ir).any(1) | (D != Dr).any(1))[0]
                print("rep %d case %d (n=%d nq=%d k=%d d=%d kind=%d cut=%d stats=%s): %d query rows differ: %s" % (
                    r, c, n, Q.shape[0], k, d, kind, cut, idx.stats, len(rows), rows[:6]), flush=True)
                q = rows[0]
                cols = np.nonzero((I[q] != Ir[q]) | (D[q] != Dr[q]))[0]
                print("  row %d: first differing ranks %s" % (q, cols[:8]))
                for j in cols[:4]:
                    print("   rank %d: got id %d score %r | want id %d score %r | exact(got) %.9g exact(want) %.9g" % (
                        j, I[q, j], D[q, j], Ir[q, j], Dr[q, j], float(P[I[q, j]].astype(np.float64) @ Q[q].astype(np.float64)) if I[q, j] >= 0 else float("nan"),
                        float(P[Ir[q, j]].astype(np.float64) @ Q[q].astype(np.float64))))
    print("mismatching (rep, case) pairs:", bad, "of", reps * len(cases))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
