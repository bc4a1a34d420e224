"""Randomised search cases against the oracle (GPU box): sizes, widths, duplicated rows (exact ties), scaled / shifted
clusters, tiny and huge k, several add() calls.  Prints the first mismatch.  Not part of the product."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402

from oracle import search as OS  # noqa: E402
from convdr_amd.search import FlatIPIndex  # noqa: E402


def main(seed0=0, cases=60):
    bad = 0
    for c in range(cases):
        rs = np.random.RandomState(seed0 + c)
        d = int(rs.choice([64, 128, 768, 768, 768, 40, 200, 1024]))
        n = int(rs.choice([1, 7, 64, 300, 4097, 9000, 33000, 70000]))
        nq = int(rs.choice([1, 3, 32, 130, 257]))
        k = int(rs.choice([1, 10, 100, 100, 333]))
        kind = rs.randint(7)
        P = rs.randn(n, d).astype(np.float32)
        if kind == 1:                      # exact duplicates -> ties
            src = rs.randint(0, n, size=n // 2 + 1)
            P[rs.randint(0, n, size=n // 2 + 1)] = P[src]
        elif kind == 2:                    # a dominant common component (encoder-like)
            P = (0.05 * P + rs.randn(1, d).astype(np.float32) * 3).astype(np.float32)
        elif kind == 3:                    # wildly different norms
            P *= np.exp(rs.randn(n, 1) * 2).astype(np.float32)
        elif kind == 4:                    # low-precision values: many exact score ties
            P = np.round(P * 2) / 2
        Q = rs.randn(nq, d).astype(np.float32)
        if kind == 4:
            Q = np.round(Q * 2) / 2
        elif kind == 5:                    # any magnitude: the fp16 scan copy / queries are moved by powers of two
            P = (P * np.float32(10.0 ** rs.uniform(-18, 10))).astype(np.float32)
            Q = (Q * np.float32(10.0 ** rs.uniform(-8, 8))).astype(np.float32)
        elif kind == 6:                    # later rows far longer than the first ones: CONVDR_IP_RANGE -> rebuilt copy
            P[n // 2:] *= np.float32(10.0 ** rs.uniform(1, 4))
        idx = FlatIPIndex(d)
        cuts = sorted(set([0, n] + list(rs.randint(0, n + 1, size=rs.randint(0, 3)))))
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                idx.add(P[a:b])
        try:
            D, I = idx.search(Q, k)
            Dr, Ir = OS.flat_ip_search(Q, P, k)
            ok = np.array_equal(I, Ir) and np.array_equal(D, Dr)
        except Exception as e:  # noqa: BLE001
            ok = False
            print("  exception:", type(e).__name__, str(e)[:200])
        print("[%s] case %d: n=%d nq=%d k=%d d=%d kind=%d adds=%d stats=%s" % ("ok" if ok else "FAIL", c, n, nq, k, d, kind,
                                                                                len(cuts) - 1, idx.stats), flush=True)
        bad += not ok
    print("mismatches:", bad)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60)
