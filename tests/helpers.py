"""Shared checkers for the parity tests."""
import numpy as np


def assert_topk_equivalent(D_ref, I_ref, D, I, tol=1e-3, k=None):
    """Ranked lists must agree up to permutations inside runs of reference scores
    closer than ``tol`` (SURVEY.md §7 hard part 1: even two fp32 summation orders
    reorder such pairs; FAISS's tie order is unspecified).  Scores must agree to
    ``tol``.  The last run may be cut by the top-k boundary, so there only
    score agreement is required."""
    D_ref, I_ref, D, I = (np.asarray(x) for x in (D_ref, I_ref, D, I))
    k = k or D_ref.shape[1]
    assert D.shape[1] >= k and I.shape[1] >= k
    for q in range(D_ref.shape[0]):
        dr, ir, d, i = D_ref[q, :k], I_ref[q, :k], D[q, :k], I[q, :k]
        np.testing.assert_allclose(d, dr, rtol=0, atol=tol * max(1.0, np.abs(dr).max()),
                                   err_msg="scores of query %d" % q)
        start = 0
        while start < k:
            end = start + 1
            while end < k and abs(dr[end - 1] - dr[end]) < tol:
                end += 1
            if end < k:  # closed run: id sets must match
                assert sorted(ir[start:end].tolist()) == sorted(i[start:end].tolist()), \
                    "query %d ranks %d..%d: %s vs %s" % (q, start, end, ir[start:end], i[start:end])
            start = end


def cosine(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return (a * b).sum(-1) / np.sqrt((a * a).sum(-1) * (b * b).sum(-1))


# ---- measured margins -------------------------------------------------------------------------------------------
# Every tolerance in the GPU parity tests is meant to sit at ~3x the value measured on an MI355X.  `margin(name, value,
# bar)` asserts value <= bar (or >= for `higher=True`) and records the pair; the session writes them to
# gpurun_out/margins.json (tests/conftest.py), which is how the bars in the test files were set and are re-checked.
MARGINS = {}


def margin(name, value, bar, higher=False):
    value = float(value)
    MARGINS[name] = {"measured": value, "bar": float(bar), "higher_is_better": bool(higher)}
    if higher:
        assert value >= bar, "%s: measured %.6g, bar >= %.6g" % (name, value, bar)
    else:
        assert value <= bar, "%s: measured %.6g, bar <= %.6g" % (name, value, bar)
