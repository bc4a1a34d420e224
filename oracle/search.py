"""Oracle: exact inner-product top-k and the block-by-block merge (CPU, numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates:
  * ``faiss.IndexFlatIP(768)`` as used at
    /root/reference/drivers/run_convdr_inference.py:353 (.add :180, .search :182,
    .reset :202).  faiss-gpu is unpinned (requirements.txt:4) and absent from
    the container; its published contract for IndexFlatIP is: exact inner
    product of every query with every stored vector, the k best per query
    returned sorted by decreasing score, tie order unspecified.
  * ``search_one_by_one``  run_convdr_inference.py:157-242  (block loop, id map
    :190, two-way merge :206-229 with ``>=`` favouring the earlier block :218,
    and its quirk that the merged lists keep 2*topN entries once >= 2 blocks
    were seen, of which only the first topN are a valid ranking).
  * ``EvalDevQuery``       run_convdr_inference.py:21-113   (offset->pid :59,
    first-occurrence pid de-dup :61-69, TREC line ``qid Q0 pid rank 200-rank ance``
    :111-113).

Canonical score.  FAISS computes scores with an fp32 SGEMM whose summation order
is unspecified, so no bit-exact score exists to match.  This oracle (and the HIP
rescoring kernel, convdr_amd/csrc/ip_topk.hip) define ONE canonical value:

    vectors are zero-padded to a multiple of 256; lane l (0..63) owns the elements
    256 j + 4 l + c  (j = 0.., c = 0..3: one float4 per 1 KB line) and accumulates
    double(q[e]) * double(p[e]) sequentially in (j, c) order (the product of two fp32
    is exact in fp64, so each step is one rounding); then the 64 partials are folded
    by a butterfly  x[l] += x[l ^ 32], ^16, ^8, ^4, ^2, ^1.

Ranking is by (canonical fp64 score descending, index ascending); the reported
``D`` is that score rounded to fp32 (what FAISS hands back is fp32).  Any fp32
SGEMM result differs from it by <~1e-6 relative; the reference-run fixtures do
contain near-ties (smallest adjacent gap 3e-7), so against them ranks whose
reference scores differ by < 1e-3 are exchangeable (tests/helpers.py:
assert_topk_equivalent) and scores agree to 1e-3.
"""
import numpy as np


def canonical_scores(Q, P, chunk=256):
    """[nq, d] fp32 x [n, d] fp32 -> [nq, n] fp64 canonical inner products."""
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    P = np.ascontiguousarray(P, dtype=np.float32)
    nq, d = Q.shape
    n = P.shape[0]
    dp = (d + 255) // 256 * 256
    nj = dp // 256
    Q64 = np.zeros((nq, dp), np.float64); Q64[:, :d] = Q
    Q64 = Q64.reshape(nq, nj, 64, 4)
    out = np.empty((nq, n), np.float64)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        P64 = np.zeros((e - s, dp), np.float64); P64[:, :d] = P[s:e]
        P64 = P64.reshape(e - s, nj, 64, 4)
        acc = np.zeros((nq, e - s, 64), np.float64)
        for j in range(nj):
            for c in range(4):
                acc += Q64[:, None, j, :, c] * P64[None, :, j, :, c]
        w = 32
        while w >= 1:
            acc = acc[..., :w] + acc[..., w:2 * w]
            w //= 2
        out[:, s:e] = acc[..., 0]
    return out


def flat_ip_search(Q, P, k):
    """IndexFlatIP.search: (D fp32 [nq,k], I int64 [nq,k]); missing -> (-inf... , -1)
    FAISS pads with index -1 and score -3.4028235e38 when fewer than k vectors."""
    S = canonical_scores(Q, P)
    nq, n = S.shape
    D = np.full((nq, k), -3.4028234663852886e38, np.float32)
    I = np.full((nq, k), -1, np.int64)
    idx = np.arange(n)
    for qi in range(nq):
        order = np.lexsort((idx, -S[qi]))[:k]
        D[qi, :len(order)] = S[qi, order].astype(np.float32)
        I[qi, :len(order)] = order
    return D, I


class FlatIP:
    """add/search/reset stand-in with IndexFlatIP's call surface."""

    def __init__(self, d):
        self.d = d
        self._x = np.zeros((0, d), np.float32)

    @property
    def ntotal(self):
        return self._x.shape[0]

    def add(self, x):
        assert x.shape[1] == self.d
        self._x = np.concatenate([self._x, np.asarray(x, np.float32)], 0)

    def search(self, q, k):
        return flat_ip_search(q, self._x, k)

    def reset(self):
        self._x = np.zeros((0, self.d), np.float32)


def merge_two(merged, cur, topN):
    """run_convdr_inference.py:215-229 for one query; lists of (score, id)."""
    out, p1, p2 = [], 0, 0
    while p1 < topN and p2 < topN:
        if merged[p1][0] >= cur[p2][0]:
            out.append(merged[p1]); p1 += 1
        else:
            out.append(cur[p2]); p2 += 1
    while p1 < topN:
        out.append(merged[p1]); p1 += 1
    while p2 < topN:
        out.append(cur[p2]); p2 += 1
    return out


def search_one_by_one(blocks, Q, topN):
    """blocks: iterable of (emb fp32 [n,d], embid int64 [n]) in block order.
    Returns (merged_D float64, merged_I int64) shaped like the reference's
    ([nq, topN] for one block, [nq, 2*topN] afterwards)."""
    merged = None
    for emb, embid in blocks:
        D, I = flat_ip_search(Q, emb, topN)
        cand = [[(float(s), int(embid[i])) for s, i in zip(dr, ir)] for dr, ir in zip(D, I)]
        if merged is None:
            merged = cand
            continue
        merged = [merge_two(m, c, topN) for m, c in zip(merged, cand)]
    mD = np.array([[c[0] for c in row] for row in merged])
    mI = np.array([[c[1] for c in row] for row in merged])
    return mD, mI


def eval_dev_query_rows(query_embedding2id, merged_D, I_nearest_neighbor, topN, offset2pid):
    """Ranking part of EvalDevQuery (:37-69): per query the de-duplicated
    [(pid, score)] list padded with (0, 0) to topN, keyed by query id."""
    out = {}
    for qx in range(len(I_nearest_neighbor)):
        qid = query_embedding2id[qx]
        seen, rank = set(), 0
        if qid not in out:
            out[qid] = [(0, 0)] * topN
        for idx, score in zip(I_nearest_neighbor[qx][:topN], merged_D[qx][:topN].tolist()):
            pid = offset2pid[idx]
            if pid not in seen:
                out[qid][rank] = (pid, score)
                rank += 1
                seen.add(pid)
    return out


def trec_lines(rows, topN):
    """run_convdr_inference.py:111-113."""
    lines = []
    for qid, passages in rows.items():
        for i in range(topN):
            lines.append(str(qid) + " Q0 " + str(passages[i][0]) + " " + str(i + 1) + " " + str(-i - 1 + 200) + " ance\n")
    return lines
