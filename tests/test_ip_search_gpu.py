"""Parity of the HIP inner-product top-k (through the C ABI) with the CPU oracle and
with the reference-run fixtures.  GPU only; `tests/test_capi_cpu.py` covers loading."""
import os
import pickle

import numpy as np
import pytest

from oracle import search as OS
from tests.helpers import assert_topk_equivalent
from tests.golden.make_golden import synth_corpus

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _index(d=768, **kw):
    from convdr_amd.search import FlatIPIndex
    return FlatIPIndex(d, **kw)


@pytest.mark.parametrize("n,nq,k,d", [
    (700, 16, 100, 768),      # n <= cap: every passage is a candidate
    (5000, 37, 100, 768),     # full-score threshold pass
    (5000, 5, 10, 64),        # small d (one k-step), small k
    (40000, 24, 100, 768),    # sampled threshold pass, ragged last tile
    (33000, 130, 7, 128),     # two query tiles, ragged both ways
    (5000, 4, 100, 72),       # width that is not a multiple of the 64-wide K step (zero columns inside the index)
    (3000, 3, 50, 32),
])
def test_search_matches_oracle_bit_exact(torch_cuda, n, nq, k, d):
    P, Q = synth_corpus(100 + n % 97, n, d), synth_corpus(7, nq, d)
    idx = _index(d)
    idx.add(P)
    D, I = idx.search(Q, k)
    Dr, Ir = OS.flat_ip_search(Q, P, k)
    np.testing.assert_array_equal(I, Ir)          # integer indexing: bit exact
    np.testing.assert_array_equal(D, Dr)          # canonical fp64 score rounded to fp32: bit exact
    assert D.dtype == np.float32 and I.dtype == np.int64


@pytest.mark.parametrize("precision", ["fp16", "fp16x3", "bf16", "bf16x3"])
@pytest.mark.parametrize("n,nq,k,d", [(700, 16, 100, 768), (40000, 140, 100, 768), (33000, 130, 7, 128)])
def test_every_pinned_rung_returns_the_same_exact_result(torch_cuda, precision, n, nq, k, d):
    """The rung (fp16 / split fp16 / bf16 / split bf16 MFMA scan) only decides which candidates are re-scored: D and I
    are defined on the canonical fp64 scores and must not depend on it."""
    P, Q = synth_corpus(200 + n % 89, n, d), synth_corpus(8, nq, d)
    idx = _index(d, precision=precision)
    idx.add(P)
    D, I = idx.search(Q, k)
    Dr, Ir = OS.flat_ip_search(Q, P, k)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D, Dr)


@pytest.mark.parametrize("scale_p,scale_q", [(1e-6, 1.0), (3e4, 1e-5), (1.0, 1e6), (1e-20, 1e-12), (1e12, 1e10)])
def test_fp16_rung_is_scale_free(torch_cuda, scale_p, scale_q):
    """Halfs span 6e-8 .. 65504: the scan copy and every query are moved to norms ~2^12 by exact powers of two, so blocks
    and queries of any magnitude certify on the first pass like unit-scale ones."""
    P = (synth_corpus(71, 20000, 768) * np.float32(scale_p)).astype(np.float32)
    Q = (synth_corpus(72, 12, 768) * np.float32(scale_q)).astype(np.float32)
    idx = _index(768, precision="fp16")
    idx.add(P)
    D, I = idx.search(Q, 100)
    assert idx.stats["retried"] <= 1 and not idx.stats.get("exhaustive_queries"), idx.stats
    Dr, Ir = OS.flat_ip_search(Q, P, 100)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D, Dr)


def test_fp16_scan_copy_is_rebuilt_when_later_rows_outgrow_its_scale(torch_cuda):
    """The power-of-two scale of the fp16 copy is fixed by the first rows an index sees.  Rows added later that are > 7x
    longer could round to inf: the cut kernel reports CONVDR_IP_RANGE instead of a result, the index re-derives the scale
    from the block's max norm, rebuilds the copy and searches again."""
    torch = torch_cuda
    P0, P1 = synth_corpus(81, 6000, 768), synth_corpus(82, 5000, 768) * np.float32(300.0)
    Q = synth_corpus(83, 9, 768)
    idx = _index(768)
    idx.add(P0)
    s0 = idx._scale
    idx.add(P1)
    st = idx.search_device(torch.from_numpy(Q).cuda(), 20)[2]
    assert (st.cpu().numpy() == 4).all()
    D, I = idx.search(Q, 20)
    assert idx.stats["rescaled"] == 1 and idx._scale < s0
    P = np.concatenate([P0, P1])
    Dr, Ir = OS.flat_ip_search(Q, P, 20)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D, Dr)
    D, I = idx.search(Q, 20)
    assert idx.stats["rescaled"] == 0
    np.testing.assert_array_equal(I, Ir)


def test_reserved_index_fills_in_place(torch_cuda):
    """reserve(n): the resident block is allocated once and add() fills it slice by slice -- no re-allocation (the
    unreserved FAISS-style append copies the whole block on every add: impossible for a 117 GB corpus)."""
    torch = torch_cuda
    P, Q = synth_corpus(91, 9000, 768), synth_corpus(92, 7, 768)
    idx = _index(768)
    idx.reserve(9000)
    base = (idx._s32.data_ptr(), idx._s16.data_ptr())
    for a in range(0, 9000, 2500):
        idx.add(torch.from_numpy(P[a:a + 2500]).cuda())
        assert (idx._s32.data_ptr(), idx._s16.data_ptr()) == base
    assert idx.ntotal == 9000 and torch.equal(idx._p32.cpu(), torch.from_numpy(P))
    D, I = idx.search(Q, 100)
    Dr, Ir = OS.flat_ip_search(Q, P, 100)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D, Dr)
    idx.reset()                                  # faiss reset keeps the reservation: next block, same memory
    assert idx.ntotal == 0
    idx.add(P[:3000])
    assert idx._s32.data_ptr() == base[0]
    D, I = idx.search(Q, 10)
    np.testing.assert_array_equal(I, OS.flat_ip_search(Q, P[:3000], 10)[1])


def test_exact_ties_lower_index_first(torch_cuda):
    base = synth_corpus(21, 300, 768)
    P = np.concatenate([base[:200], base[50:60], base[50:60]])
    Q = base[48:56] + 0.0
    idx = _index()
    idx.add(P)
    D, I = idx.search(Q, 10)
    Dr, Ir = OS.flat_ip_search(Q, P, 10)
    np.testing.assert_array_equal(I, Ir)
    for j, r in enumerate(range(48, 56)):
        if r >= 50:
            assert I[j, :3].tolist() == [r, 200 + r - 50, 210 + r - 50]


def test_tie_group_straddles_k(torch_cuda):
    """A block of identical passages ties exactly across the k-th place (k_ip_select keeps the whole boundary group and
    orders it by index), both when every passage is a candidate and behind the threshold scan."""
    base = synth_corpus(33, 6000, 768)
    for n in (1500, 6000):
        P = base[:n].copy()
        P[100:160] = P[100]                    # 60 identical rows
        Q = np.stack([P[100] * 3.0, P[100] * 3.0 + base[7] * 0.01, base[5]]).astype(np.float32)
        idx = _index()
        idx.add(P)
        for k in (10, 37, 100):
            D, I = idx.search(Q, k)
            Dr, Ir = OS.flat_ip_search(Q, P, k)
            np.testing.assert_array_equal(I, Ir)
            np.testing.assert_array_equal(D, Dr)
        assert I[0, :60].tolist() == list(range(100, 160))


def test_all_scores_equal(torch_cuda):
    P = np.repeat(synth_corpus(34, 1, 768), 2000, axis=0)
    Q = synth_corpus(35, 4, 768)
    idx = _index()
    idx.add(P)
    D, I = idx.search(Q, 100)
    assert (I == np.arange(100)[None, :]).all()
    Dr, Ir = OS.flat_ip_search(Q, P, 100)
    np.testing.assert_array_equal(D, Dr)


def test_fewer_passages_than_k_and_empty(torch_cuda):
    P, Q = synth_corpus(3, 37, 768), synth_corpus(4, 3, 768)
    idx = _index()
    idx.add(P)
    D, I = idx.search(Q, 100)
    Dr, Ir = OS.flat_ip_search(Q, P, 100)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D, Dr)
    assert (I[:, 37:] == -1).all()
    idx.reset()
    assert idx.ntotal == 0
    D, I = idx.search(Q, 5)
    assert (I == -1).all()


def test_add_appends_like_faiss(torch_cuda):
    P, Q = synth_corpus(5, 3000, 768), synth_corpus(6, 9, 768)
    idx = _index()
    idx.add(P[:1200]); idx.add(P[1200:])
    assert idx.ntotal == 3000
    D, I = idx.search(Q, 50)
    Dr, Ir = OS.flat_ip_search(Q, P, 50)
    np.testing.assert_array_equal(I, Ir)


def test_retry_path_is_exact(torch_cuda):
    """Force the certificate to fail (tiny rank target -> tau above the k-th score) and a
    candidate overflow (huge rank target on a tight capacity): the host loop must still
    return the exact answer."""
    P, Q = synth_corpus(8, 20000, 768), synth_corpus(9, 12, 768)
    Dr, Ir = OS.flat_ip_search(Q, P, 100)
    idx = _index(rank_target=100, cap=1024)       # tau ~ the 100th score: eps margin cannot hold
    idx.add(P)
    D, I = idx.search(Q, 100)
    assert idx.stats["retried"] > 0
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D, Dr)


def test_search_one_by_one_matches_reference_fixture(torch_cuda, golden_dir, tmp_path):
    from convdr_amd.search import search_one_by_one
    z = np.load(os.path.join(golden_dir, "search.npz"))
    for r, (n, s) in enumerate(zip(z["a/sizes"], z["a/seeds"])):
        for pre, arr in (("passage__emb_p_", synth_corpus(int(s), int(n), 768)),
                         ("passage__embid_p_", np.arange(int(n), dtype=np.int64) * 3 + r)):
            with open(tmp_path / ("%s_data_obj_%d.pb" % (pre, r)), "wb") as h:
                pickle.dump(arr, h, protocol=4)
    Q = synth_corpus(int(z["a/qseed"]), 16, 768)
    topN = int(z["a/topN"])
    mD, mI = search_one_by_one(str(tmp_path), _index(), Q, topN)
    assert mD.shape == z["a/merged_D"].shape and mD.dtype == np.float64 and mI.dtype == np.int64
    assert_topk_equivalent(z["a/merged_D"], z["a/merged_I"], mD, mI, k=topN)
    # and bit-exact against the oracle restatement of the same function
    blocks = [(synth_corpus(int(s), int(n), 768), np.arange(int(n), dtype=np.int64) * 3 + r)
              for r, (n, s) in enumerate(zip(z["a/sizes"], z["a/seeds"]))]
    oD, oI = OS.search_one_by_one(blocks, Q, topN)
    np.testing.assert_array_equal(mI, oI)
    np.testing.assert_array_equal(mD, oD)


def test_device_merge_equals_reference_pointer_walk(torch_cuda):
    """convdr_topk_merge against the oracle's restatement of the reference's two-pointer merge, on lists full of ties
    (duplicated scores inside and across the two lists), ragged widths and -FLT_MAX / -1 padding."""
    torch = torch_cuda
    from convdr_amd.search import merge_topk_device
    rs = np.random.RandomState(3)
    for na, nb, nq in ((100, 100, 37), (7, 100, 5), (1, 1, 3), (256, 300, 11)):
        def lists(n):
            d = np.sort(rs.randint(0, 40, size=(nq, n)).astype(np.float32) * 0.25, axis=1)[:, ::-1].copy()
            d[:, n - n // 5:] = -3.4028234663852886e38      # FAISS-style padding tail
            i = rs.randint(0, 10 ** 9, size=(nq, n)).astype(np.int64)
            i[:, n - n // 5:] = -1
            return d, i
        (Da, Ia), (Db, Ib) = lists(na), lists(nb)
        Do, Io = merge_topk_device((torch.from_numpy(Da).cuda(), torch.from_numpy(Ia).cuda()),
                                   (torch.from_numpy(Db).cuda(), torch.from_numpy(Ib).cuda()), max(na, nb))
        Do, Io = Do.cpu().numpy(), Io.cpu().numpy()
        for q in range(nq):
            # the reference walk needs equal lengths (both lists are topN long there): pad the shorter with -inf
            # entries, which the walk leaves at the tail, and compare the real prefix
            ref, p1, p2 = [], 0, 0
            A, B = list(zip(Da[q], Ia[q])), list(zip(Db[q], Ib[q]))
            while p1 < na and p2 < nb:
                if A[p1][0] >= B[p2][0]:
                    ref.append(A[p1]); p1 += 1
                else:
                    ref.append(B[p2]); p2 += 1
            ref += A[p1:] + B[p2:]
            assert [float(x) for x in Do[q]] == [float(s) for s, _ in ref]
            assert [int(x) for x in Io[q]] == [int(i) for _, i in ref]


@pytest.mark.parametrize("n", [1_000_000, 4_750_000])
def test_full_size_properties_1m_x_1k(torch_cuda, n):
    """BASELINE configs[1] size (1M x 768) and the configs[3] per-GPU shard (38M / 8 = 4.75M x 768), 1k queries, k = 100:
    size-independent properties, checked with an independent fp32 GEMM (torch/rocBLAS) on the same device."""
    torch = torch_cuda
    nq, k, d = 1000, 100, 768
    g = torch.Generator(device="cuda").manual_seed(0)
    P = torch.randn(n, d, device="cuda", generator=g)
    Q = torch.randn(nq, d, device="cuda", generator=g)
    # planted needles: passage 1000*q+17 is a copy of query q (score |q|^2 ~ 768 >> 5 sigma) -> must be rank 1
    needles = torch.arange(nq, device="cuda") * (n // nq) + 17
    P[needles] = Q
    idx = _index()
    idx.add(P)
    D, I = idx.search(Q, k)
    assert idx.stats["retried"] == 0, idx.stats
    Dt, It = torch.from_numpy(D).cuda(), torch.from_numpy(I).cuda()
    assert (It[:, 0] == needles).all()
    assert (Dt[:, :-1] >= Dt[:, 1:]).all()                                   # sorted
    assert ((It >= 0) & (It < n)).all()
    assert all(len(set(row)) == k for row in I[::50].tolist())               # no duplicates
    # scores are the true inner products of the returned ids
    exact = torch.einsum("qd,qkd->qk", Q.double(), P[It.reshape(-1)].reshape(nq, k, d).double())
    assert (exact.float() - Dt).abs().max().item() == 0.0 or \
        torch.allclose(exact.float(), Dt, rtol=0, atol=1e-4)
    # nothing outside the returned set beats the k-th score (independent fp32 GEMM, chunked)
    kth = Dt[:, -1:].clone()
    above = torch.zeros(nq, dtype=torch.int64, device="cuda")
    for s in range(0, n, 125_000):
        S = Q @ P[s:s + 125_000].T
        above += (S > kth + 5e-3).sum(1)
    assert (above <= k - 1).all(), above.max().item()


def test_cast_scale_38m_on_one_gpu(torch_cuda):
    """BASELINE's target corpus (README.md:152: 38M CAsT passages; configs[3] spreads it over 8 GPUs because 16-32 GB
    parts had to) resident on ONE MI355X: 38M x 768 fp32 (117 GB) + the fp16 scan copy (58 GB) in reserved storage
    filled from 8 generated slices of 4.75M (no re-allocation), 1k queries, k = 100.  Same size-independent properties
    as at 1M: planted needles first, sorted, no duplicates, scores = the fp64 re-dot of the returned rows, and an
    independent fp32 GEMM (rocBLAS through torch, chunked) finds fewer than k passages above the k-th score."""
    torch = torch_cuda
    torch.cuda.empty_cache()
    n_slice, slices, nq, k, d = 4_750_000, 8, 1000, 100, 768
    n = n_slice * slices
    free, _ = torch.cuda.mem_get_info()
    if free < n * d * 6 + (24 << 30):
        pytest.skip("needs %.0f GB of free HBM (%.0f free)" % ((n * d * 6 + (24 << 30)) / 1e9, free / 1e9))
    idx = _index()
    idx.reserve(n)
    base = idx._s32.data_ptr()
    Q = torch.randn(nq, d, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1234))
    needles = torch.arange(nq, device="cuda") * (n // nq) + 17
    for s in range(slices):
        P = torch.randn(n_slice, d, device="cuda", generator=torch.Generator(device="cuda").manual_seed(100 + s))
        sel = (needles >= s * n_slice) & (needles < (s + 1) * n_slice)
        P[needles[sel] - s * n_slice] = Q[sel]
        idx.add(P)
        del P
    assert idx.ntotal == n and idx._s32.data_ptr() == base
    D, I = idx.search_tensors(Q, k)
    assert idx.stats["retried"] == 0 and not idx.stats.get("exhaustive_queries"), idx.stats
    assert (I[:, 0] == needles).all()
    assert (D[:, :-1] >= D[:, 1:]).all()
    assert ((I >= 0) & (I < n)).all()
    assert all(len(set(row)) == k for row in I[::50].tolist())
    P32 = idx._p32
    exact = torch.einsum("qd,qkd->qk", Q.double(), P32[I.reshape(-1)].reshape(nq, k, d).double())
    assert torch.allclose(exact.float(), D, rtol=0, atol=1e-4)
    kth = D[:, -1:].clone()
    above = torch.zeros(nq, dtype=torch.int64, device="cuda")
    for s in range(0, n, 250_000):
        above += ((Q @ P32[s:s + 250_000].T) > kth + 5e-3).sum(1)
    assert (above <= k - 1).all(), above.max().item()
    # the HBM-bound regime (CAsT has a few hundred queries): 100 queries stream the 58 GB scan copy once
    D1, I1 = idx.search_tensors(Q[:100].contiguous(), k)
    assert torch.equal(I1, I[:100]) and torch.equal(D1, D[:100])
    idx.release()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision", ["auto", "fp16", "bf16x3"])
def test_clustered_embeddings_are_searched_exactly(torch_cuda, precision):
    """Encoder outputs share a large common component (cosine ~0.9 between passages), which puts thousands of scores
    inside the bf16 error band of the k-th one: centring + the split-bf16 rung must still return the exact top-k."""
    rs = np.random.RandomState(0)
    n, nq, d, k = 30000, 12, 768, 50
    c = rs.randn(d).astype(np.float32)
    P = (0.9 * c[None, :] + 0.12 * rs.randn(n, d)).astype(np.float32)
    Q = (0.9 * c[None, :] + 0.12 * rs.randn(nq, d)).astype(np.float32)
    Dr, Ir = OS.flat_ip_search(Q, P, k)
    idx = _index(d, precision=precision)
    idx.add(P)
    D, I = idx.search(Q, k)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D, Dr)
    raw = _index(d, precision="bf16", center=False)       # the un-centred bf16 rung cannot certify this data:
    raw.add(P)                                            # it falls through to the exhaustive rung and still answers
    D2, I2 = raw.search(Q, k)
    assert raw.stats.get("exhaustive_queries", 0) > 0, raw.stats
    np.testing.assert_array_equal(I2, Ir)
    np.testing.assert_array_equal(D2, Dr)


def test_sharded_search_exchange_over_rccl(torch_cuda):
    """The N > 1 step of bench.py on one GPU: a 1-rank RCCL group, the collectives forced (all-gather of the query slices,
    local exact top-k, all-gather of scores / record offsets, device merge) -- same tensors, dtypes and streams as with
    8 ranks; the result must equal the plain local search mapped through embid."""
    import os
    import torch
    import torch.distributed as dist
    from convdr_amd import parallel
    P, Q = synth_corpus(41, 30000, 768), synth_corpus(42, 37, 768)
    idx = _index()
    idx.add(P)
    dev = idx.device
    embid = torch.arange(5, 5 + 3 * 30000, 3, device=dev, dtype=torch.int64)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29678")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        Qd = torch.from_numpy(Q).to(dev)
        Qall = parallel.all_gather_rows(Qd, force=True)[:37]
        D, I, st = parallel.search_sharded_device(idx, Qall, 100, embid, force=True)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    Dr, Ir = OS.flat_ip_search(Q, P, 100)
    assert int(st.sum()) == 0
    np.testing.assert_array_equal(I.cpu().numpy(), 5 + 3 * Ir)
    np.testing.assert_array_equal(D.cpu().numpy(), Dr)


def test_rank_merge_on_device_equals_stable_sort(torch_cuda):
    """parallel.merge_rank_topk on the GPU (chain of convdr_topk_merge calls, what 8 ranks' lists go through) against the
    stable-sort definition the gloo tests use -- lists full of ties inside and across ranks, -1 / -FLT_MAX padding."""
    torch = torch_cuda
    from convdr_amd import parallel
    rs = np.random.RandomState(9)
    for W, nq, k in ((8, 33, 100), (2, 5, 10), (3, 7, 1), (8, 4, 256)):
        D = np.sort(rs.randint(0, 60, size=(W, nq, k)).astype(np.float32) * 0.5, axis=2)[:, :, ::-1].copy()
        I = rs.randint(0, 10 ** 12, size=(W, nq, k)).astype(np.int64)
        D[W - 1, :, k - k // 4:] = -3.4028234663852886e38
        I[W - 1, :, k - k // 4:] = -1
        Dc, Ic = parallel.merge_rank_topk(torch.from_numpy(D), torch.from_numpy(I), k)
        Dg, Ig = parallel.merge_rank_topk(torch.from_numpy(D).cuda(), torch.from_numpy(I).cuda(), k)
        np.testing.assert_array_equal(Dg.cpu().numpy(), Dc.numpy())
        np.testing.assert_array_equal(Ig.cpu().numpy(), Ic.numpy())


def test_streamed_host_block_equals_resident_block(torch_cuda, tmp_path):
    """FlatIPIndex.add of a HOST array (the mmap of a block file) goes through pinned, double-buffered chunks with the
    bf16 preparation of chunk i under the copy of chunk i + 1; results must be bit-identical to adding the same block from
    device memory -- including across several chunks with a ragged tail, for an appended second block, and through
    search_one_by_one reading BlockView mappings."""
    torch = torch_cuda
    from convdr_amd import blocks
    from convdr_amd.search import search_one_by_one
    P0, P1, Q = synth_corpus(51, 30011, 768), synth_corpus(52, 7001, 768), synth_corpus(53, 23, 768)
    ref = _index()
    ref.add(torch.from_numpy(P0).cuda())
    ref.add(torch.from_numpy(P1).cuda())
    Dr, Ir = ref.search(Q, 100)
    for r, P in enumerate((P0, P1)):
        blocks.dump_block(str(tmp_path / ("passage__emb_p__data_obj_%d.pb" % r)), P)
    idx = _index()
    idx.host_chunk_bytes = 5 << 20                           # 1706 rows per chunk: 18 chunks + a ragged tail
    views = [blocks.BlockView(str(tmp_path / ("passage__emb_p__data_obj_%d.pb" % r))) for r in range(2)]
    for v in views:
        assert isinstance(v.array, np.ndarray) and not v.array.flags.writeable      # the mmap, not a copy
        idx.add(v.array)
    D, I = idx.search(Q, 100)
    torch.cuda.synchronize()
    for v in views:
        v.close()
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D, Dr)
    # the driver function over the same files (embid = identity + 10^6 r)
    for r, P in enumerate((P0, P1)):
        blocks.dump_block(str(tmp_path / ("passage__embid_p__data_obj_%d.pb" % r)), np.arange(len(P), dtype=np.int64) + 10 ** 6 * r)
    idx2 = _index()
    idx2.host_chunk_bytes = 5 << 20
    mD, mI = search_one_by_one(str(tmp_path), idx2, Q, 100)
    oD, oI = OS.search_one_by_one([(P0, np.arange(len(P0), dtype=np.int64)), (P1, np.arange(len(P1), dtype=np.int64) + 10 ** 6)], Q, 100)
    np.testing.assert_array_equal(mI, oI)
    np.testing.assert_array_equal(mD, oD)


@pytest.mark.parametrize("n", [33000, 47104, 70000, 150000])
def test_mid_size_blocks_get_a_threshold(torch_cuda, n):
    """Blocks between the exact-threshold range (<= 32 k passages) and the 1/32-sample range: the sampled threshold must
    exist (round 2 found n = 47,104 asking for rank 1,113 of a 1,024-value sample: no threshold, every passage emitted for
    every query, certification only through the overflow retry).  The two-best-of-64 sample can only aim at rank n / 128
    here, so the first pass may come back UNCERTAIN with a retry threshold -- one more round, never an overflow -- and the
    result stays bit-exact."""
    P, Q = synth_corpus(61, n, 768), synth_corpus(62, 40, 768)
    idx = _index()
    idx.add(P)
    D, I, st, _ = idx.search_device(torch_cuda.from_numpy(Q).cuda(), 100)
    emitted, band = idx.last_counts(40, 100)
    assert 100 <= int(emitted.min()) and int(emitted.max()) < 4096, (int(emitted.min()), int(emitted.max()))
    assert int((st == 1).sum()) == 0                       # no overflow
    D, I = idx.search(Q, 100)
    assert idx.stats["rounds"] <= 2, idx.stats
    Dr, Ir = OS.flat_ip_search(Q, P, 100)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D, Dr)


def test_randomised_cases_match_the_oracle(torch_cuda):
    """A slice of tools/dbg/search_fuzz.py (60 cases there, 0 mismatches): random sizes and widths, duplicated rows
    (exact ties), a dominant common component, wildly different norms, half-integer values (many tied scores), several
    add() calls per index."""
    for c in range(14):
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
        idx = _index(d)
        cut = int(rs.randint(0, n + 1))
        for a, b in ((0, cut), (cut, n)):
            if b > a:
                idx.add(P[a:b])
        D, I = idx.search(Q, k)
        Dr, Ir = OS.flat_ip_search(Q, P, k)
        np.testing.assert_array_equal(I, Ir, err_msg="case %d" % c)
        np.testing.assert_array_equal(D, Dr, err_msg="case %d" % c)


def test_back_to_back_streamed_adds_keep_their_rows(torch_cuda):
    """Two host blocks added one right after the other both go through the same two pinned staging buffers: the second
    add() must not refill a buffer whose copy from the first is still in flight (it did, once in ~500 runs, until the
    buffers' completion events outlived the call).  Checks the resident block bit for bit, many times."""
    torch = torch_cuda
    rs = np.random.RandomState(3)
    A = rs.randn(20515, 768).astype(np.float32)
    B = rs.randn(12485, 768).astype(np.float32)
    want = torch.from_numpy(np.concatenate([A, B], 0))
    for rep in range(25):
        idx = _index(768)
        idx.add(A)
        idx.add(B)
        assert torch.equal(idx._p32.cpu(), want), "rep %d" % rep


def test_norms_spread_over_orders_of_magnitude_fall_through_to_the_exhaustive_rung(torch_cuda):
    """eps scales with the LARGEST norm of the block: with norms spread over e^+-6 the error band of the k-th score holds
    more than 8192 passages even for the split-bf16 scan.  FAISS still answers; so does the index -- slices of <= cap
    rows searched with every row a candidate, merged, and the <= 2k survivors ranked once more on the canonical fp64
    scores (two rows whose scores round to one fp32 value are not a tie) -- found by tools/dbg/search_fuzz.py."""
    rs = np.random.RandomState(359)
    n, d, nq, k = 33000, 768, 40, 333
    P = rs.randn(n, d).astype(np.float32) * np.exp(rs.randn(n, 1) * 2).astype(np.float32)
    Q = rs.randn(nq, d).astype(np.float32)
    idx = _index(d)
    idx.add(P[:17812]); idx.add(P[17812:])
    D, I = idx.search(Q, k)
    assert idx.stats.get("exhaustive_queries", 0) > 0, idx.stats      # (the case is meant to reach the last rung)
    Dr, Ir = OS.flat_ip_search(Q, P, k)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D, Dr)


def test_one_million_block_of_a_trained_statistics_model_certifies(torch_cuda):
    """The 1M x 768 block search (BASELINE configs[1]) over embeddings that a model with TRAINED-checkpoint statistics
    produced (tests/helpers.py:trained_like_: massive activations, heavy-tailed embeddings, saturated heads) -- every other
    encoded-corpus search ran on the N(0, 0.02) initialisation.  1M distinct 128-token passages are encoded by the HIP path
    (~25 s), 1k encoded 32-token queries, k = 100: every query must certify; which rung of the precision ladder did is
    recorded (gpurun_out/margins.json: trained_stats_search/*).  Correctness as in the other full-size tests: sorted, no
    duplicates, scores = the fp64 re-dot of the returned rows, and an independent fp32 GEMM (rocBLAS through torch) finds
    fewer than k passages above the k-th score."""
    torch = torch_cuda
    from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
    from tests.helpers import margin, trained_like_
    torch.manual_seed(0)
    model = trained_like_(MSMarcoConfigDict["rdot_nll"].model_class(RobertaConfig()), seed=5).cuda().eval()
    tower, head = model.roberta, (model.embeddingHead, model.norm)
    n, nq, k, d, EB = 1_000_000, 1000, 100, 768, 2048

    def tokens(B, L, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        ids = torch.randint(3, 50000, (B, L), generator=g, device="cuda")
        ids[:, 0] = 0
        return ids
    P = torch.empty((n, d), dtype=torch.float32, device="cuda")
    lens = np.full(EB, 128, np.int32)
    with torch.no_grad():
        for s0 in range(0, n, EB):
            m = min(EB, n - s0)
            P[s0:s0 + m] = tower.embed(tokens(EB, 128, 7000 + s0 // EB), None, head=head, seq_lens=lens)[:m]
        Q = tower.embed(tokens(nq, 32, 99), None, head=head, seq_lens=np.full(nq, 32, np.int32))
    assert torch.isfinite(P).all() and torch.isfinite(Q).all()
    a, b = torch.nn.functional.normalize(P[:2048], dim=1), torch.nn.functional.normalize(P[2048:4096], dim=1)
    margin("trained_stats_search/mean_pairwise_cosine_raw", float((a @ b.T).mean()), 0.9999)     # not a collapsed corpus
    idx = _index()
    idx.add(P)
    D, I = idx.search_tensors(Q, k)
    st = dict(idx.stats)
    margin("trained_stats_search/queries_retried", st.get("retried") or 0, nq)                 # (recorded: which rung certified)
    margin("trained_stats_search/queries_on_split_scan", st.get("x3_queries") or 0, nq)
    margin("trained_stats_search/kernel_rounds", st.get("rounds") or 0, 8)
    assert not st.get("exhaustive_queries"), st                                                  # nobody fell through the ladder
    assert (D[:, :-1] >= D[:, 1:]).all()
    assert ((I >= 0) & (I < n)).all()
    assert all(len(set(row)) == k for row in I[::50].tolist())
    exact = torch.einsum("qd,qkd->qk", Q.double(), P[I.reshape(-1)].reshape(nq, k, d).double())
    assert torch.allclose(exact.float(), D, rtol=0, atol=1e-4)
    kth = D[:, -1:].clone()
    above = torch.zeros(nq, dtype=torch.int64, device="cuda")
    for s in range(0, n, 125_000):
        S = Q @ P[s:s + 125_000].T
        above += (S > kth + 5e-3).sum(1)
    assert (above <= k - 1).all(), above.max().item()


@pytest.mark.parametrize("n,nq,k,d,kind", [
    (700, 16, 100, 768, "plain"),        # n <= cap: every passage is a candidate, band = everything above the cut
    (40000, 24, 100, 768, "plain"),      # sampled threshold pass
    (33000, 130, 7, 128, "plain"),       # two query tiles
    (60, 5, 100, 768, "plain"),          # k > n: -1 padding
    (20000, 9, 50, 768, "dups"),         # exact ties at the k-th score (duplicated rows): survivors ordered by index
    (30000, 12, 100, 768, "clustered"),  # large common component: wide bands, retries through the ladder
])
def test_one_launch_finish_equals_the_three_launch_chain(torch_cuda, n, nq, k, d, kind):
    """Round 5: k_ip_finish (cut + fp64 re-score + select of a query in ONE workgroup) against k_ip_cut + k_ip_rescore +
    k_ip_select (convdr_set_option("ip_fused_finish", 0)): identical (D, I), identical ladder statistics, both equal to the
    oracle.  (Every other search test of the suite runs the default = fused form.)"""
    from convdr_amd import _lib
    rs = np.random.RandomState(n + nq)
    P = synth_corpus(300 + n % 83, n, d)
    if kind == "dups":
        P[rs.randint(0, n, size=n // 4)] = P[rs.randint(0, n, size=n // 4)]
    elif kind == "clustered":
        P = (0.9 * rs.randn(d).astype(np.float32)[None, :] + 0.12 * P).astype(np.float32)
    Q = synth_corpus(9, nq, d)
    if kind == "clustered":
        Q = (0.9 * P[:nq] / 0.9 + 0.05 * Q).astype(np.float32)
    out = {}
    for fused in (1, 0):
        _lib.check(_lib.lib().convdr_set_option(b"ip_fused_finish", fused), "set_option")
        try:
            idx = _index(d)
            idx.add(P)
            D, I = idx.search(Q, k)
            out[fused] = (D, I, {s: idx.stats.get(s) for s in ("retried", "rounds", "x3_queries", "exhaustive_queries")})
        finally:
            _lib.lib().convdr_set_option(b"ip_fused_finish", 1)
    np.testing.assert_array_equal(out[1][0], out[0][0])
    np.testing.assert_array_equal(out[1][1], out[0][1])
    assert out[1][2] == out[0][2], (out[1][2], out[0][2])
    Dr, Ir = OS.flat_ip_search(Q, P, k)
    np.testing.assert_array_equal(out[1][1], Ir)
    np.testing.assert_array_equal(out[1][0], Dr)


@pytest.mark.parametrize("n,nq,k,d", [(9000, 3, 3000, 64), (13000, 3, 5000, 64), (13000, 2, 13000, 128), (6000, 2, 9000, 64)])
def test_top_n_above_the_kernel_limit_is_chunked_on_the_host(torch_cuda, n, nq, k, d):
    """The reference takes any --top_n (run_convdr_inference.py:316-319).  k in (2048, 4096] needs the 8192-entry candidate
    list; k > 4096 takes FlatIPIndex._search_large_k (slices ranked completely, merged, fp32-tie runs re-ranked): still the
    oracle's (D, I), bit for bit -- with duplicated rows in different slices so that equal scores meet across slices and
    across rank k."""
    P, Q = synth_corpus(300 + n % 83, n, d), synth_corpus(9, nq, d)
    P[n - 50:n - 10] = P[10:50]                 # exact duplicates 4096-row slices apart: cross-slice ties, index order decides
    idx = _index(d)
    idx.add(P)
    D, I = idx.search(Q, k)
    Dr, Ir = OS.flat_ip_search(Q, P, k)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D, Dr)
    if k > 4096:
        assert idx.stats.get("large_k") == k
