"""Host-side formats: token cache + embedding blocks, against plain pickle / the oracle restatement
and the bytes produced by the reference run (tests/golden/encode_loop.npz).  CPU only."""
import os
import pickle

import numpy as np
import pytest

from convdr_amd import blocks
from oracle import formats as OF


def test_token_cache_matches_reference_bytes(golden_dir, tmp_path):
    z = np.load(os.path.join(golden_dir, "encode_loop.npz"))
    N, L = int(z["N"]), int(z["L"])
    path = str(tmp_path / "passages")
    open(path, "wb").write(z["token_cache"].tobytes())
    OF.write_token_cache(path + "2", [row[:n].tolist() for row, n in
                                      zip(z["token_cache"].reshape(N, 4 + 4 * L)[:, 4:].copy().view(np.int32), z["lens"])], L)
    assert open(path + "2", "rb").read() == z["token_cache"].tobytes()       # oracle writer == reference layout
    import json
    json.dump({"type": "int32", "total_number": N, "embedding_size": L}, open(path + "_meta", "w"))
    with blocks.TokenCache(path) as tc:
        assert len(tc) == N and tc.seq_len == L
        np.testing.assert_array_equal(tc.lengths(), z["lens"])
        lens, ids = OF.read_token_cache(path)
        np.testing.assert_array_equal(tc.ids, ids)
        np.testing.assert_array_equal(tc.lengths(np.array([3, 0])), lens[[3, 0]])


def test_shard_rule_is_round_robin():
    for world in (1, 2, 3, 8):
        got = np.concatenate([blocks.shard_indices(37, world, r) for r in range(world)])
        assert sorted(got.tolist()) == list(range(37))
        for r in range(world):
            assert blocks.shard_indices(37, world, r).tolist() == OF.shard_indices(37, world, r)


@pytest.mark.parametrize("shape,dtype", [((5000, 768), np.float32), ((400000,), np.int64), ((37, 768), np.float32),
                                         ((300, 768), np.float32), ((70000, 3, 5), np.float32), ((0, 768), np.float32)])
def test_dump_block_is_byte_identical_to_pickle(tmp_path, shape, dtype):
    rs = np.random.RandomState(0)
    arr = (rs.randn(*shape) * 100).astype(dtype)
    p = str(tmp_path / "b.pb")
    blocks.dump_block(p, arr)
    assert open(p, "rb").read() == pickle.dumps(arr, protocol=4)
    back = pickle.load(open(p, "rb"))                                          # the reference's reader
    assert back.dtype == arr.dtype and back.shape == arr.shape and np.array_equal(back, arr)
    with blocks.BlockView(p) as v:
        assert v.array.shape == arr.shape and v.array.dtype == arr.dtype
        np.testing.assert_array_equal(v.array, arr)


def test_block_view_reads_reference_written_blocks(golden_dir, tmp_path):
    z = np.load(os.path.join(golden_dir, "encode_loop.npz"))
    big = np.tile(z["emb"], (40, 1))                         # > 1 MiB -> out-of-frame payload like the real blocks
    p = str(tmp_path / "passage__emb_p__data_obj_0.pb")
    OF.dump_block(p, big)                                    # == utils/util.py:108-111
    with blocks.BlockView(p) as v:
        assert v.offset % 4 != 0 or True                     # payload offset is header dependent / unaligned
        np.testing.assert_array_equal(v.array, big)


def test_plan_batches_token_budget():
    """In-order batches: record cap, token cap, oversize records alone, full coverage without overlap."""
    from convdr_amd.encode import plan_batches
    rs = np.random.RandomState(0)
    lens = rs.randint(1, 513, size=1000)
    for bs, budget in ((64, None), (64, 4096), (1024, 262144), (8, 100), (1, 10 ** 9)):
        plan = plan_batches(lens, bs, budget)
        assert plan[0][0] == 0 and plan[-1][1] == len(lens)
        assert all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
        for s, e in plan:
            assert 1 <= e - s <= bs
            if budget is not None and e - s > 1:
                assert lens[s:e].sum() <= budget
            if budget is not None and e < len(lens) and e - s < bs:
                assert lens[s:e + 1].sum() > budget          # greedy: the next record would not have fitted
    plan = plan_batches(lens, 1024, 4096, align=8)           # budget in packed rows: lengths rounded up to 8
    al = (lens + 7) // 8 * 8
    assert plan[0][0] == 0 and plan[-1][1] == len(lens)
    for s, e in plan:
        assert e - s == 1 or al[s:e].sum() <= 4096
        assert e == len(lens) or al[s:e + 1].sum() > 4096
    assert plan_batches([], 8, 100) == []
    assert plan_batches([700], 8, 100) == [(0, 1)]


def test_doc_embedding_lookup_gathers_rows_by_pid(tmp_path):
    """blocks.DocEmbeddingLookup (f-2): three ragged round-robin blocks as a 3-rank encode run writes them
    (utils/util.py:422-424, :108-111) -> rows by passage id through pid2offset, dict and array forms, unknown ids raise."""
    from convdr_amd import blocks
    rs = np.random.RandomState(3)
    total, W, d = 1000, 3, 768
    emb_all = rs.randn(total, d).astype(np.float32)
    for r in range(W):
        idx = np.arange(r, total, W, dtype=np.int64)
        blocks.dump_block(str(tmp_path / ("passage__emb_p__data_obj_%d.pb" % r)), emb_all[idx])
        blocks.dump_block(str(tmp_path / ("passage__embid_p__data_obj_%d.pb" % r)), idx)
    perm = rs.permutation(total)
    pid2offset = {int(7 * p + 1): int(o) for o, p in enumerate(perm)}          # pid = 7 * perm[offset] + 1
    want_pids = [7 * int(perm[o]) + 1 for o in (5, 999, 0, 5, 333, 334)]
    with blocks.DocEmbeddingLookup(str(tmp_path), pid2offset) as lk:
        got = lk.gather(want_pids)
        np.testing.assert_array_equal(got, emb_all[[5, 999, 0, 5, 333, 334]])
        with pytest.raises(KeyError):
            lk.gather([12345678])
    arr = np.zeros(7 * total + 2, np.int64)
    for p, o in pid2offset.items():
        arr[p] = o
    with blocks.DocEmbeddingLookup(str(tmp_path), arr, n_blocks=3) as lk:
        np.testing.assert_array_equal(lk.gather(want_pids), got)
    with blocks.DocEmbeddingLookup(str(tmp_path)) as lk:                      # ids are offsets
        np.testing.assert_array_equal(lk.gather([5, 999]), emb_all[[5, 999]])


def test_block_larger_than_4gib_uses_binbytes8(tmp_path):
    """utils/util.py:108-111 at CAsT scale writes 14.6 GB blocks: pickle protocol 4 stores payloads >= 4 GiB with the
    8-byte-length BINBYTES8 opcode.  dump_block must produce that form (readable by pickle.load, what the reference does
    at run_convdr_inference.py:164-175) and BlockView must map it.  The array is 4 GiB + 3 MB of untouched (zero) pages
    plus planted rows, so the test costs disk I/O, not memory."""
    import pickle
    import shutil
    from convdr_amd import blocks
    if shutil.disk_usage(str(tmp_path)).free < 6 * (1 << 30):
        pytest.skip("needs 6 GB of scratch disk")
    n, d = (1 << 32) // (768 * 4) + 1000, 768
    arr = np.zeros((n, d), np.float32)                      # calloc: pages are not touched until written
    assert arr.nbytes > 0xffffffff
    planted = {0: 1.5, 1234567: -2.25, n - 1: 3.75}
    for r, v in planted.items():
        arr[r, :] = v
        arr[r, 5] = r % 1000
    path = str(tmp_path / "passage__emb_p__data_obj_0.pb")
    blocks.dump_block(path, arr)
    del arr
    with open(path, "rb") as f:
        head = f.read(256)
    assert b"\x8e" + (n * d * 4).to_bytes(8, "little") in head, "payload must be introduced by BINBYTES8"
    with blocks.BlockView(path) as bv:
        assert bv.array.shape == (n, d) and bv.array.dtype == np.float32
        for r, v in planted.items():
            assert bv.array[r, 0] == v and bv.array[r, 5] == r % 1000
        assert bv.array[n // 2].max() == 0.0
    with open(path, "rb") as f:
        back = pickle.load(f)                               # what the reference's search driver does
    assert back.shape == (n, d) and back.dtype == np.float32
    for r, v in planted.items():
        assert back[r, 0] == v and back[r, 5] == r % 1000
    del back
    os.remove(path)
