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
