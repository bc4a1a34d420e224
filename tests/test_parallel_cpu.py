"""world_size-2 gloo tests of the multi-process paths (CPU): corpus sharding, sharded search merge, gradient
all-reduce.  The per-rank compute is played by the CPU oracle here; the GPU kernels are covered by the -m gpu tests."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, fn, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pickle
        res = fn(rank, world)
        with open(os.path.join(ret, "rank%d.pkl" % rank), "wb") as f:       # (`ret`: the parent's temporary directory)
            pickle.dump(res, f)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2, port=29611):
    # results come back through files, not through an mp.Manager: its server process is FORKED from a pytest process that has used
    # the GPU, and on a cold box that server was once found dead ("ConnectionRefusedError" from the proxy: the soak run of round 6)
    import pickle
    import tempfile
    with tempfile.TemporaryDirectory(prefix="convdr_mp_") as td:
        mp.spawn(_worker, args=(world, port, fn, td), nprocs=world, join=True)
        out = []
        for r in range(world):
            with open(os.path.join(td, "rank%d.pkl" % r), "rb") as f:
                out.append(pickle.load(f))
    return out


def _search_job(rank, world):
    from convdr_amd import blocks, parallel
    from oracle import search as OS
    rs = np.random.RandomState(0)
    N, d, k = 601, 64, 20
    P = rs.randn(N, d).astype(np.float32)
    P[300] = P[7]                               # exact duplicate across ranks -> tie rule
    Q = rs.randn(9, d).astype(np.float32)
    mine = blocks.shard_indices(N, world, rank)  # records i % W == rank, like the encode loop writes them

    class OracleIndex:
        device = torch.device("cpu")

        def search(self, q, kk):
            return OS.flat_ip_search(q, P[mine], kk)
    D, I = parallel.search_sharded(OracleIndex(), Q, k, mine)
    # reference semantics: search_one_by_one over blocks 0..W-1
    blocks_all = [(P[blocks.shard_indices(N, world, r)], blocks.shard_indices(N, world, r)) for r in range(world)]
    mD, mI = OS.search_one_by_one(blocks_all, Q, k)
    return bool(np.array_equal(I, mI[:, :k]) and np.allclose(D, mD[:, :k]))


def test_sharded_search_merge_matches_search_one_by_one():
    assert all(_run(_search_job, 2, 29611))


def _search_device_job(rank, world):
    """The tensor-in / tensor-out form bench.py uses at N > 1 (device = CPU here, search played by the oracle)."""
    from convdr_amd import blocks, parallel
    from oracle import search as OS
    rs = np.random.RandomState(1)
    N, d, k = 433, 64, 15
    P = rs.randn(N, d).astype(np.float32)
    P[200] = P[3]
    Q = rs.randn(7, d).astype(np.float32)
    mine = blocks.shard_indices(N, world, rank)

    class OracleIndex:
        def search_device(self, q, kk):
            D, I = OS.flat_ip_search(q.numpy(), P[mine], kk)
            return torch.from_numpy(D), torch.from_numpy(I), torch.zeros(len(D), dtype=torch.int32), None
    D, I, st = parallel.search_sharded_device(OracleIndex(), torch.from_numpy(Q), k, torch.from_numpy(mine))
    blocks_all = [(P[blocks.shard_indices(N, world, r)], blocks.shard_indices(N, world, r)) for r in range(world)]
    mD, mI = OS.search_one_by_one(blocks_all, Q, k)
    return bool(np.array_equal(I.numpy(), mI[:, :k]) and np.allclose(D.numpy(), mD[:, :k]) and int(st.sum()) == 0)


def test_sharded_search_device_tensors():
    assert all(_run(_search_device_job, 2, 29613))


def _ddp_job(rank, world):
    from convdr_amd import parallel
    torch.manual_seed(0)
    model = torch.nn.Linear(8, 4)
    if rank == 1:
        with torch.no_grad():
            model.weight.add_(1.0)               # ranks start different: the constructor must broadcast rank 0
    ddp = parallel.DataParallelStudent(model)
    w_after_bcast = model.weight.detach().clone()
    flat = torch.zeros(36)
    model.weight.grad = flat[:32].view(4, 8)     # one gradient arena, like convdr_encoder_backward produces
    model.bias.grad = flat[32:].view(4)
    flat.fill_(float(rank + 1))
    ddp.allreduce_grads()
    avg = (model.weight.grad.mean().item(), model.bias.grad.mean().item())
    # train_step's form: SUM over ranks, the 1 / W factor handed back for the clip / AdamW pass to apply
    flat.fill_(float(rank + 1))
    scale = ddp.allreduce_grads(average=False)
    return w_after_bcast.sum().item(), avg[0], avg[1], scale, model.weight.grad.mean().item()


def test_gradient_allreduce_averages_flat_arena():
    r = _run(_ddp_job, 2, 29612)
    assert r[0][0] == r[1][0]                                   # parameters broadcast from rank 0
    for w, gw, gb, scale, gsum in r:
        assert gw == pytest.approx(1.5) and gb == pytest.approx(1.5)
        assert scale == 0.5 and gsum == pytest.approx(3.0)


def _ddp_flat_job(rank, world):
    """DataParallelStudent's constructor on a model whose parameters live in the flat arena (train.flatten_parameters): ONE
    broadcast for the ~40 parameter tensors (round 4 sent them one by one), one more per buffer."""
    from convdr_amd import parallel, train as TR
    from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
    torch.manual_seed(rank)                      # ranks start different
    cfg = RobertaConfig(vocab_size=50, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                        max_position_embeddings=40)
    model = MSMarcoConfigDict["rdot_nll"].model_class(cfg)
    TR.flatten_parameters(model)
    n_flat = len(model.roberta._flat["params"])
    n_other = len(list(model.parameters())) - n_flat + len(list(model.buffers()))     # (unused pooler etc.: outside the arena)
    ddp = parallel.DataParallelStudent(model)
    digest = float(sum(p.detach().double().sum() for p in model.parameters()))
    return ddp.broadcast_collectives, n_flat, n_other, digest


def test_constructor_broadcasts_the_flat_arena_in_one_collective():
    r = _run(_ddp_flat_job, 2, 29641)
    assert r[0][3] == r[1][3]                                   # every parameter equals rank 0's
    for ncoll, n_flat, n_other, _ in r:
        assert n_flat > 30 and ncoll == 1 + n_other


def _gather_job(rank, world):
    from convdr_amd import parallel
    x = torch.full((3, 5), float(rank))
    y = parallel.all_gather_rows(x)
    return y[:, 0].tolist()


def test_all_gather_rows_is_rank_ordered():
    for r in _run(_gather_job, 2, 29613):
        assert r == [0.0] * 3 + [1.0] * 3


def _inbatch_job(rank, world):
    """Every rank holds B queries and B x K teacher documents; the gathered matrix and the positive rows must be such
    that the mean of the per-rank in-batch losses equals the loss of the whole batch on one process."""
    from convdr_amd import train as TR
    from oracle import train as OT
    B, K, E = 3, 4, 16
    g = torch.Generator().manual_seed(5)
    embs_all = torch.randn(world * B, E, generator=g)
    docs_full = torch.randn(world * B, K, E, generator=g)
    docs_all, pos = TR.gather_inbatch_docs(docs_full[rank * B:(rank + 1) * B])
    assert docs_all.shape == (world * B * K, E)
    assert torch.equal(docs_all, docs_full.reshape(-1, E))                       # rank order
    assert torch.equal(docs_all[pos], docs_full[rank * B:(rank + 1) * B, 0])     # own positives
    local = OT.inbatch_rank_loss(embs_all[rank * B:(rank + 1) * B], docs_all, pos)
    t = local.clone()
    dist.all_reduce(t)
    pos_global = torch.arange(world * B) * K
    whole = OT.inbatch_rank_loss(embs_all, docs_full.reshape(-1, E), pos_global)
    return float(t / world), float(whole)


def test_inbatch_negative_gather_is_global_batch_loss():
    for mean_of_ranks, whole in _run(_inbatch_job, port=29641):
        assert abs(mean_of_ranks - whole) < 1e-5


class _StandInTower:
    """Deterministic stand-in for the GPU encoder in the multi-process CPU test: embedding = a fixed random projection of
    the token-id histogram.  Only the shard / batching / file-writing / barrier logic around the encoder is under test."""

    def __init__(self):
        self.W = torch.from_numpy(np.random.RandomState(5).randn(500, 768).astype(np.float32))

    def embed(self, ids, mask, head=None, seq_lens=None):
        out = torch.zeros((ids.shape[0], 768))
        for b in range(ids.shape[0]):
            out[b] = self.W[ids[b, :int(seq_lens[b])].long()].sum(0)
        return out


class _StandInModel:
    def __init__(self):
        self.roberta, self.embeddingHead, self.norm = _StandInTower(), None, None

    def parameters(self):
        return iter([self.roberta.W])


def _write_cache(path, n, L, seed):
    import json
    rs = np.random.RandomState(seed)
    lens = rs.randint(1, L + 1, size=n)
    ids = rs.randint(3, 500, size=(n, L)).astype(np.int32)
    with open(path, "wb") as f:
        for i in range(n):
            ids[i, lens[i]:] = 0
            f.write(int(lens[i]).to_bytes(4, "big") + ids[i].tobytes())
    with open(path + "_meta", "w") as f:
        json.dump({"type": "int32", "total_number": n, "embedding_size": L}, f)


def _stream_doc_job(rank, world):
    """gen_passage_embeddings.py:131-169 over two ranks: every rank encodes records i % W == rank and writes its own block
    pair between the barriers; afterwards search_one_by_one over the two blocks equals the search over the one block a
    single rank writes for the same token cache."""
    from types import SimpleNamespace
    from convdr_amd import blocks
    from convdr_amd.encode import StreamInferenceDoc
    from convdr_amd.search import search_one_by_one
    from oracle import search as OS
    root = os.environ["CONVDR_TEST_DIR"]
    model = _StandInModel()
    with blocks.TokenCache(os.path.join(root, "passages")) as cache:
        emb, embid = StreamInferenceDoc(SimpleNamespace(output_dir=os.path.join(root, "w2"), per_gpu_eval_batch_size=7,
                                                        max_seq_length=24), model, cache)
        ok = bool(np.array_equal(embid, np.arange(rank, len(cache), world)))
        if rank == 0:
            one = SimpleNamespace(output_dir=os.path.join(root, "w1"), per_gpu_eval_batch_size=64, max_seq_length=24)
            os.makedirs(one.output_dir, exist_ok=True)
            full_emb, full_id = __import__("convdr_amd.encode", fromlist=["encode_shard"]).encode_shard(model, cache, 0, 1, 64, False, 24)
            blocks.dump_block(os.path.join(one.output_dir, "passage__emb_p__data_obj_0.pb"), full_emb)
            blocks.dump_block(os.path.join(one.output_dir, "passage__embid_p__data_obj_0.pb"), full_id)
    dist.barrier()
    if rank != 0:
        return ok

    class HostIndex:           # .add / .search / .reset (the reference's FAISS surface) played by the oracle
        def add(self, x):
            self.P = np.asarray(x)

        def search(self, q, k):
            return OS.flat_ip_search(q, self.P, k)

        def reset(self):
            self.P = None
    Q = np.random.RandomState(9).randn(6, 768).astype(np.float32)
    D2, I2 = search_one_by_one(os.path.join(root, "w2"), HostIndex(), Q, 15)
    D1, I1 = search_one_by_one(os.path.join(root, "w1"), HostIndex(), Q, 15)
    return ok and bool(np.array_equal(I2[:, :15], I1[:, :15]) and np.array_equal(D2[:, :15], D1[:, :15]))


def test_two_ranks_write_two_blocks_equal_to_one_rank(tmp_path):
    _write_cache(str(tmp_path / "passages"), 101, 24, 3)
    os.environ["CONVDR_TEST_DIR"] = str(tmp_path)
    try:
        assert all(_run(_stream_doc_job, 2, 29631))
    finally:
        os.environ.pop("CONVDR_TEST_DIR", None)


def _sampler_job(rank, world):
    """Per-rank batch sharding of a training run (north_star: "DistributedSampler training batches shard"): the ranks'
    samplers partition every epoch's permutation, re-seeded per epoch; shard_batch cuts a global batch the way
    nn.DataParallel's scatter does (run_convdr_train.py:77-78)."""
    from convdr_amd import parallel
    data = list(range(103))
    smp = parallel.train_sampler(data, shuffle=True, seed=5)
    epochs = []
    for ep in range(2):
        smp.set_epoch(ep)
        epochs.append(list(iter(smp)))
    batch = (torch.arange(16).view(8, 2), np.arange(8), None, {"k": 1})
    mine = parallel.shard_batch(batch)
    return epochs, mine[0].tolist(), mine[1].tolist(), mine[2], mine[3]


def test_distributed_sampler_and_batch_sharding():
    out = _run(_sampler_job, 2, 29617)
    for ep in range(2):
        a, b = out[0][0][ep], out[1][0][ep]
        assert len(a) == len(b) == 52                        # ceil(103 / 2): one index is repeated to pad, as torch does
        assert set(a) | set(b) == set(range(103)) and len(set(a) & set(b)) <= 1
    assert out[0][0][0] != out[0][0][1]                      # set_epoch reshuffles
    assert out[0][1] + out[1][1] == torch.arange(16).view(8, 2).tolist()
    assert out[0][2] + out[1][2] == list(range(8))
    assert out[0][3] is None and out[0][4] == {"k": 1}
    from convdr_amd import parallel
    from torch.utils.data import RandomSampler
    assert isinstance(parallel.train_sampler(list(range(5))), RandomSampler)      # world size 1: the reference's sampler
    # a batch that does not divide: cut like nn.DataParallel's scatter (torch.chunk: the last replica short), and the
    # per-rank loss weights n_r W / n turn the ranks' mean losses into the reference's mean over the gathered batch
    x = torch.arange(14.0).view(7, 2)
    for W in (2, 3, 4):
        ref = torch.chunk(x, W)
        got = [parallel.shard_batch((x, np.arange(7)), rank=r, world=W, return_weight=True) for r in range(len(ref))]
        for r, c in enumerate(ref):
            assert torch.equal(got[r][0][0], c) and got[r][0][1].tolist() == list(range(7))[sum(len(q) for q in ref[:r]):][:len(c)]
        assert parallel.shard_sizes(7, W)[:len(ref)] == [len(c) for c in ref]
        assert abs(sum(w for _, w in got) - W) < 1e-12
        # sum_r (w_r / W) mean_r == global mean
        assert abs(sum(w / W * b[0].mean().item() for b, w in got) - x.mean().item()) < 1e-6
    with pytest.raises(ValueError):
        parallel.shard_batch((torch.zeros(5, 2),), rank=3, world=4)      # torch.chunk(5, 4) = 2, 2, 1: rank 3 is empty
    with pytest.raises(ValueError, match="return_weight"):
        parallel.shard_batch((torch.zeros(7, 2),), rank=0, world=2)      # ragged cut without its loss weight: refused
    assert parallel.shard_batch((torch.zeros(8, 2),), rank=1, world=2)[0].shape[0] == 4


def _sparse_embedding_job(rank, world):
    """DataParallelStudent(sparse_embedding=True): the word-embedding gradient travels as (row ids, rows) in ONE all-gather
    and is summed locally in rank order; everything else dense.  Against the dense all-reduce of the same gradients."""
    from convdr_amd import parallel, train as TR
    from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
    torch.manual_seed(0)
    cfg = RobertaConfig(vocab_size=300, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256,
                        max_position_embeddings=40)
    model = MSMarcoConfigDict["rdot_nll"].model_class(cfg)
    info = TR.flatten_parameters(model)
    params = info["params"]
    g = torch.Generator().manual_seed(100 + rank)
    n = sum(p.numel() for p in params)

    def fresh_grads():
        G = torch.randn(n, generator=torch.Generator().manual_seed(100 + rank))
        o = 0
        for p in params:
            p.grad = G[o:o + p.numel()].view(p.shape)
            o += p.numel()
        wg = model.roberta.embeddings.word_embeddings.weight.grad
        # every rank touched its own few rows (some shared with the other ranks: rows 5 and 7), the rest is exactly zero
        mine = sorted({5, 7, 11 + rank, 40 + 3 * rank, 299 - rank})
        keep = torch.zeros(wg.shape[0], dtype=torch.bool)
        keep[mine] = True
        wg[~keep] = 0.0
        return G, mine
    out = {}
    for mode in ("dense", "sparse", "sparse_ids", "bf16"):
        G, mine = fresh_grads()
        ddp = parallel.DataParallelStudent(model, broadcast=False, sparse_embedding=mode.startswith("sparse"),
                                           allreduce_dtype="bf16" if mode == "bf16" else None)
        kw = {"token_ids": torch.tensor(mine + [0, 1])} if mode == "sparse_ids" else {}     # a superset (padding id etc.)
        scale = ddp.allreduce_grads(average=False, **kw)
        out[mode] = (G.clone(), scale, dict(ddp.last_comm))
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_sparse_embedding_gradient_exchange_equals_the_dense_allreduce(world):
    res = _run(_sparse_embedding_job, world, 29650 + world)
    dense = res[0]["dense"][0]
    for r in range(world):
        assert torch.equal(res[r]["dense"][0], dense)                        # (gloo's all-reduce: identical on every rank)
        for mode in ("sparse", "sparse_ids"):
            G, scale, comm = res[r][mode]
            assert scale == 1.0 / world
            assert torch.equal(G, res[0][mode][0]), mode                      # replicas stay bit-identical
            if world == 2:
                assert torch.equal(G, dense), mode                           # a + b: no summation order to differ in
            else:
                assert torch.allclose(G, dense, rtol=1e-6, atol=1e-6), mode
            assert comm["sparse_embedding"] and comm["embedding_rows_padded"] in (5, 7)
            assert comm["embedding_bytes_gathered"] == world * comm["embedding_rows_padded"] * 129 * 4
            assert comm["embedding_bytes_dense"] == 300 * 128 * 4
            assert comm["sparse_bytes_per_rank"] < comm["dense_bytes_per_rank"]
        Gb = res[r]["bf16"][0]
        assert torch.allclose(Gb, dense, rtol=2e-2, atol=2e-2) and not torch.equal(Gb, dense)
        assert res[r]["bf16"][2]["allreduce_dtype"] == "bf16"
