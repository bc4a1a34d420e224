"""world_size-2 rehearsal of the N > 1 paths with the REAL kernels: two processes share cuda:0 and talk over gloo (RCCL
refuses two ranks on one device), so everything but the transport is what `bench.py --gpus N` / a DDP run executes:
the device-side sharded search + merge chain, and the data-parallel training step with the summed-gradient all-reduce
and the 1 / W folded into the clip pass."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, fn, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pickle
        res = fn(rank, world)
        with open(os.path.join(ret, "rank%d.pkl" % rank), "wb") as f:       # (`ret`: the parent's temporary directory)
            pickle.dump(res, f)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2, port=29641):
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
    from convdr_amd.search import FlatIPIndex
    from oracle import search as OS
    rs = np.random.RandomState(0)
    N, d, k, nq = 20011, 768, 100, 37
    P = rs.randn(N, d).astype(np.float32)
    P[301] = P[7]                                # an exact duplicate that lands on the other rank: the tie rule
    Q = rs.randn(nq, d).astype(np.float32)
    mine = blocks.shard_indices(N, world, rank)  # records i % W == rank, like the encode loop writes them
    dev = torch.device("cuda", 0)
    index = FlatIPIndex(d, device=dev)
    index.add(P[mine])
    per = (nq + world - 1) // world              # every rank "encoded" a slice of the queries
    Ql = torch.zeros(per, d, device=dev)
    Ql[:max(0, min(per, nq - rank * per))] = torch.from_numpy(Q[rank * per:(rank + 1) * per]).to(dev)
    Qall = parallel.all_gather_rows(Ql)[:nq]
    D, I, status = parallel.search_sharded_device(index, Qall, k, torch.from_numpy(mine).to(dev))
    blocks_all = [(P[blocks.shard_indices(N, world, r)], blocks.shard_indices(N, world, r)) for r in range(world)]
    mD, mI = OS.search_one_by_one(blocks_all, Q, k)
    return bool(int((status != 0).sum()) == 0 and np.array_equal(I.cpu().numpy(), mI[:, :k]) and
                np.array_equal(D.cpu().numpy(), mD[:, :k].astype(np.float32)))


def test_two_ranks_sharded_search_on_device_equals_search_one_by_one():
    assert all(_run(_search_job, 2, 29641))


def _clustered_search_job(rank, world):
    """Encoder-like shards: a dominant common component (cos ~ 0.98 between passages), with and without norms spread over
    e^+-2.  A first scan pass over the raw rows cannot certify such queries; the lists that enter the exchange must
    nevertheless be exact, like the reference's per-GPU IndexFlatIP (run_convdr_inference.py:180-182, 356-367)."""
    from convdr_amd import blocks, parallel
    from convdr_amd.search import FlatIPIndex
    from oracle import search as OS
    rs = np.random.RandomState(0)
    N, d, k, nq = 24013, 768, 50, 21
    c = rs.randn(d).astype(np.float32)
    P0 = (0.9 * c[None, :] + 0.12 * rs.randn(N, d)).astype(np.float32)
    P1 = P0 * np.exp(rs.uniform(-2, 2, size=(N, 1))).astype(np.float32)
    Q = (0.9 * c[None, :] + 0.12 * rs.randn(nq, d)).astype(np.float32)
    dev = torch.device("cuda", 0)
    mine = blocks.shard_indices(N, world, rank)
    out = {}
    # ("bf16", un-centred) on the equal-norm block: the loosest rung on raw clustered rows cannot certify anything -- the
    # ladder certainly has work (with the norm spread the scores separate by norm and even that rung certifies)
    for name, P, precision, center in (("spread/auto", P1, "auto", True), ("plain/auto", P0, "auto", True),
                                       ("plain/bf16-raw", P0, "bf16", False)):
        blocks_all = [(P[blocks.shard_indices(N, world, r)], blocks.shard_indices(N, world, r)) for r in range(world)]
        mD, mI = OS.search_one_by_one(blocks_all, Q, k)
        index = FlatIPIndex(d, device=dev, precision=precision, center=center)
        index.add(P[mine])
        Qall = torch.from_numpy(Q).to(dev)
        first = index.search_device(Qall, k)[2]
        D, I, status = parallel.search_sharded_device(index, Qall, k, torch.from_numpy(mine).to(dev))
        out[name] = (int((first != 0).sum()), int((status != 0).sum()), dict(index.stats),
                     bool(np.array_equal(I.cpu().numpy(), mI[:, :k])),
                     bool(np.array_equal(D.cpu().numpy(), mD[:, :k].astype(np.float32))))
    return out


def test_two_ranks_sharded_search_is_certified_on_clustered_shards():
    res = _run(_clustered_search_job, 2, 29645)
    for out in res:
        for name, (first_bad, bad, stats, ids_ok, scores_ok) in out.items():
            assert bad == 0 and ids_ok and scores_ok, (name, first_bad, bad, stats, ids_ok, scores_ok)
        # the raw bf16 rung really needed the ladder (otherwise this test shows nothing)
        assert out["plain/bf16-raw"][0] > 0 and out["plain/bf16-raw"][2]["retried"] > 0, out["plain/bf16-raw"][:3]


def _train_job(rank, world):
    from types import SimpleNamespace
    from convdr_amd import parallel
    from convdr_amd import train as TR
    from tests.test_train_gpu import _batch, _tiny
    rs = np.random.RandomState(11)
    B = 8
    ids, mask = _batch(rs, B, 40, [40, 17, 33, 1, 8, 25, 40, 12])
    tid, tmask = _batch(rs, B, 16, [16, 9, 4, 16, 7, 3, 11, 16])
    args = SimpleNamespace(learning_rate=1e-3, adam_epsilon=1e-8, max_grad_norm=1.0, ranking_task=False, no_mse=False,
                           num_negatives=0, gradient_accumulation_steps=1)
    dev = torch.device("cuda", 0)

    def run(sel, ddp_on):
        student, teacher = _tiny(seed=3).to(dev).train(), _tiny(seed=4).to(dev).eval()
        TR.flatten_parameters(student)
        opt = TR.get_optimizer(args, student, weight_decay=0.0)
        sched = TR.get_linear_schedule_with_warmup(opt, 0, 10)
        ddp = parallel.DataParallelStudent(student) if ddp_on else None
        batch = tuple(x[sel].to(dev) for x in (ids, mask, tid, tmask))
        loss = TR.train_step(args, student, teacher, opt, sched, batch, ddp=ddp)[0]
        return loss.item(), {k: v.detach().cpu().clone() for k, v in student.state_dict().items()}

    half = slice(rank * B // world, (rank + 1) * B // world)
    loss_dp, sd_dp = run(half, True)                 # this rank's half of the batch, gradients all-reduced
    loss_1, sd_1 = run(slice(0, B), False)           # the whole batch in one process
    worst = 0.0
    for k, v in sd_1.items():
        if v.dtype.is_floating_point and "key.bias" not in k:
            # the first Adam step moves every element with a gradient by +-lr: the two runs can differ only where the
            # sign of a rounding-noise gradient flips, i.e. in a vanishing fraction of the elements
            worst = max(worst, (v - sd_dp[k]).abs().mean().item() / args.learning_rate)
    return worst, loss_dp, loss_1


def test_two_ranks_data_parallel_step_equals_one_process_on_the_whole_batch():
    out = _run(_train_job, 2, 29643)
    # the mean-over-batch loss of the whole batch is the mean of the two halves' losses
    assert abs(0.5 * (out[0][1] + out[1][1]) - out[0][2]) < 2e-5 * max(1.0, abs(out[0][2])), out
    for worst, _, _ in out:
        assert worst < 0.02, out     # mean |difference| per tensor, in units of the step size


def _rank_inbatch_job(rank, world):
    """configs[4]'s step (run_convdr_train.py:118-171 + the north_star all-gather): KD + ranking with in-batch negatives.
    Each rank owns half of the batch AND that half's documents; the ranks' document embeddings are all-gathered (gloo
    here, RCCL on real ranks), every local query is scored against all W x B x K documents (convdr_inbatch_ce_fwd_bwd), the
    gradients are summed over the ranks with 1 / W folded into the clip."""
    from types import SimpleNamespace
    from convdr_amd import parallel
    from convdr_amd import train as TR
    from oracle import train as OT
    from tests.test_train_gpu import _batch, _tiny
    rs = np.random.RandomState(13)
    B, K = 8, 4
    ids, mask = _batch(rs, B, 40, [40, 17, 33, 1, 8, 25, 40, 12])
    tid, tmask = _batch(rs, B, 16, [16, 9, 4, 16, 7, 3, 11, 16])
    docs = torch.from_numpy(rs.randn(B * K, 768).astype(np.float32) * 0.3)       # the frozen teacher's document embeddings
    args = SimpleNamespace(learning_rate=1e-3, adam_epsilon=1e-8, max_grad_norm=1.0, ranking_task=True, no_mse=False,
                           num_negatives=K - 1, gradient_accumulation_steps=1, in_batch_negatives=True)
    dev = torch.device("cuda", 0)

    # the one-process reference run lives inside the same 2-rank job: its document "gather" must stay on this rank
    solo = [dist.new_group([r]) for r in range(world)][rank]

    def run(sel, ddp_on):
        student, teacher = _tiny(seed=3).to(dev).train(), _tiny(seed=4).to(dev).eval()
        TR.flatten_parameters(student)
        opt = TR.get_optimizer(args, student, weight_decay=0.0)
        sched = TR.get_linear_schedule_with_warmup(opt, 0, 10)
        ddp = parallel.DataParallelStudent(student) if ddp_on else SimpleNamespace(group=solo, allreduce_grads=lambda **kw: 1.0)
        batch = tuple(x[sel].to(dev) for x in (ids, mask, tid, tmask))
        dsel = docs.view(B, K, 768)[sel].reshape(-1, 768).to(dev)
        with torch.no_grad():
            embs = student.eval()(batch[0], batch[1]).cpu()          # (dropout is 0 in the tiny model: = the step's forward)
        student.train()
        loss, l1, l2 = TR.train_step(args, student, teacher, opt, sched, batch, ddp=ddp, doc_embs=dsel)
        return l1.item(), l2.item(), embs, {k: v.detach().cpu().clone() for k, v in student.state_dict().items()}

    n = B // world
    half = slice(rank * n, (rank + 1) * n)
    l1_dp, l2_dp, embs_dp, sd_dp = run(half, True)
    l1_1, l2_1, embs_1, sd_1 = run(slice(0, B), False)
    # oracle: the global-batch in-batch loss from the (HIP) embeddings of all queries -- and this rank's share of it
    pos = torch.arange(B) * K
    l2_oracle = OT.inbatch_rank_loss(embs_1, docs, pos).item()
    l2_oracle_rank = OT.inbatch_rank_loss(embs_dp, docs, pos[half]).item()
    worst = 0.0
    for k, v in sd_1.items():
        if v.dtype.is_floating_point and "key.bias" not in k:
            worst = max(worst, (v - sd_dp[k]).abs().mean().item() / args.learning_rate)
    return worst, l1_dp, l2_dp, l1_1, l2_1, l2_oracle, l2_oracle_rank


def test_two_ranks_ranking_step_with_inbatch_negatives_equals_one_process_on_the_whole_batch():
    out = _run(_rank_inbatch_job, 2, 29647)
    l2_1, l2_oracle = out[0][4], out[0][5]
    # one process, whole batch: the device loss is the oracle's definition on the same embeddings
    assert abs(l2_1 - l2_oracle) < 1e-4 * max(1.0, abs(l2_oracle)), out
    # every rank's loss is the oracle's loss of ITS queries against ALL gathered documents ...
    for o in out:
        assert abs(o[2] - o[6]) < 1e-4 * max(1.0, abs(o[6])), out
    # ... and the mean over ranks is the global-batch loss (equal shares), for both terms
    assert abs(0.5 * (out[0][2] + out[1][2]) - l2_1) < 1e-4 * max(1.0, abs(l2_1)), out
    assert abs(0.5 * (out[0][1] + out[1][1]) - out[0][3]) < 2e-5 * max(1.0, abs(out[0][3])), out
    for o in out:
        assert o[0] < 0.02, out      # weights after the step: mean |difference| per tensor, in units of the step size


def test_c_abi_collectives_over_rccl_one_rank():
    """convdr_comm_* (csrc/comm.hip: RCCL behind the C ABI, for hosts that are not Python -- SURVEY.md section 8b's proposed
    wrappers): a 1-rank communicator on the one GPU of the box (RCCL refuses two ranks per device): unique id -> init ->
    ranks -> all-gather (identity at W = 1) -> in-place sum all-reduce (identity) -> destroy, through ctypes exactly as a
    C host would call them, with the RCCL torch has already loaded (no second copy of the library in the process)."""
    import ctypes as C
    from convdr_amd import _lib
    L = _lib.lib()
    ident = (C.c_char * 128)()
    _lib.check(L.convdr_comm_unique_id(ident), "convdr_comm_unique_id")
    comm = C.c_void_p()
    _lib.check(L.convdr_comm_init(C.byref(comm), 1, 0, ident), "convdr_comm_init")
    try:
        n, r = C.c_int(-1), C.c_int(-1)
        _lib.check(L.convdr_comm_ranks(comm, C.byref(n), C.byref(r)), "convdr_comm_ranks")
        assert (n.value, r.value) == (1, 0)
        x = torch.randn(1000, 768, device="cuda")
        y = torch.empty_like(x)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(L.convdr_comm_allgather(comm, _lib.ptr(x), _lib.ptr(y), x.numel() * 4, st), "convdr_comm_allgather")
        g = torch.randn(28_000_000 // 4, device="cuda")                 # one encoder layer's gradient slice
        g0 = g.clone()
        _lib.check(L.convdr_comm_allreduce_f32(comm, _lib.ptr(g), _lib.ptr(g), g.numel(), st), "convdr_comm_allreduce_f32")
        torch.cuda.synchronize()
        assert torch.equal(y, x) and torch.equal(g, g0)
        # argument errors come back as error codes, not crashes
        assert L.convdr_comm_init(C.byref(C.c_void_p()), 2, 5, ident) != 0 and b"rank 5 of 2" in L.convdr_last_error()
    finally:
        _lib.check(L.convdr_comm_destroy(comm), "convdr_comm_destroy")


def _sparse_train_job(rank, world):
    """The data-parallel KD step with the word-embedding gradient exchanged as (row ids, rows) -- DataParallelStudent(
    sparse_embedding=True) -- against the same step with the dense all-reduce: real kernels, two processes on cuda:0 over gloo."""
    from types import SimpleNamespace
    from convdr_amd import parallel
    from convdr_amd import train as TR
    from tests.test_train_gpu import _batch, _tiny
    rs = np.random.RandomState(17)
    B = 8
    ids, mask = _batch(rs, B, 40, [40, 17, 33, 1, 8, 25, 40, 12], vocab=60)      # 60 ids: the two halves share many rows
    tid, tmask = _batch(rs, B, 16, [16, 9, 4, 16, 7, 3, 11, 16], vocab=60)
    args = SimpleNamespace(learning_rate=1e-3, adam_epsilon=1e-8, max_grad_norm=1.0, ranking_task=False, no_mse=False,
                           num_negatives=0, gradient_accumulation_steps=1)
    dev = torch.device("cuda", 0)
    half = slice(rank * B // world, (rank + 1) * B // world)
    # (the embedding-table gradients without atomics: otherwise two runs of the SAME path already differ in the last bit)
    from convdr_amd import _lib
    _lib.check(_lib.lib().convdr_set_option(b"embed_bwd_deterministic", 1), "convdr_set_option")

    def run(sparse, bf16=False):
        student, teacher = _tiny(seed=3).to(dev).train(), _tiny(seed=4).to(dev).eval()
        TR.flatten_parameters(student)
        opt = TR.get_optimizer(args, student, weight_decay=0.0)
        sched = TR.get_linear_schedule_with_warmup(opt, 0, 10)
        ddp = parallel.DataParallelStudent(student, sparse_embedding=sparse, allreduce_dtype="bf16" if bf16 else None)
        batch = tuple(x[half].to(dev) for x in (ids, mask, tid, tmask))
        loss = TR.train_step(args, student, teacher, opt, sched, batch, ddp=ddp)[0]
        torch.cuda.synchronize()
        return loss.item(), student.roberta._flat["P"].detach().cpu().clone(), dict(ddp.last_comm), ddp.last_path
    l_d, P_d, comm_d, path_d = run(False)
    l_s, P_s, comm_s, path_s = run(True)
    l_b, P_b, comm_b, _ = run(False, bf16=True)
    return (l_d, l_s, float((P_d - P_s).abs().max()), comm_s, path_d, path_s, float((P_d - P_b).abs().max()), comm_b["allreduce_dtype"],
            P_s.double().sum().item())


def test_two_ranks_sparse_embedding_exchange_equals_the_dense_allreduce_on_real_kernels():
    out = _run(_sparse_train_job, 2, 29647)
    for l_d, l_s, dmax, comm, path_d, path_s, dmax_bf16, dt, digest in out:
        assert l_d == l_s                                   # same forward
        assert dmax == 0.0, dmax                            # two ranks: a + b has no summation order -> the SAME weights after the step
        assert path_d == path_s == "overlapped"             # both ran the per-layer collectives under the backward
        assert comm["sparse_embedding"] and 0 < comm["embedding_rows_padded"] <= 60
        assert comm["embedding_bytes_gathered"] < comm["embedding_bytes_dense"]
        assert dt == "bf16" and 0 < dmax_bf16 < 5e-3        # bf16 buckets: a different (rounded) sum, a step of the same size
    assert out[0][8] == out[1][8]                           # replicas bit-identical after the sparse step
