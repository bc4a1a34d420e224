"""Multi-GPU pieces of the hot path: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm;
"gloo" in the CPU tests).  Only the steps that are real exchanges use a collective:

  corpus encode   no collective: rank r encodes records i % W == r and writes its own block pair (encode.py)
  search          every rank searches its resident block with the replicated query matrix; the per-rank top-k lists
                  (k scores + k record offsets per query: 9.6 MB at W = 8, Nq = 1k, k = 100) are all-gathered once and
                  merged with the reference's tie rule "earlier block first" (run_convdr_inference.py:218);
                  queries encoded data-parallel are all-gathered first (3 MB at Nq = 1k)
  training        the reference uses single-process nn.DataParallel (run_convdr_train.py:77-78); here: one replica per
                  GPU, per-rank batches, ONE all-reduce of the flat fp32 gradient buffer per step (the backward already
                  writes every gradient into one contiguous arena, so there is nothing to bucket), averaged over ranks
"""
import torch
import torch.distributed as dist


def _world(group=None):
    """Size of `group` (default: the whole job); 1 outside a process group."""
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def all_gather_rows(x, group=None, force=False):
    """[n_local, d] on every rank (equal n_local) -> [W * n_local, d], rank order.
    (force: run the collective at world size 1 too -- single-GPU tests of the RCCL path.)"""
    W = _world(group)
    if W == 1 and not (force and dist.is_initialized()):
        return x
    parts = [torch.empty_like(x) for _ in range(W)]
    dist.all_gather(parts, x.contiguous(), group=group)
    return torch.cat(parts, 0)


def merge_rank_topk(D_all, I_all, k):
    """D_all / I_all: [W, nq, k] per-rank results (each row sorted descending) -> global [nq, k].
    Ties keep the lower rank (= earlier block) first, then the earlier position: what chaining the reference's `>=`
    two-way merge (run_convdr_inference.py:213-229) over blocks 0..W-1 gives.  On a GPU that chain is run literally with
    the device merge kernel (convdr_topk_merge, k outputs per step: entries past rank k can never re-enter); CPU tensors
    (the gloo tests) take a stable descending sort of the rank-ordered concatenation, the same permutation."""
    W, nq, kk = D_all.shape
    if D_all.is_cuda:
        from . import _lib
        L = _lib.lib()
        D_all, I_all = D_all.contiguous(), I_all.contiguous()
        Dm, Im = D_all[0], I_all[0]
        with torch.cuda.device(D_all.device):
            for r in range(1, W):
                na = Dm.shape[1]
                no = min(k, na + kk)
                Do = torch.empty((nq, no), dtype=torch.float32, device=D_all.device)
                Io = torch.empty((nq, no), dtype=torch.int64, device=D_all.device)
                _lib.check(L.convdr_topk_merge(_lib.ptr(Dm), _lib.ptr(Im), na, Dm.stride(0), _lib.ptr(D_all[r]), _lib.ptr(I_all[r]),
                                               kk, D_all[r].stride(0), nq, no, _lib.ptr(Do), _lib.ptr(Io), Do.stride(0),
                                               _lib.stream_ptr()), "convdr_topk_merge")
                Dm, Im = Do, Io
        return Dm[:, :k], Im[:, :k]
    d = D_all.permute(1, 0, 2).reshape(nq, W * kk)
    i = I_all.permute(1, 0, 2).reshape(nq, W * kk)
    order = torch.sort(d, dim=1, descending=True, stable=True).indices[:, :k]
    return torch.gather(d, 1, order), torch.gather(i, 1, order)


def search_sharded(index, queries, k, embid, group=None):
    """Exact global top-k over a corpus whose block r is resident on rank r.
    index: FlatIPIndex (or any object with .search(q, k) -> numpy (D, I)); embid: this block's int64 record offsets
    (passage__embid_p__data_obj_{rank}.pb).  Returns (D float32 [nq, k], offsets int64 [nq, k]) on every rank."""
    import numpy as np
    D, I = index.search(queries, k)
    ids = np.where(I >= 0, np.asarray(embid)[np.clip(I, 0, None)], -1)
    dev = getattr(index, "device", torch.device("cpu"))
    Dt, It = torch.from_numpy(D).to(dev), torch.from_numpy(ids).to(dev)
    W = _world(group)
    if W == 1:
        return D, ids
    Dl = [torch.empty_like(Dt) for _ in range(W)]
    Il = [torch.empty_like(It) for _ in range(W)]
    dist.all_gather(Dl, Dt, group=group)
    dist.all_gather(Il, It, group=group)
    Dm, Im = merge_rank_topk(torch.stack(Dl), torch.stack(Il), k)
    return Dm.cpu().numpy(), Im.cpu().numpy()


def search_sharded_device(index, queries, k, embid, group=None, force=False, certify=True):
    """search_sharded without the host round trip of the results: `queries` and `embid` are tensors on the index's device,
    the per-rank (scores, record offsets) are exchanged with ONE all-gather ([W, nq, k] fp32 + int64: 9.6 MB at W = 8,
    k = 100, nq = 1000) and merged on the device.  Returns device tensors (D [nq, k] fp32, offsets [nq, k] int64,
    status [nq] int32).

    The reference's sharded search is exact on every shard (faiss IndexFlatIP per GPU, run_convdr_inference.py:180-182,
    356-367), so the LOCAL lists that enter the exchange are the certified ones: `index.search_tensors` runs the whole
    precision ladder (retries, split scan, exhaustive rung) for the queries the first pass cannot certify -- each rank
    for its own block, before any collective, so ranks may take different numbers of rounds -- and `status` is all zero.
    certify=False (instrumentation only) exchanges the first pass as it is and returns its status."""
    if certify and hasattr(index, "search_tensors"):
        D, I = index.search_tensors(queries, k)
        status = torch.zeros(D.shape[0], dtype=torch.int32, device=D.device)
    else:
        D, I, status = index.search_device(queries, k)[:3]
    ids = torch.where(I >= 0, embid[I.clamp(min=0)], torch.full_like(I, -1))
    Dm, Im = exchange_topk(D, ids, k, group=group, force=force)
    return Dm, Im, status


def exchange_topk(D, ids, k, group=None, force=False):
    """The exchange step of the sharded search on its own (bench.py times it apart from the local search): every rank's
    certified (scores [nq, k] fp32, record offsets [nq, k] int64) -> ONE all-gather -> device merge -> global (D, offsets)."""
    W = _world(group)
    if W == 1 and not (force and dist.is_initialized()):
        return D, ids
    # one collective for both: [nq, k, 3] int32 = (score bits, offset low word, offset high word)
    nq = D.shape[0]
    buf = torch.empty((nq, k, 3), dtype=torch.int32, device=D.device)
    buf[:, :, 0] = D.contiguous().view(torch.int32)
    buf[:, :, 1:] = ids.contiguous().view(torch.int32).view(nq, k, 2)
    if dist.get_backend(group) == "nccl":
        out = torch.empty((W, nq, k, 3), dtype=torch.int32, device=D.device)
        dist.all_gather_into_tensor(out, buf, group=group)
    else:                                # gloo (the CPU tests) has no single-tensor all-gather
        parts = [torch.empty_like(buf) for _ in range(W)]
        dist.all_gather(parts, buf, group=group)
        out = torch.stack(parts)
    D_all = out[..., 0].contiguous().view(torch.float32)
    I_all = out[..., 1:].contiguous().view(torch.int64).view(W, nq, k)
    return merge_rank_topk(D_all, I_all, k)


def train_sampler(dataset, shuffle=True, seed=0, drop_last=False, rank=None, world=None):
    """The sampler of a one-process-per-GPU training run.  The reference draws ONE RandomSampler batch per step and lets
    nn.DataParallel scatter it over the visible GPUs (run_convdr_train.py:51-57,77-78); with a process per GPU each rank
    draws its own disjoint 1 / W of every epoch's permutation instead (torch's DistributedSampler: rank r takes
    indices r, r + W, ... of the permutation seeded with seed + epoch -- call ``sampler.set_epoch(epoch)`` at the top of
    every epoch, where the reference's RandomSampler simply reshuffles).  World size 1: the reference's samplers.
    The per-rank batch size is ``args.per_gpu_train_batch_size``; global batch = W x that (configs[4]: 8 x 64)."""
    from torch.utils.data import RandomSampler, SequentialSampler
    from torch.utils.data.distributed import DistributedSampler
    W = _world() if world is None else int(world)
    if W == 1:
        return RandomSampler(dataset) if shuffle else SequentialSampler(dataset)
    r = dist.get_rank() if rank is None else int(rank)
    return DistributedSampler(dataset, num_replicas=W, rank=r, shuffle=shuffle, seed=seed, drop_last=drop_last)


def shard_sizes(n, world):
    """Chunk sizes of a batch of n scattered over `world` replicas the way nn.DataParallel does (torch.chunk semantics:
    every replica ceil(n / W) samples until the batch runs out -- the last one short, possibly some empty)."""
    c = -(-int(n) // int(world))
    return [max(0, min(c, int(n) - r * c)) for r in range(int(world))]


def shard_batch(batch, rank=None, world=None, return_weight=False):
    """This rank's contiguous slice of a GLOBAL batch (a tuple / list / dict of tensors or arrays whose first dimension is
    the batch): what nn.DataParallel's scatter hands replica `rank` (run_convdr_train.py:52,77-78).  For drivers that keep
    the reference's single global batch (e.g. to replay one of its runs); a DistributedSampler run never needs it.
    A batch that does not divide over the ranks is cut like DataParallel cuts it (`shard_sizes`: the last replica short)
    -- but ONLY with return_weight=True.  The reference computes its mean losses over the gathered outputs of ALL replicas,
    so with per-rank mean losses the global gradient is sum_r (n_r / n) grad_r: return_weight=True also returns n_r W / n,
    the factor `train_step(..., loss_weight=)` multiplies this rank's loss by before the backward (the all-reduce sums and
    1 / W rides on the clip pass).  Without the weight a ragged cut would silently train on the mean of per-rank means,
    so return_weight=False raises ValueError for a batch that does not divide.  A rank left without samples
    raises: it would still have to join the step's collectives."""
    W = _world() if world is None else int(world)
    r = (dist.get_rank() if W > 1 else 0) if rank is None else int(rank)
    if W == 1:
        return (batch, 1.0) if return_weight else batch
    seen = []

    def cut(x):
        import numpy as np
        if not isinstance(x, (torch.Tensor, np.ndarray, list, tuple)):      # scalars, None, strings, nested dicts: replicated
            return x
        n = len(x)
        sizes = shard_sizes(n, W)
        if sizes[r] == 0:
            raise ValueError("shard_batch: a batch of %d leaves rank %d of %d without samples" % (n, r, W))
        if n % W and not return_weight:
            raise ValueError("shard_batch: a batch of %d does not divide over %d ranks; the ragged cut needs its loss weight "
                             "(return_weight=True -> train_step(..., loss_weight=))" % (n, W))
        seen.append((n, sizes[r]))
        b = sum(sizes[:r])
        return x[b:b + sizes[r]]
    out = {k: cut(v) for k, v in batch.items()} if isinstance(batch, dict) else type(batch)(cut(x) for x in batch)
    if not return_weight:
        return out
    n, nr = seen[0] if seen else (1, 1)
    return out, nr * W / float(n)


def sparse_rows_allreduce(wgrad, group=None, token_ids=None):
    """Sum over the ranks of a [V, H] fp32 gradient of which every rank touched only a few rows (the word-embedding table:
    50,265 x 768 = 154 MB -- 31 % of roberta-base's gradient bytes and the one collective that cannot start before the
    backward has ended -- of which a 64 x 256-token batch touches at most 16 k rows, a real one a few thousand).
    Instead of a dense all-reduce: ONE all-gather of every rank's (row ids, summed rows), padded to the largest count, and
    a local scatter-add.  In place; returns {"rows": this rank's count, "rows_max": the padded count, "bytes_gathered": ...}.

    Exact in fp32 up to summation order -- and, unlike an atomics scatter, IDENTICAL ON EVERY RANK (replicas must not drift):
    the union of rows is zeroed, then the ranks' rows are added in rank order, each rank's ids unique within its call.
    token_ids: optional tensor of the token ids this rank's step embedded (a superset of its non-zero rows is fine); without it
    the non-zero rows are found by a pass over the gradient."""
    W = _world(group)
    V, H = wgrad.shape
    if token_ids is not None:
        rows = torch.unique(token_ids.reshape(-1).to(wgrad.device))
        rows = rows[(rows >= 0) & (rows < V)]
    else:
        rows = torch.nonzero((wgrad != 0).any(dim=1)).flatten()
    n = int(rows.numel())
    stats = {"rows": n, "rows_max": n, "bytes_gathered": 0, "bytes_dense": int(V * H * 4)}
    if W == 1:
        return stats
    cnt = torch.tensor([n], dtype=torch.int64, device=wgrad.device)
    cnts = [torch.zeros_like(cnt) for _ in range(W)]
    dist.all_gather(cnts, cnt, group=group)
    cnts = [int(c.item()) for c in cnts]
    nmax = max(max(cnts), 1)
    # one buffer per rank: H gradient columns + the row id bit-cast into a float column (-1 = padding)
    buf = torch.zeros((nmax, H + 1), dtype=torch.float32, device=wgrad.device)
    ids32 = torch.full((nmax,), -1, dtype=torch.int32, device=wgrad.device)
    if n:
        buf[:n, :H] = wgrad.index_select(0, rows)
        ids32[:n] = rows.to(torch.int32)
    buf[:, H] = ids32.view(torch.float32)
    if dist.get_backend(group) == "nccl":
        out = torch.empty((W, nmax, H + 1), dtype=torch.float32, device=wgrad.device)
        dist.all_gather_into_tensor(out, buf, group=group)
    else:                                # gloo (the CPU tests) has no single-tensor all-gather
        parts = [torch.empty_like(buf) for _ in range(W)]
        dist.all_gather(parts, buf, group=group)
        out = torch.stack(parts)
    ids_all = [out[r, :cnts[r], H].contiguous().view(torch.int32).to(torch.int64) for r in range(W)]
    for r in range(W):
        if cnts[r]:
            wgrad.index_fill_(0, ids_all[r], 0.0)
    for r in range(W):                   # rank order, the same on every rank: replicas stay bit-identical
        if cnts[r]:
            wgrad.index_add_(0, ids_all[r], out[r, :cnts[r], :H])
    stats.update(rows_max=nmax, bytes_gathered=int(W * nmax * (H + 1) * 4))
    return stats


class DataParallelStudent:
    """Gradient synchronisation for one-process-per-GPU training of the student.

    sparse_embedding: the word-embedding gradient (31 % of the bytes, complete only when the whole backward is) is exchanged
    as (row ids, rows) with ONE all-gather + a local rank-ordered scatter-add (`sparse_rows_allreduce`) instead of a dense
    all-reduce.  Off by default.  UNMEASURED ON HARDWARE (no multi-GPU node has been available): `last_comm` carries the
    byte counts of both forms for every step so that the day a node is there the choice can be priced.
    allreduce_dtype: "bf16" casts every per-layer bucket to bf16 for its collective and back (SURVEY section 5's option: half
    the xGMI bytes, the sum rounded per hop); None / "fp32" (default): fp32, the reference's arithmetic."""

    def __init__(self, model, group=None, broadcast=True, sparse_embedding=False, allreduce_dtype=None):
        self.model, self.group = model, group
        self.sparse_embedding = bool(sparse_embedding)
        if allreduce_dtype not in (None, "fp32", "bf16"):
            raise ValueError("allreduce_dtype must be None, 'fp32' or 'bf16'")
        self.allreduce_dtype = None if allreduce_dtype in (None, "fp32") else allreduce_dtype
        self.last_comm = {}
        self.broadcast_collectives = 0
        if broadcast and _world(group) > 1:   # what the DDP constructor does (gen_passage_embeddings.py:64-69)
            # one collective for everything that lives in the flat parameter arena (train.flatten_parameters: ~200 tensors,
            # 0.5 GB for roberta-base), one each for whatever does not (buffers, un-flattened models)
            m = model.module if hasattr(model, "module") else model
            info = getattr(getattr(m, "roberta", None), "_flat", None)
            done = set()
            if info is not None:
                dist.broadcast(info["P"], 0, group=group)
                self.broadcast_collectives += 1
                # only the parameters that still ALIAS the arena are synchronised by that broadcast (model.to() or a .data
                # re-assignment after flatten_parameters re-homes them): the others take the per-tensor broadcast below
                base = info["P"].data_ptr()
                done = {id(p) for p in info["params"] if p.data_ptr() == base + 4 * info["off"][id(p)]}
                self.rehomed_parameters = len(info["params"]) - len(done)
            for t in list(model.parameters()) + list(model.buffers()):
                if id(t) not in done:
                    dist.broadcast(t.data, 0, group=group)
                    self.broadcast_collectives += 1
            # the writes went through .data (no version bump): drop the packed bf16 copies made before them
            for mod in model.modules():
                if hasattr(mod, "invalidate_packed"):
                    mod.invalidate_packed()
        for p in model.parameters():          # a training entry point: the step's auxiliary streams are picked here
            if p.is_cuda:
                from .train import reserve_streams
                reserve_streams(p.device)
            break

    def _layer_buckets(self, n_flat):
        """[(begin, end)] of each encoder layer's gradients in the flat arena (train._tower_params order: 5 embedding
        tensors, 16 per layer, 4 of the head), or None when the model has no arena of exactly n_flat elements."""
        m = self.model.module if hasattr(self.model, "module") else self.model
        tower = getattr(m, "roberta", None)
        info = getattr(tower, "_flat", None)
        if info is None:
            return None
        offs = [0]
        for p in info["params"]:
            offs.append(offs[-1] + p.numel())
        if offs[-1] != n_flat:
            return None
        nl = len(tower.encoder.layer)
        return [(offs[5 + 16 * l], offs[5 + 16 * (l + 1)]) for l in range(nl)]

    def _word_grad(self):
        """The word-embedding gradient as a [V, H] tensor, or None."""
        m = self.model.module if hasattr(self.model, "module") else self.model
        emb = getattr(getattr(getattr(m, "roberta", None), "embeddings", None), "word_embeddings", None)
        g = getattr(getattr(emb, "weight", None), "grad", None)
        return g if (g is not None and g.dim() == 2 and g.dtype == torch.float32 and g.is_contiguous()) else None

    def _reduce(self, t, async_op=False):
        """Sum `t` over the ranks in place (optionally through bf16: self.allreduce_dtype)."""
        if self.allreduce_dtype == "bf16":
            h = t.to(torch.bfloat16)
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
            return None
        return dist.all_reduce(t, group=self.group, async_op=async_op)

    def allreduce_grads(self, force_overlap=False, average=True, token_ids=None):
        """Sum the gradients over the ranks; average=True also divides them by the world size, average=False returns that
        factor (1 / W) for the caller to fold into its clip / optimizer pass (train_step does: one pass over the 0.5 GB
        gradient arena less).  With the flat arena on a GPU the all-reduce runs UNDER the backward:
        convdr_encoder_backward has only been enqueued when this is called, so one collective per encoder layer
        (28 MB of fp32 for roberta-base: large enough for the xGMI ring, 12 of them in flight behind each other) is
        queued on a communication stream behind that layer's completion events (convdr_backward_wait_layer), last
        layer first; embeddings + head follow the whole backward.  The compute stream waits for all of them at the
        end.  (force_overlap: run this path at world size 1 too -- the single-GPU test of the stream logic.)
        token_ids: this rank's embedded token ids (sparse_embedding: saves the pass that finds the non-zero rows)."""
        W = _world(self.group)
        scale = 1.0 / W
        wg = self._word_grad() if self.sparse_embedding else None

        def finish(tensors):
            if average and W > 1:
                for t in tensors:
                    t.div_(W)
            return 1.0 if average else scale
        if W == 1 and not force_overlap:
            return 1.0
        from .train import _flat_view
        grads = [p.grad for p in self.model.parameters() if p.grad is not None]
        flat = _flat_view(grads)
        buckets = self._layer_buckets(flat.numel()) if (flat is not None and flat.is_cuda) else None
        if buckets:
            m = self.model.module if hasattr(self.model, "module") else self.model
            if getattr(m.roberta, "_last_backward_arena", None) != flat.data_ptr():
                buckets = None      # accumulated gradients (see train._EncoderFn.backward): one collective after the backward
        self.last_path = "overlapped" if buckets else "single"   # (instrumentation for the tests)
        nbytes = sum(g.numel() for g in grads) * 4
        self.last_comm = {"dense_bytes_per_rank": int(nbytes), "sparse_embedding": False, "allreduce_dtype": self.allreduce_dtype or "fp32"}
        if wg is not None and flat is not None and wg.data_ptr() != flat.data_ptr():
            wg = None                                  # (the arena does not start with the word table: dense)
        if buckets:
            from . import _lib
            L = _lib.lib()
            cur = torch.cuda.current_stream(flat.device)
            # a stream that has been seen to run beside the compute stream (train._StreamSets: HIP's stream -> hardware queue
            # multiplexing decides whether the collectives overlap the backward or serialise with it; the step watchdog may
            # move the set between steps, so it is looked up per call)
            from .train import _aux_streams
            comm, works = _aux_streams(flat.device)[2], []
            with torch.cuda.device(flat.device), torch.cuda.stream(comm):
                for l in reversed(range(len(buckets))):
                    _lib.check(L.convdr_backward_wait_layer(l, comm.cuda_stream), "convdr_backward_wait_layer")
                    b, e = buckets[l]
                    works.append(self._reduce(flat[b:e], async_op=True))
                comm.wait_stream(cur)              # embeddings and head: complete only with the whole backward
                e0 = 0
                if wg is not None:                 # the word table as (ids, rows); position / type / LayerNorm stay dense
                    self._note_sparse(sparse_rows_allreduce(wg, self.group, token_ids), nbytes)
                    e0 = wg.numel()
                works.append(dist.all_reduce(flat[e0:buckets[0][0]], group=self.group, async_op=True))
                works.append(dist.all_reduce(flat[buckets[-1][1]:], group=self.group, async_op=True))
            for wk in works:
                if wk is not None:
                    wk.wait()                      # the compute stream waits; the host does not
            cur.wait_stream(comm)
            return finish([flat])
        if W == 1:
            return 1.0
        if flat is not None:                       # one arena, but not on a GPU (gloo tests): a single collective
            if wg is not None:
                self._note_sparse(sparse_rows_allreduce(wg, self.group, token_ids), nbytes)
                self._reduce(flat[wg.numel():])
            else:
                self._reduce(flat)
            return finish([flat])
        for g in grads:
            if wg is not None and g is wg:
                self._note_sparse(sparse_rows_allreduce(wg, self.group, token_ids), nbytes)
            else:
                self._reduce(g)
        return finish(grads)

    def _note_sparse(self, st, nbytes):
        self.last_comm.update(sparse_embedding=True, embedding_rows_this_rank=st["rows"], embedding_rows_padded=st["rows_max"],
                              embedding_bytes_gathered=st["bytes_gathered"], embedding_bytes_dense=st["bytes_dense"],
                              sparse_bytes_per_rank=int(nbytes - st["bytes_dense"] + st["bytes_gathered"]))
