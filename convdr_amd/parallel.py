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


class DataParallelStudent:
    """Gradient synchronisation for one-process-per-GPU training of the student."""

    def __init__(self, model, group=None, broadcast=True):
        self.model, self.group = model, group
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

    def allreduce_grads(self, force_overlap=False, average=True):
        """Sum the gradients over the ranks; average=True also divides them by the world size, average=False returns that
        factor (1 / W) for the caller to fold into its clip / optimizer pass (train_step does: one pass over the 0.5 GB
        gradient arena less).  With the flat arena on a GPU the all-reduce runs UNDER the backward:
        convdr_encoder_backward has only been enqueued when this is called, so one collective per encoder layer
        (28 MB of fp32 for roberta-base: large enough for the xGMI ring, 12 of them in flight behind each other) is
        queued on a communication stream behind that layer's completion events (convdr_backward_wait_layer), last
        layer first; embeddings + head follow the whole backward.  The compute stream waits for all of them at the
        end.  (force_overlap: run this path at world size 1 too -- the single-GPU test of the stream logic.)"""
        W = _world(self.group)
        scale = 1.0 / W

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
                    works.append(dist.all_reduce(flat[b:e], group=self.group, async_op=True))
                comm.wait_stream(cur)              # embeddings and head: complete only with the whole backward
                works.append(dist.all_reduce(flat[:buckets[0][0]], group=self.group, async_op=True))
                works.append(dist.all_reduce(flat[buckets[-1][1]:], group=self.group, async_op=True))
            for wk in works:
                wk.wait()                          # the compute stream waits; the host does not
            cur.wait_stream(comm)
            return finish([flat])
        if W == 1:
            return 1.0
        if flat is not None:                       # one arena, but not on a GPU (gloo tests): a single collective
            dist.all_reduce(flat, group=self.group)
            return finish([flat])
        for g in grads:
            dist.all_reduce(g, group=self.group)
        return finish(grads)
