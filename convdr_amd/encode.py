"""Corpus-encode loop: token cache -> passage embeddings -> per-rank block files.

Restates the reference driver loop
  /root/reference/drivers/gen_passage_embeddings.py:73-127 (InferenceEmbeddingFromStreamDataLoader),
  :131-169 (StreamInferenceDoc), :172-193 (generate_new_ann)
around the HIP encoder: record i belongs to rank i % world (utils/util.py:422-424); every rank writes
``passage__emb_p__data_obj_{rank}.pb`` (float32 [n, 768]) and ``passage__embid_p__data_obj_{rank}.pb``
(int64 [n], the record offsets i -- not pids) exactly as ``barrier_array_merge`` does (utils/util.py:108-111).

What changed for MI355X (results are identical, embeddings do not depend on batching):
  * the token cache is memory-mapped and each rank touches only its own records (the reference scans the
    whole file on every rank and builds python lists per record);
  * only the real tokens of each passage are computed (the reference pads every passage to
    --max_seq_length and computes the padding);
  * no per-batch device sync: ids go up and embeddings come back through pinned buffers asynchronously.
"""
import os

import numpy as np
import torch

from . import blocks


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def plan_batches(lens, batch_size, token_budget=None, align=1):
    """Cut a shard's records, IN ORDER, into batches of at most `batch_size` records and -- when `token_budget` is given
    -- at most `token_budget` tokens, each record counted as its length rounded up to `align` (a single longer record
    still forms a batch).  Returns [(start, stop)].
    The encoder computes packed rows, so its work per batch is the token count, not records x max length: a token
    budget keeps every launch at the size the GEMM tiles are tuned for (262,144 rows = 1,024 row tiles of 256) whatever
    the length mix of the corpus.  Embeddings do not depend on the batching (tests/test_encoder_gpu.py)."""
    lens = np.asarray(lens, dtype=np.int64)
    if align > 1:   # the encoder packs every sequence to a multiple of `align` rows: budget the rows it will compute
        lens = (lens + align - 1) // align * align
    n = len(lens)
    if token_budget is None:
        return [(s, min(s + batch_size, n)) for s in range(0, n, batch_size)]
    csum = np.concatenate([[0], np.cumsum(lens)])
    out, s = [], 0
    while s < n:
        # largest e with csum[e] - csum[s] <= budget, at least one record, at most batch_size
        e = int(np.searchsorted(csum, csum[s] + token_budget, side="right")) - 1
        e = min(max(e, s + 1), s + batch_size, n)
        out.append((s, e))
        s = e
    return out


def encode_shard(model, cache, rank=0, world=1, batch_size=1024, is_query_inference=False, max_seq_length=None,
                 progress=None, token_budget=None):
    """-> (embedding float32 [n, D] numpy, embedding2id int64 [n]) for this rank's records, in record order.
    token_budget: see plan_batches (batch_size then only caps the record count and sizes the staging buffers)."""
    tower_call = _embed_fn(model, is_query_inference)
    dev = next(model.parameters()).device
    idx = blocks.shard_indices(len(cache), world, rank)
    L = cache.seq_len if max_seq_length is None else min(cache.seq_len, int(max_seq_length))
    lens_all = np.minimum(cache.lengths(idx), L).astype(np.int32) if len(idx) else np.zeros(0, np.int32)
    out = None
    on_gpu = dev.type == "cuda"       # (the encoder itself is GPU-only; a host "device" only occurs with a stand-in tower in
    pin = (lambda t: t.pin_memory()) if on_gpu else (lambda t: t)     #  the multi-process CPU tests of the shard / file logic)
    stage = [pin(torch.empty((batch_size, L), dtype=torch.int32)) for _ in range(2)]
    events = [None, None]
    for bi, (s, e) in enumerate(plan_batches(lens_all, batch_size, token_budget, align=8)):
        sel = idx[s:e]
        lens = lens_all[s:e]
        n, lmax = len(sel), int(lens.max())
        buf = stage[bi & 1]
        if events[bi & 1] is not None:
            events[bi & 1].synchronize()                  # its previous H2D copy has been consumed
        np.take(cache.ids[:, :lmax], sel, axis=0, out=buf.numpy()[:n, :lmax])
        ids = buf[:n, :lmax].to(dev, non_blocking=True) if on_gpu else buf[:n, :lmax].clone()
        if on_gpu:
            ev = torch.cuda.Event()
            ev.record()
            events[bi & 1] = ev
        with torch.no_grad():
            emb = tower_call(ids, lens)
        if out is None:
            out = pin(torch.empty((len(idx), emb.shape[1]), dtype=torch.float32))
        out[s:s + n].copy_(emb, non_blocking=True)
        if progress:
            progress(n)
    if on_gpu:
        torch.cuda.synchronize(dev)
        from .train import check_status
        check_status(model)  # token ids outside the embedding table: IndexError like the reference's lookup (models.py:141)
    if out is None:
        return np.zeros((0, 768), np.float32), idx
    return out.numpy(), idx


def _embed_fn(model, is_query):
    """Resolve (tower, head) of the reference model classes for the int32 / host-lengths fast path."""
    m = model.module if hasattr(model, "module") else model
    if hasattr(m, "roberta"):
        tower, head = m.roberta, (m.embeddingHead, m.norm)
    elif hasattr(m, "question_model"):
        tower, head = (m.question_model if is_query else m.ctx_model), None
    else:
        raise TypeError("unsupported model class %s" % type(m).__name__)
    return lambda ids, lens: tower.embed(ids, None, head=head, seq_lens=lens)


def StreamInferenceDoc(args, model, cache, prefix="passage_", is_query_inference=False, batch_size=None):
    """Encode this rank's shard and write its two block files; same file names / contents as the reference
    (`merge=False` path, gen_passage_embeddings.py:156-167)."""
    dist = _dist()
    rank = dist.get_rank() if dist else int(getattr(args, "rank", 0) or 0)
    world = dist.get_world_size() if dist else int(getattr(args, "world_size", 1) or 1)
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
    if dist:
        dist.barrier()
    emb, embid = encode_shard(model, cache, rank, world, batch_size or getattr(args, "per_gpu_eval_batch_size", 64),
                              is_query_inference, getattr(args, "max_seq_length", None))
    blocks.dump_block(os.path.join(args.output_dir, "%s_emb_p__data_obj_%d.pb" % (prefix, rank)), emb)
    blocks.dump_block(os.path.join(args.output_dir, "%s_embid_p__data_obj_%d.pb" % (prefix, rank)), embid)
    if dist:
        dist.barrier()
    return emb, embid


def generate_new_ann(args, model):
    """gen_passage_embeddings.py:172-193 with an already-loaded model."""
    with blocks.TokenCache(os.path.join(args.data_dir, "passages")) as cache:
        return StreamInferenceDoc(args, model, cache, "passage_", is_query_inference=False)
