"""KD + ranking training step on MI355X.

Mirrors /root/reference/drivers/run_convdr_train.py:101-193 (the step body of ``train``) and
/root/reference/utils/dpr_utils.py:80-87 (``get_optimizer``): student forward + backward, MSE KD loss, optional
ranking cross-entropy over 1 positive + ``num_negatives`` documents, global-norm clip, HF-semantics AdamW and the
linear warm-up/decay schedule -- every tensor operation runs in libconvdr_hip.so (csrc/train*.hip).

Integration with torch autograd: ``encoder_autograd`` is a ``torch.autograd.Function`` whose forward keeps the
activations inside a device workspace and whose backward runs the hand-written backward kernels and returns the
parameter gradients (views of one flat fp32 buffer), so ``loss.backward()``, ``.grad``, ``zero_grad`` and any
``torch.optim`` optimizer keep working -- as do the fused ``clip_grad_norm_`` / ``AdamW`` below.

Dropout (run_convdr_train.py:107 ``model.train()``: hidden / attention-probability dropout inside the HF encoder, 0.1 in
the released configs): torch's RNG stream cannot be reproduced by any other implementation, so the masks are a documented
counter-based function of (seed, site, layer, element) -- csrc/dropout.hpp, restated bit for bit in oracle/dropout.py.  A
forward draws ``seed`` from ``model.dropout_seed`` (default: torch.initial_seed()) plus a per-call counter, the backward
regenerates the masks from it; parity tests replay the same seed through the oracle.
"""
import collections
import ctypes as C
import logging
import math
import os
import weakref

import numpy as np
import torch

from . import _lib


# --------------------------------------------------------------------------------------------
# parameter order of the flat gradient buffer (per layer: q,k,v weights adjacent -> one [3H, H] region)
# --------------------------------------------------------------------------------------------
def _tower_params(tower, head):
    e = tower.embeddings
    out = [e.word_embeddings.weight, e.position_embeddings.weight, e.token_type_embeddings.weight,
           e.LayerNorm.weight, e.LayerNorm.bias]
    for ly in tower.encoder.layer:
        s = ly.attention.self
        out += [s.query.weight, s.key.weight, s.value.weight, s.query.bias, s.key.bias, s.value.bias,
                ly.attention.output.dense.weight, ly.attention.output.dense.bias,
                ly.attention.output.LayerNorm.weight, ly.attention.output.LayerNorm.bias,
                ly.intermediate.dense.weight, ly.intermediate.dense.bias,
                ly.output.dense.weight, ly.output.dense.bias, ly.output.LayerNorm.weight, ly.output.LayerNorm.bias]
    if head is not None:
        out += [head[0].weight, head[0].bias, head[1].weight, head[1].bias]
    return out


def flatten_parameters(model):
    """Re-home the student's trainable parameters into ONE contiguous fp32 arena (same order as the gradient arena
    that convdr_encoder_backward fills), keeping every nn.Parameter object and name: ``p.data`` becomes a view.
    Enables: a single fused AdamW launch, a single cast for the bf16 weight copies, one all-reduce for DDP.
    Call after ``model.to(device)`` and before creating the optimizer."""
    m = model.module if hasattr(model, "module") else model
    if not hasattr(m, "roberta"):
        return None                      # BiEncoder: two towers; the generic per-parameter path is used
    tower, head = m.roberta, (m.embeddingHead, m.norm)
    params = _tower_params(tower, head)
    sizes = [p.numel() for p in params]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    dev = params[0].device
    P = torch.empty(int(offs[-1]), dtype=torch.float32, device=dev)
    off = {}
    with torch.no_grad():
        for p, o, n in zip(params, offs[:-1], sizes):
            view = P[int(o):int(o) + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
            off[id(p)] = int(o)
    tower._flat = {"P": P, "off": off, "w0": int(offs[5]), "head": head[0], "params": params}
    tower._packed = None
    _ARENA_OWNERS[P.data_ptr()] = weakref.ref(tower)
    return tower._flat


_ARENA_OWNERS = {}      # flat parameter arena (data_ptr) -> weak reference to the tower whose parameters are its views


def _arena_owner(P):
    ref = _ARENA_OWNERS.get(P.data_ptr())
    tower = ref() if ref is not None else None
    if ref is not None and tower is None:
        del _ARENA_OWNERS[P.data_ptr()]
    return tower


_PACKT_FROM_BF16 = os.environ.get("CONVDR_PACKT_FROM_BF16", "1") != "0"      # 0: from the fp32 master weights (A/B)


def _packed_t(tower, head):
    """Transposed bf16 weights for the data-gradient GEMMs (cached with the forward packing)."""
    c, w, keep = tower.packed(head)
    cache = getattr(tower, "_packed_t", None)
    if cache is not None and cache[0] is keep:
        return cache[1], cache[2]
    L = _lib.lib()
    dev = tower.embeddings.word_embeddings.weight.device
    flat = getattr(tower, "_flat", None)
    if flat is not None and flat["head"] is (None if head is None else head[0]):
        mats = []
        for ly in tower.encoder.layer:
            s = ly.attention.self
            mats += [(s.query.weight, 3), (ly.attention.output.dense.weight, 1), (ly.intermediate.dense.weight, 1),
                     (ly.output.dense.weight, 1)]
        if head is not None:
            mats.append((head[0].weight, 1))
        n = (C.c_int32 * len(mats))(*[p.shape[0] * mul for p, mul in mats])
        k = (C.c_int32 * len(mats))(*[p.shape[1] for p, _ in mats])
        src = (C.c_int64 * len(mats))(*[flat["off"][id(p)] for p, _ in mats])
        dsts = np.concatenate([[0], np.cumsum([int(a) * int(b) for a, b in zip(n, k)])]).astype(np.int64)
        dst = (C.c_int64 * len(mats))(*dsts[:-1].tolist())
        T = flat.get("Pt")
        if T is None:
            T = flat["Pt"] = torch.empty(int(dsts[-1]), dtype=torch.bfloat16, device=dev)
        with torch.cuda.device(dev):
            Pb = flat.get("Pb")
            if Pb is not None and _PACKT_FROM_BF16:
                # tower.packed() above made sure the bf16 copy of the weight range is current (cast, or rewritten by the optimizer
                # step itself): transposing THAT moves a third less data than re-rounding the fp32 master weights, same bits
                srcb = (C.c_int64 * len(mats))(*[flat["off"][id(p)] - flat["w0"] for p, _ in mats])
                _lib.check(L.convdr_pack_transposed_bf16(_lib.ptr(Pb), len(mats), srcb, n, k, dst, _lib.ptr(T), _lib.stream_ptr()),
                           "convdr_pack_transposed_bf16")
            else:
                _lib.check(L.convdr_pack_transposed(_lib.ptr(flat["P"]), len(mats), src, n, k, dst, _lib.ptr(T), _lib.stream_ptr()),
                           "convdr_pack_transposed")
        arr = (_lib.LayerWeightsT * len(tower.encoder.layer))()
        for i in range(len(tower.encoder.layer)):
            arr[i].wqkv_t, arr[i].wo_t, arr[i].w1_t, arr[i].w2_t = [T.data_ptr() + 2 * int(dsts[4 * i + j]) for j in range(4)]
        head_t = T.data_ptr() + 2 * int(dsts[-2]) if head is not None else None
        tower._packed_t = (keep, arr, head_t, [T, arr])
        return arr, head_t
    hold = []

    def tr(t):
        t = t.detach().float().contiguous()
        n, k = t.shape
        o = torch.empty((k, n), dtype=torch.bfloat16, device=dev)
        _lib.check(L.convdr_transpose_f32_bf16(_lib.ptr(t), n, k, _lib.ptr(o), _lib.stream_ptr()), "convdr_transpose_f32_bf16")
        hold.append((t, o))
        return o.data_ptr()

    with torch.cuda.device(dev):
        arr = (_lib.LayerWeightsT * len(tower.encoder.layer))()
        for i, ly in enumerate(tower.encoder.layer):
            s = ly.attention.self
            arr[i].wqkv_t = tr(torch.cat([s.query.weight, s.key.weight, s.value.weight], 0))
            arr[i].wo_t = tr(ly.attention.output.dense.weight)
            arr[i].w1_t = tr(ly.intermediate.dense.weight)
            arr[i].w2_t = tr(ly.output.dense.weight)
        head_t = tr(head[0].weight) if head is not None else None
    hold.append(arr)
    tower._packed_t = (keep, arr, head_t, hold)
    return arr, head_t


class _WsEntry:
    """One activation workspace of a tower.  `owner` is a weak reference to the token held by the autograd context whose
    saved activations live in it: the buffer may be handed to another forward only once that context has run its backward
    (token.done) or has been dropped without one.  `gen` counts hand-outs, so a backward that finds its workspace re-used
    (retain_graph + a second backward after a newer forward) fails loudly instead of reading foreign activations."""
    __slots__ = ("ws", "owner", "gen")

    def __init__(self, ws):
        self.ws, self.owner, self.gen = ws, None, 0


class _WsToken:
    __slots__ = ("done", "__weakref__")

    def __init__(self):
        self.done = False


def _take_workspace(tower, need, dev):
    """A workspace of >= need bytes that no pending backward still reads (one shared buffer per tower let the
    2nd / 3rd differentiable forward of NLL.forward(q, a, b) overwrite the activations saved by the first)."""
    pool = tower.__dict__.setdefault("_train_ws_pool", [])
    free = []
    for ent in pool:
        tok = ent.owner() if ent.owner is not None else None
        if tok is None or tok.done:
            free.append(ent)
    for ent in free:
        if ent.ws.numel() >= need and ent.ws.device == dev:
            break
    else:
        for ent in free:                       # too small / wrong device: let the allocator have it back
            pool.remove(ent)
        ent = _WsEntry(torch.empty(int(need * 1.1), dtype=torch.uint8, device=dev))
        pool.append(ent)
    token = _WsToken()
    ent.owner, ent.gen = weakref.ref(token), ent.gen + 1
    return ent, token


_PIN_RINGS = {}
_PIN_CAPTURED = object()     # marks a ring slot that a captured copy node reads at every replay (see _pinned_upload)
_PIN_SLOTS = 32
_PIN_RETIRED = []


def _pinned_upload(arr, dev):
    """Small host array -> device tensor without blocking the host on the stream (pageable copies do).
    The pinned staging comes from a ring of reusable slots per size class (``tensor.pin_memory()`` per call costs a host
    allocator round trip -- and, inside a hipGraph capture, its event queries are illegal); a slot is rewritten only
    after the copy that last read it has completed (an event per slot, 32 uploads later: never a real wait)."""
    nbytes = max(int(arr.nbytes), 1)
    cls = 1 << max(10, (nbytes - 1).bit_length())
    ring = _PIN_RINGS.get(cls)
    if ring is None:
        ring = _PIN_RINGS[cls] = {"buf": torch.empty((_PIN_SLOTS, cls), dtype=torch.uint8).pin_memory(), "ev": [None] * _PIN_SLOTS, "next": 0}
    dev = torch.device(dev)
    capturing = torch.cuda.is_current_stream_capturing()
    # A captured copy node reads its pinned source at every REPLAY: the slot it took must never be handed out again.  It is
    # marked as owned by the capture (pinning a fresh buffer here is not an option: host allocations are illegal while a
    # stream captures); a ring whose slots have all gone that way is replaced -- outside a capture -- by a fresh one.
    for _ in range(_PIN_SLOTS):
        i = ring["next"]
        ring["next"] = (i + 1) % _PIN_SLOTS
        if ring["ev"][i] is not _PIN_CAPTURED:
            break
    else:
        if capturing:
            raise RuntimeError("_pinned_upload: every staging slot of this size belongs to a captured graph")
        _PIN_RETIRED.append(ring)       # (its slots are still read by the graphs that captured them)
        ring = _PIN_RINGS[cls] = {"buf": torch.empty((_PIN_SLOTS, cls), dtype=torch.uint8).pin_memory(), "ev": [None] * _PIN_SLOTS, "next": 1}
        i = 0
    if capturing:
        ring["ev"][i] = _PIN_CAPTURED
        host = ring["buf"][i, :nbytes].view(torch.from_numpy(arr).dtype).view(arr.shape)
        host.copy_(torch.from_numpy(arr))
        with torch.cuda.device(dev):
            return host.to(dev, non_blocking=True)
    if ring["ev"][i] is not None:
        ring["ev"][i].synchronize()
    host = ring["buf"][i, :nbytes].view(torch.from_numpy(arr).dtype).view(arr.shape)
    host.copy_(torch.from_numpy(arr))
    # the slot's guard event is recorded on the stream the copy was issued on: the current stream OF THE TARGET DEVICE (not
    # of whatever device happens to be current in the calling thread)
    with torch.cuda.device(dev):
        out = host.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
    ring["ev"][i] = ev
    return out


# ---- status word of the encoder forward (include/convdr_hip.h: CONVDR_ENC_STATUS_*) -------------------------------
# With caller-provided host lengths a forward has no device -> host round trip, so nothing on the host can look at the
# token ids.  The packing kernel does (clamps + flags); the flag word is copied to a pinned slot behind every forward
# and looked at without waiting: at the start of the tower's next forward, and by `check_status(model)` which the
# loops call at their own sync points (encode.py / inference.py after the embeddings came back).  So a bad batch raises
# the reference's IndexError at the latest one call later, never reads out of bounds, and costs no sync.
_STATUS_RING = 64


def _status_post(tower, ws):
    if torch.cuda.is_current_stream_capturing():       # (hipGraph capture of a step: no host-visible side effects)
        return
    st = tower.__dict__.get("_status")
    if st is None:
        st = tower.__dict__["_status"] = {"ring": torch.zeros(_STATUS_RING, dtype=torch.int32).pin_memory(), "pending": [], "next": 0}
    if len(st["pending"]) >= _STATUS_RING - 1:
        _status_poll(tower, sync=True)
    slot = st["next"]
    st["next"] = (slot + 1) % _STATUS_RING
    st["ring"][slot:slot + 1].copy_(ws[:4].view(torch.int32), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    st["pending"].append((slot, ev))


def _status_poll(tower, sync=False):
    st = tower.__dict__.get("_status")
    if not st or not st["pending"] or torch.cuda.is_current_stream_capturing():
        return
    flags, left = 0, []
    for slot, ev in st["pending"]:
        if sync:
            ev.synchronize()
        if sync or ev.query():
            flags |= int(st["ring"][slot])
        else:
            left.append((slot, ev))
    st["pending"] = left
    if flags & 1:
        raise IndexError("token id out of range for the %d-row word-embedding table (flagged by the packing kernel; the "
                         "embeddings of that batch are meaningless)" % tower.embeddings.word_embeddings.num_embeddings)
    if flags & 2:
        raise ValueError("seq_lens does not match the number of unmasked tokens of a sequence (or exceeds the padded length)")
    if flags & 4:
        raise ValueError("every sequence needs attention_mask[:, 0] == 1")


def check_status(model, sync=True):
    """Raise what the reference's embedding lookup would have raised (IndexError) for any forward of `model`'s towers
    enqueued so far whose packing kernel flagged its inputs.  sync=True waits for those forwards."""
    m = model.module if hasattr(model, "module") else model
    for mod in m.modules():
        if "_status" in mod.__dict__:
            _status_poll(mod, sync=sync)


def _lens_and_check(ids, mask, vocab, seq_lens=None):
    """(device int32 lens, host int32 lens).  Without caller-provided host lengths this is the one device -> host
    round trip of a forward; the largest / smallest token id ride along and are validated like the reference's
    embedding lookup would (IndexError instead of an out-of-bounds read)."""
    dev = ids.device
    if seq_lens is not None:
        lens_host = np.ascontiguousarray(np.asarray(seq_lens, dtype=np.int32))
        return _pinned_upload(lens_host, dev), lens_host
    lens_dev = mask.sum(1).to(torch.int32)
    stats = torch.cat([lens_dev, ids.max().reshape(1).to(torch.int32), ids.min().reshape(1).to(torch.int32),
                       mask[:, 0].min().reshape(1).to(torch.int32)]).cpu().numpy()
    lens_host = np.ascontiguousarray(stats[:-3])
    if stats[-3] >= vocab or stats[-2] < 0:
        raise IndexError("token id out of range for the %d-row word-embedding table (min %d, max %d)" % (vocab, stats[-2], stats[-3]))
    if stats[-1] == 0:
        raise ValueError("every sequence needs attention_mask[:, 0] == 1")
    return lens_dev, lens_host


_FRESH_BWD = os.environ.get("CONVDR_FRESH_BWD", "1") != "0"      # 0: the whole arena filled + convdr_encoder_backward (rounds 2-5; A/B)
_POISON_FRESH_ARENA = False      # tests: NaN in every gradient the backward is supposed to STORE (convdr_encoder_backward_fresh)


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tower, head, input_ids, attention_mask, seq_lens, dropout, *params):
        L_ = _lib.lib()
        ids = input_ids.long().contiguous()
        mask = attention_mask.long().contiguous()
        if ids.device.type != "cuda":
            raise _lib.ConvdrError("encoder inputs must be CUDA tensors (no CPU fallback)")
        B, L = ids.shape
        dev = ids.device
        _status_poll(tower)
        seq_lens_dev, lens_host = _lens_and_check(ids, mask, tower.embeddings.word_embeddings.num_embeddings, seq_lens)
        if lens_host.min() < 1:
            raise ValueError("every sequence needs at least one unmasked token")
        tower.check_positions(int(lens_host.max()))
        cu_host = np.zeros(B + 1, np.int32)
        np.cumsum((lens_host + 7) // 8 * 8, out=cu_host[1:])
        rows, max_len = int(cu_host[-1]), int(lens_host.max())
        cu = _pinned_upload(cu_host, dev)
        with torch.cuda.device(dev):
            c, w, _keep = tower.packed(head)
            out = torch.empty((B, c.out_dim or c.hidden), dtype=torch.float32, device=dev)
            need = L_.convdr_encoder_train_workspace_bytes(C.byref(c), rows, B)
            ent, token = _take_workspace(tower, need, dev)
            ws = ent.ws
            drop = None if dropout is None else C.byref(_lib.Dropout(float(dropout[0]), float(dropout[1]), int(dropout[2]) & 0xffffffff))
            # the transposed weight copies are needed by the backward only: pack them on the side stream, under this
            # forward (one LDS-free launch, train_kernels.hpp: k_transpose_bf16_batch; behind the forward instead it measured
            # the same, 9.03 vs 9.00 ms: profiles/r06_ab_transpose.txt).  The side
            # stream forks off BEFORE the forward is enqueued -- the weights are final in stream order here; forking after it
            # (rounds 2-4) made the packing wait for the whole forward and the loss wait for the packing: 0.08 ms per step
            main = torch.cuda.current_stream()
            # (round 5: on the third auxiliary stream instead -- in train_step the teacher's forward is enqueued on this stream
            #  BEFORE the student's, so the packing only starts when the teacher is done -- measured no different: 9.61-9.89 vs
            #  9.61-9.79 ms, three alternations)
            side = _side_stream(dev)
            side.wait_event(main.record_event())
            _lib.check(L_.convdr_encoder_train_forward(C.byref(c), C.byref(w), _lib.ptr(ids), 0, _lib.ptr(mask), B, L,
                                                       _lib.ptr(cu), _lib.ptr(seq_lens_dev), rows, max_len, _lib.ptr(ws),
                                                       ws.numel(), _lib.ptr(out), drop, _lib.stream_ptr()),
                       "convdr_encoder_train_forward")
            _status_post(tower, ws)
            # the gradient arena of this forward's backward: allocated here, and its EMBEDDING prefix (the scatter-added tables +
            # the embedding LayerNorm: 0.15 of the 0.5 GB for roberta-base) zeroed on the side stream under the forward; every other
            # gradient is stored, not accumulated, by convdr_encoder_backward_fresh (rounds 2-5 filled all of it: 0.06 ms per step
            # wherever the fill was put, and the weight-gradient tiles read the zeros back)
            ctx.grad_arena = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
            ctx.fresh_prefix = sum(p.numel() for p in params[:5])
            # (the arena comes from the main stream's pool but is first written on `side`: if this graph is dropped without a
            #  backward the block must not return to the main pool while the fill may still be pending)
            ctx.grad_arena.record_stream(side)
            with torch.cuda.stream(side):
                ctx.packed_t = _packed_t(tower, head)
                if _POISON_FRESH_ARENA:
                    ctx.grad_arena.fill_(float("nan"))
                if _FRESH_BWD:
                    ctx.grad_arena[:ctx.fresh_prefix].zero_()
                else:
                    ctx.grad_arena.zero_()
                ctx.packed_t_ready = side.record_event()
        ctx.tower, ctx.head, ctx.dropout = tower, head, dropout
        ctx.packed = (c, w, _keep)           # the weights cannot change between a forward and its backward
        ctx.saved = (cu, seq_lens_dev, B, rows, max_len, ent, ent.gen, token)
        ctx.shapes = [p.shape for p in params]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        L_ = _lib.lib()
        tower, head = ctx.tower, ctx.head
        cu, seq_lens, B, rows, max_len, ent, gen, token = ctx.saved
        if ent.gen != gen:
            raise RuntimeError("the activation workspace of this forward has been handed to a newer forward "
                               "(a second backward through the same graph after retain_graph is not supported)")
        ws = ent.ws
        dev = grad_out.device
        go = grad_out.float().contiguous()
        sizes = [int(np.prod(s)) for s in ctx.shapes]
        flat, ctx.grad_arena = ctx.grad_arena, None            # (prefix zeroed under the forward; the wait on packed_t_ready below covers it)
        if flat is None:
            flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        views = [v.view(s) if len(s) != 1 else v for v, s in zip(flat.split_with_sizes(sizes), ctx.shapes)]
        ptr = [v.data_ptr() for v in views]
        # the arena this call's kernels write: DataParallelStudent overlaps the all-reduce with the backward only when the
        # parameters' .grad ARE this arena (grads were None: autograd adopts the views); after an accumulation the
        # .grad buffers are completed by autograd's own add kernels, which the per-layer events know nothing about
        tower._last_backward_arena = flat.data_ptr()
        nl = len(tower.encoder.layer)
        with torch.cuda.device(dev):
            c, w, _keep = ctx.packed
            torch.cuda.current_stream().wait_event(ctx.packed_t_ready)
            wt, head_t = ctx.packed_t
            lg = (_lib.LayerGrads * nl)()
            for i in range(nl):
                b = 5 + 16 * i
                g = lg[i]
                g.wqkv, g.bqkv = ptr[b], ptr[b + 3]           # q,k,v weights / biases are adjacent
                g.wo, g.bo, g.ln1_g, g.ln1_b = ptr[b + 6], ptr[b + 7], ptr[b + 8], ptr[b + 9]
                g.w1, g.b1, g.w2, g.b2 = ptr[b + 10], ptr[b + 11], ptr[b + 12], ptr[b + 13]
                g.ln2_g, g.ln2_b = ptr[b + 14], ptr[b + 15]
            gr = _lib.EncoderGrads()
            gr.word_emb, gr.pos_emb, gr.type_emb, gr.emb_ln_g, gr.emb_ln_b = ptr[0], ptr[1], ptr[2], ptr[3], ptr[4]
            gr.layers = C.cast(lg, C.POINTER(_lib.LayerGrads))
            if head is not None:
                b = 5 + 16 * nl
                gr.head_w, gr.head_b, gr.head_ln_g, gr.head_ln_b = ptr[b], ptr[b + 1], ptr[b + 2], ptr[b + 3]
            dropout = ctx.dropout
            drop = None if dropout is None else C.byref(_lib.Dropout(float(dropout[0]), float(dropout[1]), int(dropout[2]) & 0xffffffff))
            # (`flat` is this call's own arena: nothing else has written it -- autograd adds the views to existing .grad itself)
            _lib.check((L_.convdr_encoder_backward_fresh if _FRESH_BWD else L_.convdr_encoder_backward)(C.byref(c), C.byref(w), wt, _lib.ptr(cu), _lib.ptr(seq_lens),
                                                        C.c_void_p(head_t) if head_t else None, B, rows, max_len, _lib.ptr(ws),
                                                        ws.numel(), _lib.ptr(go), C.byref(gr), drop, _lib.stream_ptr()),
                       "convdr_encoder_backward_fresh")
        token.done = True      # (stream order: a later forward on this stream overwrites the workspace after these kernels)
        return (None, None, None, None, None, None) + tuple(views)


def _mix32(a):
    """csrc/dropout.hpp: drop_mix32 (per-call seed derivation only; the masks themselves are generated on the device)."""
    a &= 0xffffffff
    a = ((a + 0x7ed55d16) + (a << 12)) & 0xffffffff
    a = ((a ^ 0xc761c23c) ^ (a >> 19)) & 0xffffffff
    a = ((a + 0x165667b1) + (a << 5)) & 0xffffffff
    a = ((a + 0xd3a2646c) ^ (a << 9)) & 0xffffffff
    a = ((a + 0xfd7046c5) + (a << 3)) & 0xffffffff
    a = ((a ^ 0xb55a4f09) ^ (a >> 16)) & 0xffffffff
    return a


def next_dropout_seed(model):
    """Seed of the next differentiable forward of `model`: mix32(base + number of forwards so far), base =
    ``model.dropout_seed`` (set it for reproducible / replayable masks) or torch.initial_seed().  Exposed so that a parity
    test can read the seed a step will use (``peek=True`` semantics: call dropout_seed_of(model, n))."""
    n = model.__dict__.get("_dropout_calls", 0)
    model.__dict__["_dropout_calls"] = n + 1
    return dropout_seed_of(model, n)


def dropout_seed_of(model, call_index):
    base = getattr(model, "dropout_seed", None)
    if base is None:
        base = torch.initial_seed()
    return _mix32((int(base) + int(call_index)) & 0xffffffff)


def encoder_autograd(model, tower, head, input_ids, attention_mask, seq_lens=None):
    """Differentiable embeddings of `tower` (+ optional (Linear, LayerNorm) head).
    seq_lens: optional HOST int array of the rows' token counts (right padding); with it the forward has no device ->
    host round trip, so the host can run a whole step ahead of the GPU."""
    cfg = tower.config
    dropout = None
    p_h, p_a = float(cfg.hidden_dropout_prob or 0.0), float(cfg.attention_probs_dropout_prob or 0.0)
    if model.training and (p_h > 0.0 or p_a > 0.0):
        dropout = (p_h, p_a, next_dropout_seed(model))
        model.__dict__["_last_dropout"] = dropout          # (instrumentation: what the most recent forward used)
    return _EncoderFn.apply(tower, head, input_ids, attention_mask, seq_lens, dropout, *_tower_params(tower, head))


# --------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------
class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, t):
        s32, t32 = s.float().contiguous(), t.float().contiguous()
        loss = torch.empty((), dtype=torch.float32, device=s.device)
        ds = torch.empty_like(s32)
        with torch.cuda.device(s.device):
            _lib.check(_lib.lib().convdr_mse_fwd_bwd(_lib.ptr(s32), _lib.ptr(t32), s32.numel(), 1.0, _lib.ptr(loss),
                                                     _lib.ptr(ds), _lib.stream_ptr()), "convdr_mse_fwd_bwd")
        ctx.save_for_backward(ds)
        return loss

    @staticmethod
    def backward(ctx, g):
        (ds,) = ctx.saved_tensors
        return ds * g, None


def _mse_value_and_grad(student_embs, teacher_embs):
    """(MSE, d MSE / d student) in the one launch _MSE.forward makes, without the autograd node: train_step seeds the
    encoder's backward with the gradient directly when the KD term is the whole loss."""
    s32, t32 = student_embs.detach().float().contiguous(), teacher_embs.detach().float().contiguous()
    loss = torch.empty((), dtype=torch.float32, device=s32.device)
    ds = torch.empty_like(s32)
    with torch.cuda.device(s32.device):
        _lib.check(_lib.lib().convdr_mse_fwd_bwd(_lib.ptr(s32), _lib.ptr(t32), s32.numel(), 1.0, _lib.ptr(loss), _lib.ptr(ds),
                                                 _lib.stream_ptr()), "convdr_mse_fwd_bwd")
    return loss, ds


def mse_loss(student_embs, teacher_embs):
    """nn.MSELoss() (run_convdr_train.py:460, :115); gradient flows to the student only."""
    return _MSE.apply(student_embs, teacher_embs.detach())


class _RankCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, embs, docs):
        e32, d32 = embs.float().contiguous(), docs.float().contiguous()
        B, K, E = d32.shape
        per = torch.empty(B, dtype=torch.float32, device=embs.device)
        de = torch.empty_like(e32)
        with torch.cuda.device(embs.device):
            _lib.check(_lib.lib().convdr_rank_ce_fwd_bwd(_lib.ptr(e32), _lib.ptr(d32), B, K, E, 1.0, _lib.ptr(per),
                                                         _lib.ptr(de), 0, _lib.stream_ptr()), "convdr_rank_ce_fwd_bwd")
        ctx.save_for_backward(de)
        return per.mean()

    @staticmethod
    def backward(ctx, g):
        (de,) = ctx.saved_tensors
        return de * g, None


def ranking_loss(embs, pos_and_negs_embeddings):
    """CrossEntropy(logits, 0) with logits[b, k] = <embs[b], docs[b, k]> (run_convdr_train.py:160-170);
    docs [B, K, E] come from the frozen teacher (no gradient)."""
    return _RankCE.apply(embs, pos_and_negs_embeddings.detach())


class _PairNLL(torch.autograd.Function):
    """NLL.forward's triple branch / NLL_MultiChunk's MaxP form (convdr_pair_nll_fwd_bwd): loss and the gradients of all three
    inputs in one launch."""

    @staticmethod
    def forward(ctx, q, a, b, bias_a, bias_b):
        q32, a32, b32 = q.float().contiguous(), a.float().contiguous(), b.float().contiguous()
        B, E = q32.shape
        Cn = a32.shape[1] if a32.dim() == 3 else 1
        ba = None if bias_a is None else bias_a.float().contiguous()
        bb = None if bias_b is None else bias_b.float().contiguous()
        per = torch.empty(B, dtype=torch.float32, device=q.device)
        dq, da, db = torch.empty_like(q32), torch.empty_like(a32), torch.empty_like(b32)
        with torch.cuda.device(q.device):
            _lib.check(_lib.lib().convdr_pair_nll_fwd_bwd(_lib.ptr(q32), _lib.ptr(a32), _lib.ptr(b32), _lib.ptr(ba), _lib.ptr(bb), B, Cn,
                                                          E, 1.0, _lib.ptr(per), _lib.ptr(dq), _lib.ptr(da), _lib.ptr(db),
                                                          _lib.stream_ptr()), "convdr_pair_nll_fwd_bwd")
        ctx.save_for_backward(dq, da, db)
        return per.mean()

    @staticmethod
    def backward(ctx, g):
        dq, da, db = ctx.saved_tensors
        return dq * g, da * g, db * g, None, None


def pairwise_nll(q_embs, a_embs, b_embs, bias_a=None, bias_b=None):
    """mean_i -log_softmax([s_a, s_b])[0] with s_x = <q, x> (a / b [B, E]) or max over chunks of <q, x_c> + bias (a / b
    [B, C, E], MaxP): models.py:66-75 and :92-126."""
    return _PairNLL.apply(q_embs, a_embs, b_embs, bias_a, bias_b)


class _InBatchCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, embs, docs_all, pos):
        e32, d32 = embs.float().contiguous(), docs_all.float().contiguous()
        B, (N, E) = e32.shape[0], d32.shape
        p32 = pos.to(torch.int32).contiguous()
        per = torch.empty(B, dtype=torch.float32, device=embs.device)
        de = torch.empty_like(e32)
        with torch.cuda.device(embs.device):
            _lib.check(_lib.lib().convdr_inbatch_ce_fwd_bwd(_lib.ptr(e32), _lib.ptr(d32), B, N, E, _lib.ptr(p32), 1.0,
                                                            _lib.ptr(per), _lib.ptr(de), 0, _lib.stream_ptr()),
                       "convdr_inbatch_ce_fwd_bwd")
        ctx.save_for_backward(de)
        return per.mean()

    @staticmethod
    def backward(ctx, g):
        (de,) = ctx.saved_tensors
        return de * g, None, None


def gather_inbatch_docs(docs, group=None):
    """docs [B, K, E] (this rank's teacher document embeddings, positive first) -> (docs_all [W * B * K, E] in rank
    order, pos int64 [B] = row of each LOCAL query's positive inside docs_all).  One all-gather over the ranks
    (RCCL on GPUs; 512 x 10 x 768 fp32 = 15.7 MB at the configs[4] size); identity on one process."""
    from . import parallel
    B, K, E = docs.shape
    flat = docs.reshape(B * K, E).contiguous()
    allv = parallel.all_gather_rows(flat, group)
    import torch.distributed as dist
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    pos = rank * B * K + torch.arange(B, device=docs.device, dtype=torch.int64) * K
    return allv, pos


def ranking_loss_inbatch(embs, docs_all, pos):
    """In-batch-negative ranking loss (BASELINE configs[4]; oracle/train.py:inbatch_rank_loss): CrossEntropy over the
    scores of every gathered document, target = the query's own positive.  The mean is over the LOCAL queries; with the
    gradient all-reduce averaging over ranks (DataParallelStudent) that is the global mean."""
    return _InBatchCE.apply(embs, docs_all.detach(), pos)


# --------------------------------------------------------------------------------------------
# clip + optimizer + schedule
# --------------------------------------------------------------------------------------------
def _flat_view(tensors):
    """The single flat buffer the tensors tile contiguously, or None."""
    if not tensors:
        return None
    tensors = sorted(tensors, key=lambda t: t.data_ptr())     # parameter order != arena order
    base = tensors[0]
    start = base.data_ptr()
    off = start
    for t in tensors:
        if not t.is_contiguous() or t.dtype != torch.float32 or t.data_ptr() != off or \
                t.untyped_storage().data_ptr() != base.untyped_storage().data_ptr():
            return None
        off += t.numel() * 4
    n = (off - start) // 4
    return torch.as_strided(base, (n,), (1,), base.storage_offset())


_EMB_SUMSQ_MAIN = os.environ.get("CONVDR_EMB_SUMSQ_MAIN", "0") == "1"     # A/B: the embedding slice's sum behind the backward's join (rounds 3-4)


def _overlapped_sumsq(flat, tower, scratch, dev):
    """Sum of squares of a FRESH gradient arena, piece by piece: every encoder layer's slice on a side stream as soon as
    the backward's completion events say it is final (convdr_backward_wait_layer) -- under the backward of the layers
    below --, the embeddings behind the main chain's last kernels (beside layer 0's weight-gradient branch), the head on the
    current stream after the backward.  Returns the number of partial sums written
    to `scratch`, or 0 when the arena is not the one the tower's last backward wrote."""
    info = getattr(tower, "_flat", None)
    if info is None or getattr(tower, "_last_backward_arena", None) != flat.data_ptr():
        return 0
    offs = info.get("offs")
    if offs is None:
        offs = info["offs"] = np.concatenate([[0], np.cumsum([p.numel() for p in info["params"]])]).astype(np.int64)
    nl = len(tower.encoder.layer)
    if int(offs[-1]) != flat.numel() or len(offs) < 6 + 16 * nl:
        return 0
    L = _lib.lib()
    per, nb_emb, nb_head = 64, 256, 16
    if scratch.numel() < nl * per + nb_emb + nb_head:
        return 0
    main = torch.cuda.current_stream(dev)
    side = _norm_stream(dev)
    base, sbase = flat.data_ptr(), scratch.data_ptr()
    b0, e1 = int(offs[5]), int(offs[5 + 16 * nl])
    with torch.cuda.stream(side):
        for l in reversed(range(nl)):
            if l == 0 and not _EMB_SUMSQ_MAIN:
                # the embedding slice (a third of the arena) is written by the main chain's last kernels, BEFORE that chain
                # waits for layer 0's weight-gradient branch: summed here, beside that branch, instead of behind the join
                # (and ahead of layer 0's slice on this stream, whose event is that branch's end)
                _lib.check(L.convdr_backward_wait_layer(-1, side.cuda_stream), "convdr_backward_wait_layer")
                _lib.check(L.convdr_grad_sumsq(C.c_void_p(base), b0, C.c_void_p(sbase + 4 * nl * per), nb_emb, side.cuda_stream),
                           "convdr_grad_sumsq")
            b, e = int(offs[5 + 16 * l]), int(offs[5 + 16 * (l + 1)])
            _lib.check(L.convdr_backward_wait_layer(l, side.cuda_stream), "convdr_backward_wait_layer")
            _lib.check(L.convdr_grad_sumsq(C.c_void_p(base + 4 * b), e - b, C.c_void_p(sbase + 4 * l * per), per, side.cuda_stream),
                       "convdr_grad_sumsq")
    if _EMB_SUMSQ_MAIN:
        _lib.check(L.convdr_grad_sumsq(C.c_void_p(base), b0, C.c_void_p(sbase + 4 * nl * per), nb_emb, main.cuda_stream), "convdr_grad_sumsq")
    _lib.check(L.convdr_grad_sumsq(C.c_void_p(base + 4 * e1), flat.numel() - e1, C.c_void_p(sbase + 4 * (nl * per + nb_emb)), nb_head,
                                   main.cuda_stream), "convdr_grad_sumsq")
    main.wait_stream(side)
    return nl * per + nb_emb + nb_head


def _norm_stream(device):
    return _side_stream(device)      # (idle during the backward: the teacher's forward and the weight packing are long done)


def clip_grad_norm_(parameters, max_norm, defer_to=None, extra_scale=1.0, overlap_backward=False):
    """torch.nn.utils.clip_grad_norm_ (run_convdr_train.py:188-189) in one or a few kernels.  Returns the total
    norm as a device scalar (no host sync).
    defer_to: an ``AdamW`` of this module.  When the gradients form one flat arena the clip coefficient is then only
    computed and handed to the optimizer, whose update kernel multiplies it into the gradient on the fly -- one pass over
    the 125 M gradients less; the stored gradients stay unscaled (train_step zeroes them right after the update).
    extra_scale: the gradients are (sum over ranks) and still have to be multiplied by this factor (1 / world size):
    the norm and the clip coefficient are those of the scaled gradients, the scaling itself rides on the same pass.
    overlap_backward: the gradients come straight from a backward that may still be running on the device and nothing has
    touched them since (no all-reduce): the per-layer slices of the arena are then summed on a side stream behind that
    backward's per-layer completion events, and only embeddings + head (a third of the arena) after it."""
    parameters = list(parameters)
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.zeros(())
    dev = grads[0].device
    L = _lib.lib()
    scratch = torch.empty(1024, dtype=torch.float32, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        flat = _flat_view(grads)
        if flat is not None:
            defer = defer_to is not None and defer_to.can_flat_step(flat)
            count = 0
            if defer and overlap_backward:
                ops = defer_to.__dict__.get("_flat_ops_cache")
                tower = _arena_owner(ops[1]) if ops is not None else None
                if tower is not None:
                    scratch = torch.empty(4096, dtype=torch.float32, device=dev)
                    count = _overlapped_sumsq(flat, tower, scratch, dev)
            if count:
                _lib.check(L.convdr_grad_norm_finish(_lib.ptr(scratch), count, float(max_norm), float(extra_scale), _lib.ptr(out),
                                                     _lib.stream_ptr()), "convdr_grad_norm_finish")
            else:
                _lib.check(L.convdr_grad_norm_clip(_lib.ptr(flat), flat.numel(), float(max_norm), float(extra_scale),
                                                   _lib.ptr(scratch), _lib.ptr(out), 0 if defer else 1, _lib.stream_ptr()),
                           "convdr_grad_norm_clip")
            if defer:
                defer_to._pending_grad_scale = out[1:2]
            return out[0]
        if extra_scale != 1.0:
            sc = torch.full((1,), float(extra_scale), dtype=torch.float32, device=dev)
            for g in grads:
                _lib.check(L.convdr_scale_f32(_lib.ptr(g), g.numel(), _lib.ptr(sc), _lib.stream_ptr()), "convdr_scale_f32")
        norms = torch.empty((len(grads), 2), dtype=torch.float32, device=dev)
        for i, g in enumerate(grads):
            assert g.is_contiguous() and g.dtype == torch.float32
            _lib.check(L.convdr_grad_norm_clip(_lib.ptr(g), g.numel(), float(max_norm), 1.0, _lib.ptr(scratch), _lib.ptr(norms[i]),
                                               0, _lib.stream_ptr()), "convdr_grad_norm_clip")
        total = norms[:, 0].double().pow(2).sum().sqrt().float()
        coef = (max_norm / (total + 1e-6)).clamp(max=1.0).reshape(1).contiguous()
        for g in grads:
            _lib.check(L.convdr_scale_f32(_lib.ptr(g), g.numel(), _lib.ptr(coef), _lib.stream_ptr()), "convdr_scale_f32")
        return total


def _bump_version(t):
    inc = getattr(torch.autograd.graph, "increment_version", None)
    if inc is not None:
        inc(t)
    else:
        with torch.no_grad():
            t.add_(0)


class AdamW(torch.optim.Optimizer):
    """transformers==2.3.0 ``AdamW`` (what utils/dpr_utils.py:87 constructs) with the fused HIP update.

    ``state`` keeps the reference optimizer's layout -- per parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` -- so
    ``state_dict()`` / ``load_state_dict()`` round-trip through the reference's checkpoints (run_convdr_train.py:34).
    When the parameters live in one flat arena (flatten_parameters) the moments are views of two arenas of the same
    layout and the whole model is updated by one launch."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))

    # ---- flat-arena path ------------------------------------------------------------------------
    def _flat_operands(self, G=None):
        """(params, P, G) when one launch can update the whole model: parameters in a flat arena (flatten_parameters),
        gradients in the matching arena written by the backward, all groups sharing their hyper-parameters."""
        g0 = self.param_groups[0]
        keys = ("lr", "betas", "eps", "weight_decay", "correct_bias")
        if any(g[k] != g0[k] for g in self.param_groups for k in keys):
            return None
        ps = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not ps:
            return ps, None, None
        P = _flat_view([p.data for p in ps])
        if G is None:
            G = _flat_view([p.grad for p in ps])
        if P is None or G is None or P.numel() != G.numel():
            return None
        p0, g0p = P.data_ptr(), G.data_ptr()
        if any(p.grad.data_ptr() - g0p != p.data.data_ptr() - p0 for p in ps):     # same order in both arenas
            return None
        return ps, P, G

    def can_flat_step(self, G=None):
        ops = self._flat_operands(G)
        ok = ops is not None and ops[1] is not None
        self.__dict__["_flat_ops_cache"] = ops if ok else None     # reused by the step() that follows a clip
        return ok

    def _adopt_flat_state(self, ps, P):
        """Make ``self.state[p]['exp_avg' / 'exp_avg_sq']`` views of two arenas laid out like P (moving any state that
        load_state_dict or the per-parameter path created into them).  Raises if the arena layout changed under existing
        moments instead of silently restarting them from zero."""
        st = self.__dict__.get("_flat_state")
        if st is not None and (st["m"].numel() != P.numel() or st["m"].device != P.device or st["base"] != P.data_ptr()):
            raise RuntimeError("AdamW: the flat parameter arena changed (size %d -> %d) under existing optimizer moments; "
                               "build a new optimizer after flatten_parameters / resize_token_embeddings"
                               % (st["m"].numel(), P.numel()))
        if st is None:
            st = self.__dict__["_flat_state"] = {"m": torch.zeros_like(P), "v": torch.zeros_like(P), "base": P.data_ptr(),
                                                 "adopted": set()}
        if len(st["adopted"]) == len(ps):
            return st
        base = P.data_ptr()
        for p in ps:
            if id(p) in st["adopted"]:
                continue
            o, n = (p.data.data_ptr() - base) // 4, p.numel()
            mv, vv = st["m"][o:o + n].view(p.shape), st["v"][o:o + n].view(p.shape)
            ps_state = self.state[p]
            if len(ps_state) == 0:
                ps_state["step"] = 0
            else:
                mv.copy_(ps_state["exp_avg"])
                vv.copy_(ps_state["exp_avg_sq"])
            ps_state["exp_avg"], ps_state["exp_avg_sq"] = mv, vv
            st["adopted"].add(id(p))
        return st

    def _try_flat_step(self):
        ops = self.__dict__.pop("_flat_ops_cache", None) or self._flat_operands()
        ok = ops is not None
        if ok and ops[0]:
            ps, P, G = ops
            st = self._adopt_flat_state(ps, P)
            steps = {int(self.state[p]["step"]) for p in ps}
            ok = len(steps) == 1              # parameters at different step counts: per-parameter update
        scale = self.__dict__.pop("_pending_grad_scale", None)
        if not ok:
            if scale is not None:
                # clip_grad_norm_ deferred the clip coefficient (and a folded 1 / world) to this step: it must not be lost
                # on the per-parameter fallback -- apply it to the gradients now
                for g in self.param_groups:
                    for p in g["params"]:
                        if p.grad is not None:
                            with torch.cuda.device(p.grad.device):
                                _lib.check(_lib.lib().convdr_scale_f32(_lib.ptr(p.grad), p.grad.numel(), _lib.ptr(scale),
                                                                       _lib.stream_ptr()), "convdr_scale_f32")
            return False
        if not ops[0]:
            return True
        g0 = self.param_groups[0]
        step = steps.pop() + 1
        b1, b2 = g0["betas"]
        # When P is a tower's whole arena, the update kernel also rewrites the packed bf16 weights the GEMMs read (every
        # element of the copy: all of P is updated), so the next forward finds them current instead of re-casting 0.5 GB
        tower = _arena_owner(P)
        target = tower.fused_update_target(P) if tower is not None else None
        Pb, w0 = target if target is not None else (None, 0)
        with torch.cuda.device(P.device):
            _lib.check(_lib.lib().convdr_adamw_step_packed(_lib.ptr(P), _lib.ptr(G), _lib.ptr(st["m"]), _lib.ptr(st["v"]),
                                                           P.numel(), g0["lr"], b1, b2, g0["eps"], g0["weight_decay"], step,
                                                           int(g0["correct_bias"]), _lib.ptr(scale), _lib.ptr(Pb), w0,
                                                           _lib.stream_ptr()),
                       "convdr_adamw_step_packed")
        for p in ps:
            self.state[p]["step"] = step
            _bump_version(p)
        if target is not None:
            tower.adopt_fused_update()
        return True

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        st = self.__dict__.get("_flat_state")
        if st is not None:
            st["adopted"] = set()             # the loaded moments are fresh tensors: move them into the arenas on the next step

    @torch.no_grad()
    def step(self, closure=None):
        L = _lib.lib()
        if self._try_flat_step():
            return
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p.data, dtype=torch.float32)
                    st["exp_avg_sq"] = torch.zeros_like(p.data, dtype=torch.float32)
                st["step"] = int(st["step"]) + 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                assert p.data.is_contiguous() and p.dtype == torch.float32 and g.dtype == torch.float32
                assert st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()
                with torch.cuda.device(p.device):
                    _lib.check(L.convdr_adamw_step(_lib.ptr(p.data), _lib.ptr(g), _lib.ptr(st["exp_avg"]),
                                                   _lib.ptr(st["exp_avg_sq"]), p.numel(), group["lr"], b1, b2, group["eps"],
                                                   group["weight_decay"], st["step"], int(group["correct_bias"]), None,
                                                   _lib.stream_ptr()), "convdr_adamw_step")
                _bump_version(p)     # the kernel wrote through the raw pointer: tell torch (packed-weight cache key)


def get_optimizer(args, model, weight_decay=0.0):
    """utils/dpr_utils.py:80-87: two groups split on 'bias' / 'LayerNorm.weight' in the parameter NAME
    (so the head's ``norm.weight`` lands in the decayed group, exactly as in the reference)."""
    no_decay = ["bias", "LayerNorm.weight"]
    groups = [
        {"params": [p for n, p in model.named_parameters() if not any(nd in n for nd in no_decay)],
         "weight_decay": weight_decay},
        {"params": [p for n, p in model.named_parameters() if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
    ]
    # a training entry point: pick the step's auxiliary streams here, not lazily inside the first step (train.reserve_streams)
    for p in model.parameters():
        if p.is_cuda:
            reserve_streams(p.device)
        break
    return AdamW(groups, lr=args.learning_rate, eps=args.adam_epsilon)


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    """transformers.get_linear_schedule_with_warmup (run_convdr_train.py:71-74)."""
    def lr_lambda(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda, last_epoch)


_SIDE_STREAMS = {}
_log = logging.getLogger("convdr_amd.train")


def _fork_join_pattern(main, side, big, small):
    """A miniature of the backward's stream pattern: four times { the main stream forks `side` off with an event, `side` runs
    one long kernel, the main stream five short dependent ones }, then a join."""
    for blk in range(4):
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            big.mul_(1.0)
        for _ in range(5):
            small.mul_(1.0)
    fin = torch.cuda.Event()
    fin.record(side)
    main.wait_event(fin)


def _fork_join_scores(main, cands, big, small, reps=3):
    """Wall time (us) of _fork_join_pattern per candidate stream: streams that HIP has put on hardware queues that serialise
    against the main stream's take ~16 % longer (1.15 vs 0.98 ms) -- and this, unlike a plain "does a short kernel overtake a
    long one" test, separates exactly the stream choices with which the real step loses its overlap
    (tools/dbg/stream_proxy_probe.py).  Round 5: the part is warmed up first and the repetitions are interleaved over the
    candidates (minimum per candidate) -- a candidate timed while the clocks were still ramping up used to be able to lose
    against a serialising one timed later."""
    import time
    for c in cands[:2]:
        for _ in range(3):
            _fork_join_pattern(main, c, big, small)
    best = [float("inf")] * len(cands)
    for rep in range(reps):
        for i, c in enumerate(cands):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _fork_join_pattern(main, c, big, small)
            torch.cuda.synchronize()
            best[i] = min(best[i], (time.perf_counter() - t0) * 1e6)
    return best


class _StreamSets:
    """The auxiliary streams of one device's training steps, and the watchdog that re-picks them.

    sets[k] = (A, B, C): the stream of the frozen teacher's forward / weight packing / per-layer gradient norms, the stream
    of the backward's weight-gradient branches, the gradient all-reduce stream of parallel.py.  HIP multiplexes all streams
    of a process onto GPU_MAX_HW_QUEUES = 4 hardware queues in order of first use, and a stream that shares the main
    stream's queue serialises against it: the configs[2] step measured 10.3 .. 12.4 ms depending on nothing but how many
    streams the process had used before (tools/dbg/stream_queue_probe.py; extra priorities, CU masks or more queues are far
    worse: 17-31 ms).  So the first use times a miniature of the backward's fork / join pattern on eight streams of torch's
    pool (~40 ms, once) and forms two disjoint sets from the six fastest.

    Round 5 -- the choice checks itself on the REAL step (round 4 measured 1 process in 30-50 whose calibrated pair still
    serialised).  train_step stamps an event at its start; the period between two stamps, divided by the step's token count,
    is the step's cost.  After `PROBE` steps on set 0 the next `PROBE` run on set 1; whichever has the lower median cost
    stays (set 1 must win by more than `MARGIN` to replace set 0; 2.5 %: the probe's own spread is ~1.5 %, a set in the
    shared-queue mode costs 3-4 %).  From then on, three consecutive steps more than `DRIFT`
    above the best median this process has seen move the step to the other set.  Every decision is logged (logger
    "convdr_amd.train") and kept in `.decisions`.  Switching is safe between steps: every backward ends with the main stream
    waiting for its side streams.  CONVDR_STREAM_SELFCHECK=0 turns the watchdog off."""
    PROBE, MARGIN, DRIFT, REBASE = 4, 0.025, 0.08, 24

    def __init__(self, device):
        self.device = device
        self.sets, self.scores, self.decisions = [], None, collections.deque(maxlen=64)
        self.active = 0
        self.enabled = os.environ.get("CONVDR_STREAM_SELFCHECK", "1") != "0"
        self.pending = collections.deque()      # (event at the start of a step, tokens of that step, set index)
        self.samples = {}                       # set index -> costs (ms per k-token) since the last switch
        self.median = {}                        # set index -> best median seen
        self.phase = "probe0"
        self.skip = 2                           # steps whose period is not used (the first ones, and those around a switch)
        self.high = 0
        self.moves = collections.deque(maxlen=4)     # step indices of the watchdog's last drift moves (re-baselining, below)
        self.nsteps = 0
        self._calibrate()
        self._apply()

    def _calibrate(self):
        device = self.device
        main = torch.cuda.current_stream(device)
        cands = [torch.cuda.Stream(device=device) for _ in range(8)]
        order = list(range(len(cands)))
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            _log.warning("convdr_amd.train: stream calibration skipped (a graph capture is in progress): uncalibrated streams; "
                         "call train.reserve_streams() before capturing")
        else:
            try:
                with torch.cuda.device(device):
                    big = torch.zeros(128 << 20, dtype=torch.float32, device=device)       # one pass: ~250 us
                    small = torch.zeros(16 << 20, dtype=torch.float32, device=device)      # one pass: ~30 us
                    t = _fork_join_scores(main, cands, big, small)
                    order = sorted(order, key=lambda i: t[i])
                    self.scores = [(round(t[i]), i) for i in order]                          # (us, candidate): diagnostics
                    del big, small
            except RuntimeError as e:            # (out of memory: keep the uncalibrated order, say so)
                _log.warning("convdr_amd.train: stream calibration failed (%s): uncalibrated streams", e)
        s = [cands[i] for i in order]
        if self.scores:
            # LOUD when the runtime gives the step nothing to work with (VERDICT r05 "weak" 10): fewer than two streams that run
            # beside the main one means the weight-gradient branch and / or the teacher's forward will serialise with the main
            # chain (3-20 % per step, r04) whatever this class picks -- e.g. a GPU_MAX_HW_QUEUES setting or a runtime whose
            # stream -> queue mapping differs from the one this calibration was built against
            n_good = sum(1 for us, _ in self.scores if us <= 1.08 * self.scores[0][0])
            self.concurrent_streams = n_good
            if n_good < 2:
                _log.warning("convdr_amd.train: only %d of %d candidate streams run beside the main stream (fork / join pattern, us: %s): "
                             "the training step's side branches will serialise with its main chain; check GPU_MAX_HW_QUEUES and "
                             "the number of streams this process created before training", n_good, len(self.scores),
                             [us for us, _ in self.scores])
        self.sets = [tuple(s[0:3]), tuple(s[3:6])]
        self.clusters = None
        by_queue = self._sets_by_queue(cands, order, main) if (self.scores and not capturing) else None
        if by_queue is not None:
            self.sets = by_queue
        elif self.scores and self.scores[5][0] > 1.08 * self.scores[0][0]:
            # fewer than six streams run beside the main one: the second set re-uses the good ones in other roles
            good = [cands[i] for us, i in self.scores if us <= 1.08 * self.scores[0][0]]
            if len(good) >= 2:
                self.sets[1] = tuple((good[::-1] * 3)[:3])
            else:
                self.sets[1] = self.sets[0]
        self._keep = cands

    def _sets_by_queue(self, cands, order, main):
        """Round 5, late: the roles of a set must not share a hardware queue with EACH OTHER either.  The step's +3 % mode
        (19 % of fresh processes, profiles/r05_stream_queue_clusters.txt) was exactly the sets whose stream A (teacher / norms)
        and stream B (weight-gradient branches) sat on one queue: A's per-layer norm kernels wait for events that B records, and
        a waiting packet holds up whatever that queue carries behind it.  Streams are grouped by queue with a pairwise test --
        one spinning thread on each of two streams (torch.cuda._sleep: no resource contention): together they take one spin
        time on two queues, two on one -- and every set takes one stream from each of three groups.  None when the test is
        unavailable or finds fewer than two groups (the caller keeps the order-of-calibration sets)."""
        import time
        sleep = getattr(torch.cuda, "_sleep", None)
        if sleep is None or not self.scores:
            return None
        best, us_of = self.scores[0][0], {c: us for us, c in self.scores}
        good = [i for i in order if us_of[i] <= 1.08 * best]      # (the streams that run beside the main one)
        if len(good) < 2:
            return None
        spin = 200_000            # ~0.1 ms

        def pair_us(x, y):
            t = float("inf")
            for _ in range(3):
                torch.cuda.synchronize()
                ev = torch.cuda.Event()
                ev.record(main)
                t0 = time.perf_counter()
                for st in (x, y):
                    with torch.cuda.stream(st):
                        st.wait_event(ev)
                        sleep(spin)
                torch.cuda.synchronize()
                t = min(t, time.perf_counter() - t0)
            return t
        try:
            with torch.cuda.device(self.device):
                one = pair_us(cands[good[0]], cands[good[0]]) / 2.0
                groups, left = [], list(good)
                while left:
                    head, rest = left[0], left[1:]
                    same = [j for j in rest if pair_us(cands[head], cands[j]) > 1.6 * one]
                    groups.append([head] + same)
                    left = [j for j in rest if j not in same]
        except RuntimeError:
            return None
        self.clusters = groups
        if len(groups) < 2:
            return None
        groups = sorted(groups, key=len, reverse=True)[:3]
        sets = []
        for k in range(2):
            pick = [cands[g[k % len(g)]] for g in groups]
            if len(pick) == 2:                     # two queues beside the main one: A and B apart, C rides with A
                pick.append(pick[0])
            sets.append(tuple(pick))
        return sets

    def _apply(self):
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().convdr_train_set_side_stream(C.c_void_p(self.sets[self.active][1].cuda_stream)),
                       "convdr_train_set_side_stream")

    def current(self):
        return self.sets[self.active]

    def _decide(self, msg, new_active=None):
        if new_active is not None and new_active != self.active:
            self.active = new_active
            self._apply()
            self.skip = 2
        self.samples = {}
        self.high = 0
        self.decisions.append(msg)
        _log.info("convdr_amd.train: %s", msg)

    def step_begin(self, tokens):
        """Called at the top of every train_step, on the main stream."""
        if not self.enabled or self.sets[0] is self.sets[1] or torch.cuda.is_current_stream_capturing():
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.device))
        self.pending.append((ev, max(1.0, float(tokens)), self.active))
        while len(self.pending) >= 2 and self.pending[1][0].query():
            e0, w0, k0 = self.pending.popleft()
            if self.skip > 0 or k0 != self.pending[0][2] or k0 != self.active:
                self.skip = max(0, self.skip - 1)
                continue
            self._feed(k0, e0.elapsed_time(self.pending[0][0]) / w0 * 1e3)
        if len(self.pending) > 64:              # (a host that never lets the GPU catch up: keep the queue bounded)
            self.pending.popleft()

    def _feed(self, k, cost):
        xs = self.samples.setdefault(k, [])
        xs.append(cost)
        self.nsteps += 1
        if len(xs) > 64:
            del xs[:-16]
        med = float(np.median(xs[-self.PROBE:]))
        if self.phase == "probe0":
            if len(xs) >= self.PROBE:
                self.median[0] = med
                self.phase = "probe1"
                self._decide("stream self-check: set 0 costs %.4f ms per k-token over %d steps; trying set 1" % (med, self.PROBE), 1)
        elif self.phase == "probe1":
            if len(xs) >= self.PROBE:
                self.median[1] = med
                self.phase = "steady"
                if med < (1.0 - self.MARGIN) * self.median[0]:
                    self._decide("stream self-check: set 1 is %.1f %% cheaper than set 0 (%.4f vs %.4f ms per k-token): keeping set 1"
                                 % (100 * (1 - med / self.median[0]), med, self.median[0]), 1)
                else:
                    self._decide("stream self-check: set 0 stays (%.4f vs %.4f ms per k-token on set 1)" % (self.median[0], med), 0)
        else:
            if len(xs) >= self.PROBE and med < self.median.get(k, float("inf")):
                self.median[k] = med
            if not self.median:                  # (just re-baselined: no judgement until this set has PROBE fresh samples)
                return
            best = min(self.median.values())
            self.high = self.high + 1 if cost > (1.0 + self.DRIFT) * best else 0
            if self.high >= 3:
                # Two drift moves within REBASE steps mean BOTH sets have just been seen "high": it is the workload that got
                # dearer per token (longer sequences: attention is not linear in length; ranking documents), not a stream
                # set gone bad -- `best` never rises by itself, so without this the watchdog would hop sets every ~5 steps
                # for good.  Re-baseline: forget the medians, stay, probe this set afresh.
                if self.moves and self.nsteps - self.moves[-1] <= self.REBASE:
                    self.moves.clear()
                    self.median = {}
                    self._decide("stream self-check: both sets high within %d steps (%.4f vs %.4f ms per k-token): the workload "
                                 "changed, not the streams -- re-baselining on set %d" % (self.REBASE, cost, best, k))
                    return
                self.moves.append(self.nsteps)
                other = 1 - k
                self.median.pop(other, None)     # (re-measured after the move; if it is no better the next drift moves back)
                self._decide("stream self-check: three steps in a row %.0f %% above this process's best (%.4f vs %.4f ms per k-token):"
                             " moving to set %d" % (100 * (cost / best - 1), cost, best, other), other)
                self.median[other] = float("inf")


def _stream_sets(device):
    key = (device.type, device.index)
    ss = _SIDE_STREAMS.get(key)
    if ss is None:
        ss = _SIDE_STREAMS[key] = _StreamSets(device)
    return ss


def _aux_streams(device):
    """(A, B, C) of the device's active stream set (see _StreamSets)."""
    return _stream_sets(device).current()


def _watch_step_begin(device, sig, doc_tokens=0.0):
    concat_lens, target_lens, cshape, tshape = sig
    # a step's work in tokens: the student's rows count three times (forward + backward), the teacher's once -- its target
    # rows and, in a ranking step that re-encodes them, its document rows (doc_tokens)
    ws = float(np.sum(concat_lens)) if concat_lens is not None else float(cshape[0] * cshape[1])
    wt = float(np.sum(target_lens)) if target_lens is not None else float(tshape[0] * tshape[1])
    _stream_sets(device).step_begin(3.0 * ws + wt + float(doc_tokens))


def reserve_streams(device=None):
    """Pick the training step's auxiliary streams NOW (~40 ms: six synchronisations per candidate, 0.6 GB of temporaries).
    Called by get_optimizer and DataParallelStudent, so that the calibration never lands inside a step (or inside a graph
    capture, where it is skipped with a warning); bench.py calls it before anything else: HIP's stream -> hardware-queue
    assignment follows the order of first use."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    return _aux_streams(dev)


def stream_decisions(device=None):
    """The stream watchdog's log for `device` (diagnostics): calibration scores and every decision taken so far."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    ss = _stream_sets(dev)
    return {"scores_us": ss.scores, "queue_groups": getattr(ss, "clusters", None), "active_set": ss.active,
            "concurrent_streams": getattr(ss, "concurrent_streams", None),
            "decisions": list(ss.decisions), "medians": dict(ss.median),
            "phase": ss.phase if ss.enabled and ss.sets[0] is not ss.sets[1] else "off"}


def settle_streams(step_fn, device=None, max_steps=32):
    """Run `step_fn(i)` (one training step) until the stream watchdog has probed both of its stream sets on the real step
    and settled (~14 steps); returns the number of steps run.  For benchmarks that want their timed region free of the
    probe; a training run simply settles during its first steps."""
    n = 0
    while n < max_steps and stream_decisions(device)["phase"] in ("probe0", "probe1"):
        step_fn(n)
        n += 1
    return n


def _side_stream(device):
    return _aux_streams(device)[0]


def _set_mode(module, training):
    """module.train(training) only when some module of the tree disagrees.  The reference re-flags the whole tree every step
    (run_convdr_train.py:107,112: model.train() / teacher_model.eval()); nn.Module.train() writes ~190 attributes through
    __setattr__ (0.7 ms of host time per call), reading them is ~30 us -- and, unlike a test of the root flag alone, repairs a
    submodule that was flipped by itself (model.roberta.eval() in an evaluation helper)."""
    for m in module.modules():
        if m.training != training:
            module.train(training)
            return


class TeacherEmbeddingCache:
    """Embeddings of the FROZEN, eval-mode teacher on the samples' targets, keyed by sample id (SURVEY.md section 8f-2's idea
    applied to run_convdr_train.py:110-112).  The teacher has no dropout in eval mode and never changes, so
    teacher_model(target_ids, target_id_mask) of a sample is the same tensor every epoch: from the second epoch on (or after a
    precompute pass) `train_step(..., teacher_embs=cache.lookup(ids))` replaces the teacher's forward -- 0.45-0.6 ms of a
    9.8 ms configs[2] step -- with a row gather.  The rows ARE earlier outputs of the same forward; they are bit-identical
    to what the step's own teacher forward would produce only when that earlier forward saw the same packed-row count (the
    launcher picks whole-contraction tiles, the split-K FFN2 or the fused LayerNorm path by row count, and the fp32 summation
    order differs between them: 1 - cos ~ 1e-5, far inside the 1e-3 bar).  `fill(chunk=B)` with the training batch size (or
    `put` from the first epoch's own steps) reproduces the training step's kernel path.  The reference flow (teacher run
    every step) stays the default of train_step.
    Storage: one fp32 [capacity, E] device tensor + a host dict id -> row."""

    def __init__(self, capacity, dim=768, device="cuda"):
        self.store = torch.empty((int(capacity), int(dim)), dtype=torch.float32, device=device)
        self.row_of = {}

    def __len__(self):
        return len(self.row_of)

    def has_all(self, sample_ids):
        r = self.row_of
        return all(int(i) in r for i in sample_ids)

    def put(self, sample_ids, embs):
        rows = []
        for i in sample_ids:
            i = int(i)
            if i not in self.row_of:
                if len(self.row_of) >= self.store.shape[0]:
                    raise ValueError("TeacherEmbeddingCache: capacity %d exhausted" % self.store.shape[0])
                self.row_of[i] = len(self.row_of)
            rows.append(self.row_of[i])
        idx = torch.as_tensor(rows, dtype=torch.int64).to(self.store.device, non_blocking=True)
        self.store.index_copy_(0, idx, embs.detach().to(self.store.dtype))

    def lookup(self, sample_ids):
        idx = torch.as_tensor([self.row_of[int(i)] for i in sample_ids], dtype=torch.int64).to(self.store.device, non_blocking=True)
        return self.store.index_select(0, idx)

    @torch.no_grad()
    def fill(self, teacher_model, sample_ids, target_ids, target_id_mask, chunk=256, **kw):
        """Precompute pass: run the teacher over (target_ids, target_id_mask) in chunks and store the rows."""
        _set_mode(teacher_model, False)
        for i in range(0, len(sample_ids), chunk):
            self.put(sample_ids[i:i + chunk], teacher_model(target_ids[i:i + chunk], target_id_mask[i:i + chunk], **kw))


def train_step(args, model, teacher_model, optimizer, scheduler, batch, doc_ids=None, doc_mask=None, ddp=None, doc_embs=None,
               force_overlap=False, step=None, teacher_embs=None, loss_weight=1.0):
    """One iteration of the reference loop body (run_convdr_train.py:101-193) with pre-tokenised ranking documents
    (`doc_ids` / `doc_mask` int64 [B * (num_negatives + 1), Ld], positive first within each group).
    batch: (concat_ids, concat_id_mask, target_ids, target_id_mask) as in the reference, optionally followed by the two
    HOST length arrays (concat_lens, target_lens) -- the collate function knows them for free, and with them the step has
    no device -> host round trip, so the host enqueues a whole step ahead of the GPU.
    step: index of this micro-batch in the epoch (the reference's ``step``); clip / optimizer / scheduler / zero_grad run
    only when (step + 1) % gradient_accumulation_steps == 0 (run_convdr_train.py:172-193).  Required when accumulating.
    teacher_embs: optional [B, E] embeddings of the frozen teacher on this batch's targets (TeacherEmbeddingCache.lookup);
    with it the teacher's forward is skipped.  None (default): the reference flow, the teacher runs every step (:110-112).
    loss_weight: factor on this rank's loss before the backward -- parallel.shard_batch(..., return_weight=True) for a global
    batch that does not divide over the ranks (nn.DataParallel's short last chunk); the returned losses stay unweighted.
    Returns (loss, loss1, loss2) as device scalars."""
    concat_ids, concat_id_mask, target_ids, target_id_mask = batch[:4]
    concat_lens, target_lens = (batch[4], batch[5]) if len(batch) >= 6 else (None, None)
    gas = int(getattr(args, "gradient_accumulation_steps", 1) or 1)
    if gas > 1 and step is None:
        raise ValueError("train_step: gradient_accumulation_steps = %d needs the micro-batch index `step`" % gas)
    do_step = gas == 1 or (step + 1) % gas == 0
    _set_mode(model, True)
    if teacher_model is not None:
        _set_mode(teacher_model, False)
    doc_tokens = float(doc_ids.numel()) if (getattr(args, "ranking_task", False) and doc_embs is None and doc_ids is not None) else 0.0
    _watch_step_begin(concat_ids.device, (concat_lens, target_lens, concat_ids.shape, target_ids.shape), doc_tokens)
    # The frozen teacher's forward is independent of the student's and, at 64 x 64 tokens, fills barely a third of the
    # CUs: it runs on a side stream under the student's forward and is joined before the loss needs it.
    main = torch.cuda.current_stream()
    kw_t = {} if target_lens is None else {"seq_lens": target_lens}
    kw_s = {} if concat_lens is None else {"seq_lens": concat_lens}
    if teacher_embs is None:
        side = _side_stream(concat_ids.device)
        side.wait_stream(main)
        with torch.cuda.stream(side), torch.no_grad():
            teacher_embs = teacher_model(target_ids, target_id_mask, **kw_t).detach()
        embs = model(concat_ids, concat_id_mask, **kw_s)
        main.wait_stream(side)
        teacher_embs.record_stream(main)
    else:
        teacher_embs = teacher_embs.detach()
        embs = model(concat_ids, concat_id_mask, **kw_s)
    # KD term alone, no accumulation / weighting: `loss.backward()` would seed _MSE.backward with a ones scalar and multiply
    # the stored gradient by it -- a fill and an elementwise launch on the chain between forward and backward -- so the
    # encoder's backward is seeded with the gradient itself (bit-identical: the factor is exactly 1.0)
    direct = (not getattr(args, "no_mse", False) and not getattr(args, "ranking_task", False) and gas == 1 and loss_weight == 1.0
              and embs.requires_grad and os.environ.get("CONVDR_KD_DIRECT_BACKWARD", "1") != "0")
    if direct:
        loss1, seed = _mse_value_and_grad(embs, teacher_embs)
        embs.backward(seed)
        loss_out, loss2 = loss1, None
        return _finish_step(args, model, optimizer, scheduler, ddp, force_overlap, do_step, gas, loss_out, loss1, loss2, concat_ids)
    loss1 = None if getattr(args, "no_mse", False) else mse_loss(embs, teacher_embs)
    loss, loss2 = loss1, None
    if getattr(args, "ranking_task", False):
        bs = concat_ids.shape[0]
        if doc_embs is not None:
            # the frozen teacher's document embeddings already exist in the corpus blocks whenever the
            # ranking file carries doc ids; looking them up replaces 10 x 512-token encodes per sample per step
            docs = doc_embs.view(bs, args.num_negatives + 1, -1)
        else:
            # The reference encodes the B x (K + 1) documents 8 at a time (doc_batch_size = 8, :139 -- a memory
            # measure for 16-32 GB GPUs).  Embeddings do not depend on the batching, and 8 x 512 tokens leave the chip
            # two thirds idle (640 documents: 164 ms in eights, 68 ms in chunks of 64, tools/dbg/doc_enc.py).
            chunk = int(getattr(args, "doc_batch_size", 128))
            outs = []
            with torch.no_grad():
                for i in range(0, doc_ids.shape[0], chunk):
                    outs.append(teacher_model(doc_ids[i:i + chunk], doc_mask[i:i + chunk], is_query=False).detach())
            docs = torch.cat(outs, 0).view(bs, args.num_negatives + 1, -1)
        if getattr(args, "in_batch_negatives", False):
            # configs[4] variant, off by default (the reference scores a query against its own K + 1 documents only)
            docs_all, pos = gather_inbatch_docs(docs, getattr(ddp, "group", None))
            loss2 = ranking_loss_inbatch(embs, docs_all, pos)
        else:
            loss2 = ranking_loss(embs, docs)
        loss = loss1 + loss2 if loss1 is not None else loss2
    loss_out = loss
    if gas > 1:
        loss = loss / gas
    if loss_weight != 1.0:
        loss = loss * float(loss_weight)
    loss.backward()
    return _finish_step(args, model, optimizer, scheduler, ddp, force_overlap, do_step, gas, loss_out, loss1, loss2, concat_ids)


def _finish_step(args, model, optimizer, scheduler, ddp, force_overlap, do_step, gas, loss_out, loss1, loss2, token_ids=None):
    """train_step after the backward: all-reduce, clip, optimizer, scheduler, zero_grad (run_convdr_train.py:172-193)."""
    if do_step:
        scale = 1.0
        if ddp is not None:
            # per-layer collectives under the backward when the gradients are fresh (parallel.py); SUM over ranks --
            # the 1 / world factor rides on the clip / AdamW pass instead of costing its own pass over 0.5 GB
            # (token_ids: for DataParallelStudent(sparse_embedding=True) -- the ids this rank embedded, padding included: a
            #  superset of the word-table rows its gradient touched; only valid without accumulation over micro-batches)
            kw = {"token_ids": token_ids} if (getattr(ddp, "sparse_embedding", False) and gas == 1) else {}
            scale = ddp.allreduce_grads(force_overlap=force_overlap, average=False, **kw)
        clip_grad_norm_(list(model.parameters()), args.max_grad_norm, defer_to=optimizer if isinstance(optimizer, AdamW) else None,
                        extra_scale=scale, overlap_backward=ddp is None and gas == 1)
        optimizer.step()
        scheduler.step()
        model.zero_grad()
    return (loss_out / gas if gas > 1 else loss_out).detach(), loss1, loss2
