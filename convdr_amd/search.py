"""Brute-force inner-product search over passage-embedding blocks on MI355X.

Mirrors, name for name, what the reference does with FAISS:
  * ``FlatIPIndex``         <-> ``faiss.IndexFlatIP(768)`` (+ ``index_cpu_to_gpu_multiple``)
                               /root/reference/drivers/run_convdr_inference.py:353-368,
                               ``.add`` :180, ``.search`` :182, ``.reset`` :202
  * ``search_one_by_one``   <-> run_convdr_inference.py:157-242 (same merge rule, same output shapes)
  * ``EvalDevQuery``        <-> run_convdr_inference.py:21-113  (same .trec / .jsonl text)
All scoring / selection runs in libconvdr_hip.so (csrc/ip_topk.hip).
"""
import json
import os
import pickle
import warnings

import numpy as np

from . import _lib

def _torch():
    import torch
    return torch


STATUS_OK, STATUS_OVERFLOW, STATUS_TOO_FEW, STATUS_UNCERTAIN, STATUS_RANGE = 0, 1, 2, 3, 4

_KINDS = {"auto": "f16", "fp16": "f16", "fp16x3": "f16", "bf16": "bf16", "bf16x3": "bf16"}


class FlatIPIndex:
    """Exact inner-product index resident in HBM.  ``add`` keeps the fp32 block and builds its 16-bit scan copy;
    ``search`` returns FAISS-shaped ``(D float32 [nq,k], I int64 [nq,k])`` and is certified exact
    (see include/convdr_hip.h).

    precision: "auto" (default) scans in fp16 (v_mfma_f32_32x32x16_f16: the bf16 rate, 8x tighter error band) and
    re-runs the queries that cannot be certified with the split-fp16 scan (three MFMA passes, another 4x); "fp16" /
    "fp16x3" / "bf16" / "bf16x3" pin one rung (the bf16 pair is the round-1/2 ladder: same engine, u = 2^-8).  Whatever
    the rung, a query it cannot certify ends on the exhaustive rung, so the result never depends on the choice.
    center: subtract the column mean of the first added block from every passage before rounding (ranking-neutral,
    shrinks the error band by |p| / |p - mean|; essential for real encoder outputs, which share a large common
    component).
    reserve(n): allocate the resident block for n passages up front; add() then fills it in place.  Without it every
    add() after the first re-allocates (FAISS semantics need one contiguous block): fine for the reference's one add per
    reset, not for building a 117 GB corpus from slices."""

    def __init__(self, d, device=None, cap=4096, rank_target=0, precision="auto", center=True, prepin=True):
        import torch
        if not torch.cuda.is_available():
            raise _lib.ConvdrError("FlatIPIndex needs a GPU (no CPU fallback)")
        assert precision in _KINDS, precision
        _lib.lib()
        # the scan contracts in 64-wide K steps: other widths get zero columns, which add exact zeros to every score
        self.d_in = int(d)
        self.d = (self.d_in + 63) // 64 * 64
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.cap, self.rank_target = int(cap), int(rank_target)
        self.precision, self.center = precision, bool(center)
        self.kind = _KINDS[precision]
        self._half_dtype = torch.float16 if self.kind == "f16" else torch.bfloat16
        self.host_chunk_bytes = 64 << 20       # staging-buffer size of the streamed host -> HBM path (add); 64 MB x 16
                                               # threads measured best on the bench host (tools/dbg/block_load_sweep.py)
        self.host_copy_threads = 16            # file reads / memcpy slices in flight while filling a staging buffer
        self.host_stage_buffers = 4            # pinned staging buffers (3 chunks being read while one crosses PCIe; 37-41 GB/s of a
                                               # 44 GB/s pinned-H2D ceiling on the r03 box, tools/dbg/block_load_sweep.py)
        self.stats = {}
        self._s32 = self._s16 = self._slo = None
        self.reset()
        if prepin:
            self.prepin_staging()

    def prepin_staging(self, wait=False):
        """Allocate and page-lock the process-wide staging buffers of the block loader (4 x 64 MB on the GPU's NUMA node, ~60 ms)
        NOW, on a background thread: the first add() of a block file used to pay for it (19 GB/s against 49 GB/s for every
        later block; the reference's flow constructs the index, then loads 8 block files: run_convdr_inference.py:353,164-180).
        Called by the constructor (prepin=True); add() joins the thread if it is still running."""
        import threading
        key = _staging_key(self.device, self.d, max(2, int(self.host_stage_buffers)))
        rows = max(1, int(self.host_chunk_bytes) // (4 * self.d))
        with _STAGING_LOCK:
            ent = _STAGING.get(key)
            th = _STAGING_THREADS.get(key)
            if (ent is not None and ent[0][0].shape[0] >= rows) or (th is not None and th.is_alive()):
                pass
            else:
                th = threading.Thread(target=_staging, args=(self.device, rows, self.d, key[2]), daemon=True, name="convdr-prepin")
                _STAGING_THREADS[key] = th
                th.start()
        if wait and th is not None:
            th.join()

    def twin(self):
        """An empty index with this one's parameters (search_one_by_one keeps two blocks in flight: one being searched, one
        being loaded)."""
        t = FlatIPIndex(self.d_in, device=self.device, cap=self.cap, rank_target=self.rank_target, precision=self.precision,
                        center=self.center, prepin=False)
        t.host_chunk_bytes, t.host_copy_threads, t.host_stage_buffers = self.host_chunk_bytes, self.host_copy_threads, self.host_stage_buffers
        return t

    # -- faiss-like surface ---------------------------------------------------
    @property
    def ntotal(self):
        return self._n

    # the resident block (views of the used rows of the backing storage)
    @property
    def _p32(self):
        return None if self._s32 is None else self._s32[:self._n]

    @property
    def _pbf(self):
        return None if self._s16 is None else self._s16[:self._n]

    @property
    def _plo(self):
        return None if self._slo is None else self._slo[:self._n]

    def reset(self):
        """faiss ``index.reset()``: forget the passages.  A reservation made with reserve() survives (its memory is reused
        by the next block); ``release()`` drops it."""
        import torch
        self._n = 0
        if not getattr(self, "_reserved", False):
            self._s32 = self._s16 = self._slo = None
        self._centre = None
        self._scale = 1.0                       # power of two applied to the fp16 scan copy (1 for bf16)
        self._max_norm = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._ws = None
        self._x3_first = False
        if not hasattr(self, "_copy_stream"):
            self._copy_stream, self._stage = None, None

    def release(self):
        self._reserved = False
        self.reset()

    def reserve(self, n, lo=None):
        """Backing storage for n passages (fp32 block + 16-bit scan copy, + the remainder copy of the split scan when
        `lo`, default: only if the precision pins it).  Existing rows are kept."""
        import torch
        n = int(n)
        want_lo = (self.precision in ("bf16x3", "fp16x3")) if lo is None else bool(lo)
        if self._s32 is not None and self._s32.shape[0] >= n and (self._slo is not None or not want_lo):
            self._reserved = True
            return
        with torch.cuda.device(self.device):
            def grow(old, dtype):
                t = torch.empty((n, self.d), dtype=dtype, device=self.device)
                if old is not None and self._n:
                    t[:self._n].copy_(old[:self._n])
                return t
            self._s32 = grow(self._s32, torch.float32)
            self._s16 = grow(self._s16, self._half_dtype)
            if want_lo or self._slo is not None:
                self._slo = grow(self._slo, self._half_dtype)
        self._reserved = True

    def _prepare_into(self, src32, dst16, dstlo):
        """scan copy (and remainder copy) of the fp32 rows `src32`; folds their norms into the block's max norm"""
        L = _lib.lib()
        if self.kind == "f16":
            _lib.check(L.convdr_ip_prepare_block_f16(_lib.ptr(src32), src32.shape[0], self.d, _lib.ptr(self._centre),
                                                     float(self._scale), _lib.ptr(dst16), _lib.ptr(dstlo),
                                                     _lib.ptr(self._max_norm), _lib.stream_ptr()), "convdr_ip_prepare_block_f16")
        else:
            _lib.check(L.convdr_ip_prepare_block(_lib.ptr(src32), src32.shape[0], self.d, _lib.ptr(self._centre), _lib.ptr(dst16),
                                                 _lib.ptr(dstlo), _lib.ptr(self._max_norm), _lib.stream_ptr()),
                       "convdr_ip_prepare_block")

    def _first_rows(self, t):
        """The first rows an empty index sees fix its centring vector and, for the fp16 scan, the power-of-two scale of the
        scan copy (from these rows' largest centred norm: one extra pass over them and one host read per index
        lifetime; later rows may be up to 7x longer before the copy has to be rebuilt, see _rebuild_scaled)."""
        if self.center:
            self._set_centre(t)
        if self.kind == "f16":
            L = _lib.lib()
            _lib.check(L.convdr_ip_prepare_block_f16(_lib.ptr(t), t.shape[0], self.d, _lib.ptr(self._centre), 1.0, None, None,
                                                     _lib.ptr(self._max_norm), _lib.stream_ptr()), "convdr_ip_prepare_block_f16")
            self._scale = float(L.convdr_ip_f16_scale(float(self._max_norm.item())))

    def _rebuild_scaled(self):
        """CONVDR_IP_RANGE: rows added after the scale was fixed are more than 7x longer than the first block's longest.
        Re-derive the scale from the block's max norm and rebuild the fp16 copies from the resident fp32 rows."""
        import torch
        with torch.cuda.device(self.device):
            self._scale = float(_lib.lib().convdr_ip_f16_scale(float(self._max_norm.item())))
            if self._n:
                self._prepare_into(self._p32, self._pbf, self._plo)

    def _grow_for(self, m, want_lo):
        """rows [n, n + m) of the backing storage, growing it when needed (exact fit unless reserved larger)"""
        need = self._n + m
        if self._s32 is None or self._s32.shape[0] < need or (want_lo and self._slo is None):
            keep = getattr(self, "_reserved", False)
            self.reserve(need, lo=want_lo or self._slo is not None)
            self._reserved = keep
        a, b = self._n, need
        return self._s32[a:b], self._s16[a:b], (self._slo[a:b] if self._slo is not None else None)

    def add(self, x, chunk_bytes=None):
        """x: numpy / torch [n, d] float32 (host or device).  Appends to the index.
        A HOST array -- typically the memory-mapped payload of a block file (blocks.BlockView) -- is streamed: chunks of
        ~chunk_bytes (default 64 MB) go through pinned staging buffers, the H2D copy of chunk i + 1 (copy stream) runs under the
        centring / rounding / norm pass of chunk i (convdr_ip_prepare_block*), and the host fills one staging buffer
        (page faults on the mmap = the disk read) while the others are in flight.  The reference does pickle.load (a full
        host copy of the 14.6 GB block) and a pageable copy (run_convdr_inference.py:164-180)."""
        import torch
        chunk_bytes = int(chunk_bytes or self.host_chunk_bytes)
        if self.d != self.d_in:
            x = self._pad_columns(x.array if hasattr(x, "array") else x)
        if hasattr(x, "read_rows_into") and hasattr(x, "array"):        # a blocks.BlockView: positioned reads from the file
            arr = x.array
            if arr.dtype == np.float32 and arr.ndim == 2 and arr.nbytes > chunk_bytes // 2:
                return self._add_host_streamed(arr, chunk_bytes, reader=x.read_rows_into)
            x = arr
        if isinstance(x, np.ndarray) or (isinstance(x, torch.Tensor) and x.device.type == "cpu" and not x.is_pinned()):
            arr = x if isinstance(x, np.ndarray) else x.numpy()
            if arr.dtype == np.float32 and arr.ndim == 2 and arr.shape[0] * arr.shape[1] * 4 > chunk_bytes // 2:
                return self._add_host_streamed(arr, chunk_bytes)
        from_host = not (isinstance(x, torch.Tensor) and x.is_cuda)
        if isinstance(x, np.ndarray) and not x.flags.writeable:
            # a read-only array (the mmap of a small block file): torch.as_tensor would alias it without knowing it must
            # never write (a UserWarning today, undefined behaviour the day someone does) -- small by construction (large
            # host blocks took the streamed path above), so it is copied first
            x = np.array(x, dtype=np.float32, order="C")
        t = torch.as_tensor(x)
        if t.dtype != torch.float32:
            t = t.float()
        # (a host source that is not pinned -- e.g. the mmap of a small block file, which search_one_by_one closes right
        #  after add() returns -- is copied synchronously: nothing may still be reading it when add() is back)
        t = t.to(self.device, non_blocking=not from_host or t.is_pinned()).contiguous()
        assert t.dim() == 2 and t.shape[1] == self.d, "expected [n, %d], got %s" % (self.d, tuple(t.shape))
        m = int(t.shape[0])
        if m == 0:
            return
        want_lo = self.precision in ("bf16x3", "fp16x3") or self._slo is not None
        with torch.cuda.device(self.device):
            if self._n == 0:
                self._first_rows(t)
            if self._s32 is None and not want_lo:
                # first block of an unreserved index: adopt the caller's device tensor instead of copying it
                self._s32 = t
                self._s16 = torch.empty((m, self.d), dtype=self._half_dtype, device=self.device)
                dst32, dst16, dstlo = self._s32, self._s16, None
            else:
                dst32, dst16, dstlo = self._grow_for(m, want_lo)
                dst32.copy_(t)
            self._prepare_into(dst32, dst16, dstlo)
        self._n += m

    def _pad_columns(self, x):
        import torch
        t = torch.as_tensor(x)
        t = (t if t.dtype == torch.float32 else t.float()).to(self.device)
        assert t.dim() == 2 and t.shape[1] == self.d_in, "expected [n, %d], got %s" % (self.d_in, tuple(t.shape))
        return torch.nn.functional.pad(t, (0, self.d - self.d_in))

    def _set_centre(self, t):
        import torch
        self._centre = torch.empty(self.d, dtype=torch.float32, device=self.device)
        scratch = torch.empty(1024 * self.d, dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().convdr_ip_column_mean(_lib.ptr(t), t.shape[0], self.d, _lib.ptr(scratch),
                                                   _lib.ptr(self._centre), _lib.stream_ptr()), "convdr_ip_column_mean")

    def _add_host_streamed(self, arr, chunk_bytes, reader=None):
        import time
        import torch
        n, d = arr.shape
        assert d == self.d, "expected [n, %d], got %s" % (self.d, arr.shape)
        rows_per = max(1, int(chunk_bytes) // (4 * d))
        want_lo = self.precision in ("bf16x3", "fp16x3") or self._slo is not None
        nbuf = max(2, int(self.host_stage_buffers))
        t0 = time.perf_counter()
        with torch.cuda.device(self.device):
            main = torch.cuda.current_stream()
            if getattr(self, "_copy_stream", None) is None:
                self._copy_stream = torch.cuda.Stream(device=self.device)
            stage, freed = _staging(self.device, min(rows_per, n), d, nbuf)
            cs = self._copy_stream
            p32, p16, plo = self._grow_for(n, want_lo)
            cs.wait_stream(main)                    # (allocation order / the copy of the old rows in _grow_for)
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            threads = max(1, min(int(self.host_copy_threads), avail))
            pool = _copy_pool(threads * (nbuf - 1), self.device)
            chunks = [(s, min(n, s + rows_per)) for s in range(0, n, rows_per)]

            def fill(ci):
                """start filling staging buffer ci % nbuf with chunk ci: `threads` positioned reads / memcpy slices"""
                s, e = chunks[ci]
                bi = ci % nbuf
                # staging buffer bi may be refilled once its previous H2D copy has completed.  The events live with the
                # buffers (module-level, shared by every index of the process): the last copies of one add() are still in
                # flight when the next add() starts filling (1 run in ~500 put rows of a second block into the first
                # before they did, tests/test_ip_search_gpu.py::test_back_to_back_streamed_adds_keep_their_rows)
                if freed[bi] is not None:
                    freed[bi].synchronize()
                dst = stage[bi].numpy()[:e - s]
                if reader is not None:
                    return reader(dst, s, e, pool=pool, parts=threads, wait=False)
                step = (e - s + threads - 1) // threads
                return [pool.submit(np.copyto, dst[a:a + step], arr[s + a:min(e, s + a + step)]) for a in range(0, e - s, step)]
            first = self._n == 0
            inflight = {ci: fill(ci) for ci in range(min(nbuf - 1, len(chunks)))}
            for ci, (s, e) in enumerate(chunks):
                for f in inflight.pop(ci):
                    f.result()
                bi = ci % nbuf
                with torch.cuda.stream(cs):
                    p32[s:e].copy_(stage[bi][:e - s], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(cs)
                freed[bi] = ev
                main.wait_event(ev)
                if ci + nbuf - 1 < len(chunks):
                    # nbuf - 1 chunks are being read while this one crosses PCIe (the copy stream never waits for the host;
                    # with two buffers -- round 2 -- the reads of chunk i + 1 only started after chunk i had been enqueued)
                    inflight[ci + nbuf - 1] = fill(ci + nbuf - 1)
                if first and ci == 0:
                    # centre = column mean of the first chunk (>= 40 k passages): any centre keeps the search exact -- it
                    # shifts every score of a query by the same constant -- it only has to be close to the mean to shrink
                    # the rounding-error band; likewise the fp16 scale comes from the first chunk's norms
                    self._first_rows(p32[s:e])
                self._prepare_into(p32[s:e], p16[s:e], plo[s:e] if plo is not None else None)
            self._n += n
        self.stats["add_host_s"] = time.perf_counter() - t0     # host time to enqueue (the last chunks are still in flight)
        self.stats["add_host_bytes"] = n * d * 4

    def _ensure_lo(self):
        """Remainder copy for the split scan, built on first use."""
        import torch
        if self._slo is None and self._s32 is not None:
            with torch.cuda.device(self.device):
                self._slo = torch.empty((self._s32.shape[0], self.d), dtype=self._half_dtype, device=self.device)
                if self._n:
                    self._prepare_into(self._p32, self._pbf, self._plo)

    def update_rows(self, row0, emb):
        """Overwrite rows [row0, row0 + len(emb)) of the resident block with freshly encoded embeddings
        (device fp32 [m, d]) and refresh their scan copy / the block's max norm.  No sync."""
        import torch
        m = int(emb.shape[0])
        assert emb.dtype == torch.float32 and emb.is_contiguous() and row0 + m <= self.ntotal
        dst32, dstbf = self._p32[row0:row0 + m], self._pbf[row0:row0 + m]
        dstlo = None if self._slo is None else self._plo[row0:row0 + m]
        dst32[:, :self.d_in].copy_(emb)      # (zero columns of a padded width stay zero)
        with torch.cuda.device(self.device):
            self._prepare_into(dst32, dstbf, dstlo)

    def _workspace(self, nbytes):
        import torch
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws

    def _search_call(self, q, nq, p32, p16, plo, n, k, tau_in, cap, rank_target, ws, D, I, status, tau_retry):
        L = _lib.lib()
        if self.kind == "f16":
            _lib.check(L.convdr_ip_search_f16(_lib.ptr(q), nq, _lib.ptr(p32), _lib.ptr(p16), _lib.ptr(plo), float(self._scale), n,
                                              self.d, k, _lib.ptr(self._max_norm), _lib.ptr(tau_in), cap, rank_target,
                                              _lib.ptr(ws), ws.numel(), _lib.ptr(D), _lib.ptr(I), _lib.ptr(status),
                                              _lib.ptr(tau_retry), _lib.stream_ptr()), "convdr_ip_search_f16")
        else:
            _lib.check(L.convdr_ip_search(_lib.ptr(q), nq, _lib.ptr(p32), _lib.ptr(p16), _lib.ptr(plo), n, self.d, k,
                                          _lib.ptr(self._max_norm), _lib.ptr(tau_in), cap, rank_target,
                                          _lib.ptr(ws), ws.numel(), _lib.ptr(D), _lib.ptr(I), _lib.ptr(status),
                                          _lib.ptr(tau_retry), _lib.stream_ptr()), "convdr_ip_search")

    def search_device(self, q, k, tau_in=None, cap=None, x3=None):
        """One enqueue of the kernel pipeline; q is a device fp32 [nq, d] tensor.
        Returns device tensors (D, I, status, tau_retry); no sync.
        status (per query): 0 = certified exact; 1 / 2 / 3 = not certified, re-run with tau_retry (search_tensors walks that
        ladder); 4 = CONVDR_IP_RANGE, fp16 rungs only: the scan copy was built with a scale that later, longer rows (or an
        astronomically long query) overflow -- D / I of such a query are NOT usable, and no retry with another threshold
        helps: call ``_rebuild_scaled()`` (search_tensors / search_finish do) and search again.  Callers that take one
        uncertified pass (search_sharded_device(certify=False), the C ABI) must treat any non-zero status as "no result"."""
        import torch
        L = _lib.lib()
        cap = cap or self.cap
        while cap < 2 * k and cap < 8192:      # the candidate list holds at least 2k entries (csrc/ip_topk.hip: k <= cap / 2)
            cap *= 2
        if x3 is None:
            x3 = self.precision in ("bf16x3", "fp16x3")
        if x3:
            self._ensure_lo()
        nq, n = int(q.shape[0]), self.ntotal
        D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        status = torch.empty(nq, dtype=torch.int32, device=self.device)
        tau_retry = torch.empty(nq, dtype=torch.float32, device=self.device)
        need = L.convdr_ip_workspace_bytes(nq, n, self.d, k, cap)
        ws = self._workspace(need)
        p32 = self._p32 if n else q  # never dereferenced when n == 0
        pbf = self._pbf if n else q
        plo = self._plo if (x3 and n) else None
        with torch.cuda.device(self.device):
            self._search_call(q, nq, p32, pbf, plo, n, k, tau_in, cap, self.rank_target, ws, D, I, status, tau_retry)
        return D, I, status, tau_retry

    def last_counts(self, nq, k, cap=None):
        """(emitted, band) int32 tensors [nq] of the last search_device call (instrumentation)."""
        L = _lib.lib()
        cap = cap or self.cap
        out = []
        for fn in (L.convdr_ip_debug_counts, L.convdr_ip_debug_band):
            off = fn(_lib.ptr(self._ws), nq, self.ntotal, self.d, k, cap) - self._ws.data_ptr()
            out.append(self._ws[off:off + 4 * nq].view(_torch().int32))
        return tuple(out)

    def _certify(self, qt, k, D, I, status, tau_retry, x3):
        """Host loop around the kernel's certificate: re-run the queries that are not OK with the threshold the kernel
        proposes (and a larger candidate capacity when needed).  Returns the indices still uncertified."""
        import torch
        st = status.cpu().numpy()
        cap = self.cap
        bad = np.nonzero(st != 0)[0]
        rounds = 0
        while len(bad) and rounds < 6:
            rounds += 1
            self.stats["rounds"] += 1
            idx = torch.as_tensor(bad, device=self.device)
            tau = tau_retry[idx].contiguous()
            if (st[bad] == STATUS_OVERFLOW).any():
                if cap >= 8192:
                    break
                cap *= 2
            Db, Ib, sb, tb = self.search_device(qt[idx].contiguous(), k, tau_in=tau, cap=cap, x3=x3)
            D[idx], I[idx], tau_retry[idx] = Db, Ib, tb
            sb = sb.cpu().numpy()
            st[bad] = sb
            bad = bad[sb != 0]
        return bad

    def search(self, q, k):
        """FAISS ``index.search``: numpy in, numpy (D, I) out; always the exact top-k (queries no scan can certify take the
        exhaustive rung, `stats["exhaustive_queries"]`)."""
        D, I = self.search_tensors(q, k)
        return D.cpu().numpy(), I.cpu().numpy()

    def search_tensors(self, q, k):
        """``search`` with the certified result left on the device (torch fp32 [nq, k], int64 [nq, k]).

        precision="auto": the fp16 scan first; queries it cannot certify are re-run -- with a lower threshold while their
        error band still fits the candidate list, with the split scan (4x tighter band) once the band has swallowed
        the whole list.  The index remembers when most queries of a block ended on the split scan and starts there next
        time (`x3_first`)."""
        return self.search_finish(self.search_begin(q, k))

    def search_begin(self, q, k):
        """First half of search_tensors: the first scan pass is ENQUEUED (no host round trip) and a handle returned;
        search_finish(handle) reads the certificates and walks the ladder for whatever the first pass left open.  Between the
        two the host is free -- search_one_by_one loads the next block file meanwhile."""
        import torch
        qt = torch.as_tensor(q)
        if qt.dtype != torch.float32:
            qt = qt.float()
        qt = (self._pad_columns(qt) if self.d != self.d_in else qt.to(self.device)).contiguous()
        assert qt.dim() == 2 and qt.shape[1] == self.d
        k = int(k)
        if k > self.MAX_K:
            # the reference takes any --top_n (run_convdr_inference.py:316-319); the kernel pipeline's candidate lists end at
            # 8192 entries, so larger k takes the chunked host-side route (exact, slow): see _search_large_k
            return (qt, k, None, None, None, None, None)
        x3 = self.precision in ("bf16x3", "fp16x3") or (self.precision == "auto" and getattr(self, "_x3_first", False) and self.ntotal > 0)
        return (qt, k, x3) + tuple(self.search_device(qt, k, x3=x3))

    def search_finish(self, handle):
        import torch
        qt, k, x3, D, I, status, tau_retry = handle
        if x3 is None:
            self.stats = {"retried": 0, "rounds": 0, "x3_queries": 0, "x3_first": False, "rescaled": 0, "large_k": k}
            return self._search_large_k(qt, k)
        nq = int(qt.shape[0])
        rescaled = 0
        n_range, n_bad = torch.stack([(status == STATUS_RANGE).sum(), (status != 0).sum()]).tolist()   # one host round trip
        if self.kind == "f16" and n_range:
            self._rebuild_scaled()
            rescaled = 1
            D, I, status, tau_retry = self.search_device(qt, k, x3=x3)
            n_bad = int((status != 0).sum().item())
        self.stats = {"retried": int(n_bad), "rounds": 1, "x3_queries": nq if x3 else 0, "x3_first": bool(x3),
                      "rescaled": rescaled}
        bad = []
        if self.stats["retried"]:
            if self.precision == "auto" and not x3:
                # a band that already covers every emitted candidate only grows with a lower threshold: those queries go
                # straight to the split scan, the others get their single-pass retries
                emitted, band = self.last_counts(nq, k)
                st = status.cpu().numpy()
                sat = ((band >= emitted) & (emitted > 0)).cpu().numpy() & (st == STATUS_UNCERTAIN)
                retry_idx = np.nonzero((st != 0) & ~sat)[0]
                bad = list(np.nonzero(sat)[0])
                if len(retry_idx):
                    sub = torch.as_tensor(retry_idx, device=self.device)
                    Db, Ib, sb, tb = D[sub], I[sub], status[sub], tau_retry[sub]
                    left = self._certify(qt[sub].contiguous(), k, Db, Ib, sb, tb, False)
                    D[sub], I[sub] = Db, Ib
                    bad += list(retry_idx[np.asarray(left, dtype=np.int64)]) if len(left) else []
                bad = np.asarray(sorted(bad), dtype=np.int64)
            else:
                bad = self._certify(qt, k, D, I, status, tau_retry, x3)
        if len(bad) and self.precision == "auto" and not x3:
            # second rung: split scan for the queries the single-pass error band cannot separate
            idx = torch.as_tensor(bad, device=self.device)
            qs = qt[idx].contiguous()
            self.stats["x3_queries"] = len(bad)
            Db, Ib, sb, tb = self.search_device(qs, k, x3=True)
            self.stats["rounds"] += 1
            bad2 = self._certify(qs, k, Db, Ib, sb, tb, True) if int((sb != 0).sum().item()) else []
            D[idx], I[idx] = Db, Ib
            bad = bad[np.asarray(bad2, dtype=np.int64)] if len(bad2) else []
        if self.precision == "auto":
            self._x3_first = self.stats["x3_queries"] > nq // 2
        if len(bad):
            # last rung: more than 8192 passages inside the error band of the k-th score even with the split scan
            # (blocks whose norms spread over orders of magnitude: eps scales with the LARGEST norm).  Every slice of
            # <= cap rows is searched with all of its rows as candidates -- exact by construction -- and the slices are
            # merged in row order (earlier rows win ties): slow (one small launch chain per slice) but always an answer
            idx = torch.as_tensor(np.asarray(bad, dtype=np.int64), device=self.device)
            Db, Ib = self._search_exhaustive(qt[idx].contiguous(), k)
            D[idx], I[idx] = Db, Ib
            self.stats["exhaustive_queries"] = len(bad)
        return D, I

    MAX_K = 4096         # convdr_ip_search: k <= cap / 2, cap <= 8192

    def _search_large_k(self, q, k, q_chunk=8):
        """Exact top-k for k > MAX_K (any --top_n, run_convdr_inference.py:316-319) by chunking on the host side of the
        same kernels: every slice of <= 4096 rows is RANKED COMPLETELY by the exhaustive plan (all rows candidates, canonical
        fp64 scores, kq = rows), the slices' sorted lists are merged by a stable descending sort of the fp32-rounded scores
        (rounding is monotone: only rows whose scores round to the SAME fp32 value can be out of canonical order, and only
        across slices), and every run of equal fp32 scores that spans slices or straddles rank k is ranked once more as one
        slice.  Slow (n / 4096 launch chains per 8 queries, an [8, n] sort), always the exhaustive exact answer."""
        import torch
        L = _lib.lib()
        nq, n = int(q.shape[0]), self.ntotal
        Dout = torch.full((nq, k), -3.4028234663852886e38, dtype=torch.float32, device=self.device)
        Iout = torch.full((nq, k), -1, dtype=torch.int64, device=self.device)
        if n == 0:
            return Dout, Iout
        cap, step = 8192, 4096

        def exact(p32, pbf, m, kq, qq):
            D = torch.empty((qq.shape[0], kq), dtype=torch.float32, device=self.device)
            I = torch.empty((qq.shape[0], kq), dtype=torch.int64, device=self.device)
            status = torch.empty(qq.shape[0], dtype=torch.int32, device=self.device)
            tau_retry = torch.empty(qq.shape[0], dtype=torch.float32, device=self.device)
            ws = self._workspace(L.convdr_ip_workspace_bytes(int(qq.shape[0]), m, self.d, kq, cap))
            self._search_call(qq, int(qq.shape[0]), p32, pbf, None, m, kq, None, cap, 0, ws, D, I, status, tau_retry)
            if int((status != 0).sum().item()):
                raise _lib.ConvdrError("convdr_ip_search: exhaustive slice of %d rows not certified" % m)
            return D, I

        with torch.cuda.device(self.device):
            for j0 in range(0, nq, q_chunk):
                qq = q[j0:j0 + q_chunk].contiguous()
                Ds, Is = [], []
                for s0 in range(0, n, step):
                    m = min(n, s0 + step) - s0
                    D, I = exact(self._p32[s0:s0 + m], self._pbf[s0:s0 + m], m, m, qq)
                    Ds.append(D)
                    Is.append(I + s0)
                Dall, Iall = torch.cat(Ds, 1), torch.cat(Is, 1)
                order = torch.sort(Dall, dim=1, descending=True, stable=True).indices
                Dall, Iall = torch.gather(Dall, 1, order), torch.gather(Iall, 1, order)
                kk = min(k, n)
                for j in range(qq.shape[0]):
                    d, i = Dall[j], Iall[j]
                    # end of the run of equal fp32 scores that contains rank kk - 1
                    end = kk + int((d[kk:] == d[kk - 1]).sum().item()) if kk < n else kk
                    d, i = d[:end].clone(), i[:end].clone()
                    # runs of equal scores (start, length) with more than one member
                    new = torch.ones(end, dtype=torch.bool, device=self.device)
                    new[1:] = d[1:] != d[:-1]
                    starts = torch.nonzero(new).flatten()
                    lens = torch.diff(torch.cat([starts, torch.tensor([end], device=self.device)]))
                    for b, ln in zip(starts[lens > 1].tolist(), lens[lens > 1].tolist()):
                        if ln > step:
                            raise _lib.ConvdrError("FlatIPIndex.search: %d passages share one fp32 score around rank %d; "
                                                   "k > %d cannot order a tie group that large" % (ln, b, self.MAX_K))
                        rows = torch.sort(i[b:b + ln]).values
                        Dj, Ij = exact(self._p32[rows].contiguous(), self._pbf[rows].contiguous(), ln, ln, qq[j:j + 1].contiguous())
                        d[b:b + ln], i[b:b + ln] = Dj[0], rows[Ij[0]]
                    Dout[j0 + j, :kk], Iout[j0 + j, :kk] = d[:kk], i[:kk]
        return Dout, Iout

    def _search_exhaustive(self, q, k):
        import torch
        L = _lib.lib()
        nq, n = int(q.shape[0]), self.ntotal
        kk = min(2 * k, 4096)           # candidates carried through the merges (see the re-ranking below)
        cap = 4096
        while cap < 2 * kk:
            cap *= 2
        step = cap                      # n <= cap: the plan takes every row as a candidate (no threshold pass, tau = -inf)

        def exact(p32, pbf, m, kq, qq):
            D = torch.empty((qq.shape[0], kq), dtype=torch.float32, device=self.device)
            I = torch.empty((qq.shape[0], kq), dtype=torch.int64, device=self.device)
            status = torch.empty(qq.shape[0], dtype=torch.int32, device=self.device)
            tau_retry = torch.empty(qq.shape[0], dtype=torch.float32, device=self.device)
            ws = self._workspace(L.convdr_ip_workspace_bytes(int(qq.shape[0]), m, self.d, kq, cap))
            self._search_call(qq, int(qq.shape[0]), p32, pbf, None, m, kq, None, cap, 0, ws, D, I, status, tau_retry)
            if int((status != 0).sum().item()):
                raise _lib.ConvdrError("convdr_ip_search: exhaustive slice of %d rows not certified" % m)
            return D, I

        merged = None
        with torch.cuda.device(self.device):
            for s0 in range(0, n, step):
                e0 = min(n, s0 + step)
                D, I = exact(self._p32[s0:e0], self._pbf[s0:e0], e0 - s0, kk, q)
                I = torch.where(I >= 0, I + s0, I)
                merged = (D, I) if merged is None else tuple(t[:, :kk].contiguous() for t in merge_topk_device(merged, (D, I), kk))
            # The merges compare the fp32-rounded scores; the result's order is defined on the canonical fp64 scores (two
            # rows whose scores round to the same fp32 value are NOT a tie).  So each query's <= 2k survivors -- a superset
            # of its top-k: rounding is monotone -- are gathered in row order and ranked once more, exactly, as one slice.
            Dout = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            Iout = torch.empty((nq, k), dtype=torch.int64, device=self.device)
            for j in range(nq):
                rows = merged[1][j]
                rows = torch.sort(rows[rows >= 0]).values
                m = int(rows.numel())
                if m == 0:
                    Dout[j] = -3.4028234663852886e38
                    Iout[j] = -1
                    continue
                Dj, Ij = exact(self._p32[rows].contiguous(), self._pbf[rows].contiguous(), m, k, q[j:j + 1].contiguous())
                Dout[j] = Dj[0]
                Iout[j] = torch.where(Ij[0] >= 0, rows[Ij[0].clamp_min(0)], Ij[0])
        return Dout, Iout


# Pinned staging buffers and the copy thread pool are process-wide: pinning 64 MB costs ~20 ms (hipHostMalloc), i.e. a
# fresh set per index would cost as much as loading a 3 GB block through them.
_STAGING = {}
_STAGING_THREADS = {}
_POOLS = {}
import threading as _threading
_STAGING_LOCK = _threading.Lock()        # guards the dictionaries
_STAGING_BUILD = _threading.Lock()       # held while buffers are allocated and pinned
_NUMA = {}


def gpu_numa_cpus(device):
    """CPUs of the NUMA node the GPU hangs off (None when the topology cannot be read, or on a one-node host).
    A pinned buffer is placed where its allocating thread runs; across the socket link the same H2D copy runs at 29 GB/s
    instead of 57 (measured on a 2-socket MI355X host, tools/dbg/loader_probe.py) -- which is why the loader's rate, and the
    'pinned H2D ceiling' beside it, used to differ by a factor of two between boxes of one pool."""
    import torch
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx in _NUMA:
        return _NUMA[idx]
    cpus = None
    try:
        import ctypes
        buf = ctypes.create_string_buffer(64)
        # (through libconvdr_hip.so, which is bound to the HIP runtime torch loaded: dlopen-ing "libamdhip64.so" by name
        #  brings a SECOND runtime into the process, which slowed every later launch -- the training step went host-bound)
        if _lib.lib().convdr_device_pci_bus_id(int(idx), buf, 64) == 0:
            bdf = buf.value.decode().lower()
            node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
            import glob
            if node >= 0 and len(glob.glob("/sys/devices/system/node/node[0-9]*")) > 1:
                got = set()
                for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
                    lo, _, hi = part.partition("-")
                    got.update(range(int(lo), int(hi or lo) + 1))
                got &= os.sched_getaffinity(0)
                cpus = got or None
    except Exception:
        cpus = None
    _NUMA[idx] = cpus
    return cpus


class _on_cpus:
    """Confine the calling thread to `cpus` for the duration of the block (no-op for None)."""

    def __init__(self, cpus):
        self.cpus, self.old = cpus, None

    def __enter__(self):
        if self.cpus:
            try:
                self.old = os.sched_getaffinity(0)
                os.sched_setaffinity(0, self.cpus)
            except OSError:
                self.old = None

    def __exit__(self, *exc):
        if self.old is not None:
            os.sched_setaffinity(0, self.old)


def pinned_near(device, shape, dtype):
    """A pinned host tensor allocated (first-touched and locked) on the GPU's NUMA node."""
    import torch
    with _on_cpus(gpu_numa_cpus(device)):
        t = torch.empty(shape, dtype=dtype).pin_memory()
    return t


def _staging_key(device, d, nbuf):
    return (str(device), int(d), int(nbuf))


def _staging(device, rows, d, nbuf):
    """([nbuf pinned [rows, d] fp32 buffers], [their last H2D-copy events]) of `device`: process-wide, grown on demand.
    Serialised by _STAGING_BUILD (a background pre-pin -- FlatIPIndex.prepin_staging -- and the first add() may race)."""
    import torch
    key = _staging_key(device, d, nbuf)
    with _STAGING_BUILD:
        ent = _STAGING.get(key)
        if ent is None or ent[0][0].shape[0] < rows:
            ent = ([pinned_near(device, (rows, d), torch.float32) for _ in range(nbuf)], [None] * nbuf)
            with _STAGING_LOCK:
                _STAGING[key] = ent
    return ent


def _copy_pool(threads, device=None):
    """Reader threads of the block loader; confined to the GPU's NUMA node (their destination is the pinned staging there)."""
    cpus = gpu_numa_cpus(device) if device is not None else None
    key = (threads, None if cpus is None else min(cpus))
    pool = _POOLS.get(key)
    if pool is None:
        from concurrent.futures import ThreadPoolExecutor

        def init():
            if cpus:
                try:
                    os.sched_setaffinity(0, cpus)
                except OSError:
                    pass
        pool = _POOLS[key] = ThreadPoolExecutor(max_workers=threads, initializer=init)
    return pool


def load_block(path):
    """pickle.load, as run_convdr_inference.py:164-175 does."""
    with open(path, "rb") as h:
        return pickle.load(h)


def merge_topk(merged, cand, topN):
    """The reference's two-way list merge (run_convdr_inference.py:213-229) for all queries at
    once.  merged/cand: (D float64 [nq, >=topN], I int64 [nq, >=topN]), rows sorted descending.
    The reference walks two pointers until one list is exhausted and then appends the other
    list's remainder -- i.e. a complete merge of the two sorted topN-lists in which ties keep the
    earlier block first (`>=`, :218) -- and returns all 2*topN entries.  A stable sort of the
    concatenation on descending score is that same permutation."""
    aD = np.concatenate([merged[0][:, :topN], cand[0][:, :topN]], axis=1)
    aI = np.concatenate([merged[1][:, :topN], cand[1][:, :topN]], axis=1)
    order = np.argsort(-aD, axis=1, kind="stable")
    return np.take_along_axis(aD, order, 1), np.take_along_axis(aI, order, 1)


def merge_topk_device(merged, cand, topN):
    """``merge_topk`` on the device: (D fp32 [nq, na], I int64 [nq, na]) torch tensors in and out, same permutation
    (convdr_topk_merge: A before B on equal scores, each list in its own order)."""
    import torch
    Da, Ia = merged[0][:, :topN], merged[1][:, :topN]
    Db, Ib = cand[0][:, :topN], cand[1][:, :topN]
    nq, na, nb = Da.shape[0], Da.shape[1], Db.shape[1]
    Do = torch.empty((nq, na + nb), dtype=torch.float32, device=Da.device)
    Io = torch.empty((nq, na + nb), dtype=torch.int64, device=Da.device)
    with torch.cuda.device(Da.device):
        _lib.check(_lib.lib().convdr_topk_merge(_lib.ptr(Da), _lib.ptr(Ia), na, Da.stride(0), _lib.ptr(Db), _lib.ptr(Ib), nb,
                                                Db.stride(0), nq, na + nb, _lib.ptr(Do), _lib.ptr(Io), Do.stride(0),
                                                _lib.stream_ptr()), "convdr_topk_merge")
    return Do, Io


def search_one_by_one(ann_data_dir, gpu_index, query_embedding, topN, max_blocks=8, timings=None):
    """Block-by-block search + merge; same contract as the reference function (float64 scores, int64 offsets,
    2 * topN columns once two blocks have been merged).  With a FlatIPIndex the embedding block is memory-mapped
    (blocks.BlockView: no pickle.load copy) and streamed to HBM in pinned chunks (FlatIPIndex.add), and the per-block
    results, the offset lookup ``embid[I]`` and the running merge stay on the device; any other index object
    (``.add/.search/.reset``) takes the reference's host path.
    Device path, pipelined: TWO blocks are in flight -- the first scan pass of block i is enqueued (search_begin, no host
    round trip), then the host reads block i + 1 into the twin index (file -> pinned staging -> HBM on the copy stream)
    while the GPU searches block i; block i's certificates are read (search_finish), its result merged and its storage
    dropped only after that.  The reference's load -> add -> search -> merge -> reset is strictly serial (:157-242).
    HBM residency: with two blocks in flight TWO fp32 blocks and their 16-bit scan copies are resident at once (2 x 1.5 x the
    block: 44 GB for CAsT's 14.6 GB blocks, of 288 GB) -- pass an index that is not a FlatIPIndex-with-twin, or search block
    files one at a time with index.add(BlockView) / search / reset yourself, where that does not fit.
    timings (optional dict): filled with the wall seconds spent per stage."""
    import time
    from . import blocks
    on_device = hasattr(gpu_index, "search_begin")
    merged = None
    tm = {"load_add_s": 0.0, "search_finish_merge_s": 0.0, "blocks": 0, "bytes": 0}

    def paths(block_id):
        return (os.path.join(ann_data_dir, "passage__emb_p__data_obj_%d.pb" % block_id),
                os.path.join(ann_data_dir, "passage__embid_p__data_obj_%d.pb" % block_id))
    if on_device:
        import torch
        twins = [gpu_index, None]
        pending = None                       # (handle, index, embid) of the block whose first pass is in flight

        def finish(p):
            nonlocal merged
            handle, idx, ids = p
            D, I = idx.search_finish(handle)
            embid = torch.as_tensor(np.asarray(ids, dtype=np.int64), device=D.device)
            found = torch.where(I >= 0, embid[I.clamp_min(0)], I) if embid.numel() else I   # -1 padding when n < topN
            cand = (D, found)
            merged = cand if merged is None else merge_topk_device(merged, cand, topN)
            idx.reset()
        for block_id in range(max_blocks):
            emb_path, id_path = paths(block_id)
            try:
                view = blocks.BlockView(emb_path)
            except Exception:
                break
            try:
                try:
                    ids = load_block(id_path)
                except Exception:
                    break
                idx = twins[block_id & 1]
                if idx is None:
                    idx = twins[block_id & 1] = gpu_index.twin()
                t0 = time.perf_counter()
                idx.add(view)                # the host reads; the GPU meanwhile searches the previous block
                t1 = time.perf_counter()
                tm["load_add_s"] += t1 - t0
                tm["bytes"] += int(view.array.nbytes)
                tm["blocks"] += 1
                if pending is not None:
                    finish(pending)
                pending = (idx.search_begin(query_embedding, topN), idx, ids)
                tm["search_finish_merge_s"] += time.perf_counter() - t1
            finally:
                view.close()                 # (streamed blocks: read by host threads that add() has joined; small blocks:
                                             #  copied synchronously by add() -- nothing reads the mapping any more)
        if pending is not None:
            t1 = time.perf_counter()
            finish(pending)
            tm["search_finish_merge_s"] += time.perf_counter() - t1
        if timings is not None:
            timings.update(tm)
        if merged is None:
            raise FileNotFoundError("no passage blocks under %s" % ann_data_dir)
        return merged[0].double().cpu().numpy(), merged[1].cpu().numpy()
    for block_id in range(max_blocks):
        emb_path, id_path = paths(block_id)
        try:
            passage_embedding = load_block(emb_path)
            passage_embedding2id = load_block(id_path)
        except Exception:
            break
        gpu_index.add(passage_embedding)
        D, I = gpu_index.search(query_embedding, topN)
        cand = (D.astype(np.float64), np.asarray(passage_embedding2id)[I])
        merged = cand if merged is None else merge_topk(merged, cand, topN)
        gpu_index.reset()
    if merged is None:
        raise FileNotFoundError("no passage blocks under %s" % ann_data_dir)
    return merged


def EvalDevQuery(query_embedding2id, merged_D, dev_query_positive_id, I_nearest_neighbor, topN, output_file,
                 output_trec_file, offset2pid, raw_data_dir, output_query_type, raw_sequences=None,
                 load_collection=None):
    """Result writer with the reference's exact text output (run_convdr_inference.py:21-113).
    The offset -> pid mapping and the first-occurrence de-duplication (:56-69) run as array operations per query (one
    gather + one np.unique over the topN candidates) instead of the reference's Python loop over nq x topN entries."""
    ranked = {}
    raw = {}
    o2p = np.asarray(offset2pid)
    I_top = np.asarray(I_nearest_neighbor)[:, :topN]
    D_top = np.asarray(merged_D)[:, :topN]
    pids_all = o2p[I_top]                                   # offset -> pid for every candidate at once
    for query_idx in range(len(I_top)):
        query_id = query_embedding2id[query_idx]
        if query_id not in ranked:
            ranked[query_id] = [(0, 0)] * topN
        raw[query_id] = raw_sequences[query_idx]
        row = pids_all[query_idx]
        _, first = np.unique(row, return_index=True)        # first occurrence of every pid ...
        first.sort()                                        # ... in rank order
        scores = D_top[query_idx][first].tolist()
        for rank, (pid, score) in enumerate(zip(row[first].tolist(), scores)):
            ranked[query_id][rank] = (pid, score)
    queries = {}
    with open(os.path.join(raw_data_dir, "queries." + output_query_type + ".tsv")) as f:
        for line in f:
            qid, query = line.strip().split("\t")
            queries[qid] = query
    collection = os.path.join(raw_data_dir, "collection.jsonl")
    if not os.path.exists(collection):
        collection = os.path.join(raw_data_dir, "collection.tsv")
        if not os.path.exists(collection):
            raise FileNotFoundError("Neither collection.tsv nor collection.jsonl found in {}".format(raw_data_dir))
    all_passages = (load_collection or _load_collection)(collection)
    with open(output_file, "w") as f, open(output_trec_file, "w") as g:
        for qid, passages in ranked.items():
            for i in range(topN):
                pid, score = passages[i]
                label = 0 if qid not in dev_query_positive_id else dev_query_positive_id[qid].get(pid, 0)
                f.write(json.dumps({"query": queries[qid], "doc": all_passages[pid], "label": label,
                                    "query_id": str(qid), "doc_id": str(pid), "retrieval_score": score,
                                    "input": raw[qid]}) + "\n")
                g.write(str(qid) + " Q0 " + str(pid) + " " + str(i + 1) + " " + str(-i - 1 + 200) + " ance\n")


class _Passages(dict):
    def __missing__(self, key):
        return "[INVALID DOC ID]"


def _load_collection(collection_file):
    """utils/util.py:327-352 without the 50M-entry list: a dict with the same default."""
    all_passages = _Passages()
    ext = collection_file[collection_file.rfind(".") + 1:]
    if ext not in ["jsonl", "tsv"]:
        raise TypeError("Unrecognized file type")
    with open(collection_file) as f:
        for line in f:
            line = line.strip()
            if ext == "jsonl":
                obj = json.loads(line)
                all_passages[int(obj["id"])] = obj["title"] + "[SEP]" + obj["text"]
            else:
                try:
                    arr = line.split("\t")
                    all_passages[int(arr[0])] = arr[1].rstrip()
                except IndexError:
                    print("bad passage")
                except ValueError:
                    print("bad pid")
    return all_passages
