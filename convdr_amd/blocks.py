"""On-disk formats either side of the hot path (host side, no arithmetic).

  * token cache  ``passages`` + ``passages_meta``  -- written by the reference's tokenizer driver
    (/root/reference/data/tokenizing.py:41-57,116), read there by ``EmbeddingCache``
    (/root/reference/utils/util.py:355-405).  Record = 4-byte big-endian length || L x int32 ids.
    ``TokenCache`` memory-maps the file instead of seek+read per record.
  * embedding blocks ``passage__emb_p__data_obj_{rank}.pb`` / ``passage__embid_p__data_obj_{rank}.pb``
    = ``pickle.dump(ndarray, protocol=4)`` (/root/reference/utils/util.py:108-111), consumed with
    ``pickle.load`` (/root/reference/drivers/run_convdr_inference.py:164-175).
    ``dump_block`` streams a byte-identical pickle straight from the array's buffer (numpy's own
    ``__reduce__`` copies the payload first: 14.6 GB per block at CAsT scale, SURVEY.md §7 hard part 9);
    ``BlockView`` memory-maps the payload of such a file (offset is header dependent and unaligned).
"""
import io
import json
import mmap
import os
import pickle
import pickletools
import struct

import numpy as np

_BE32 = np.array([1 << 24, 1 << 16, 1 << 8, 1], np.int64)


class TokenCache:
    """mmap view of the fixed-width token cache: ``lengths()`` int64 [N], ``ids`` int32 [N, L]."""

    def __init__(self, base_path):
        with open(base_path + "_meta") as f:
            meta = json.load(f)
        self.dtype = np.dtype(meta["type"])
        self.total_number = int(meta["total_number"])
        self.seq_len = int(meta["embedding_size"])
        self.record_size = self.seq_len * self.dtype.itemsize + 4
        self._f = open(base_path, "rb")
        size = os.fstat(self._f.fileno()).st_size
        need = self.total_number * self.record_size
        if size < need:
            raise ValueError("token cache %s holds %d bytes, its meta needs %d" % (base_path, size, need))
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ) if size else None
        raw = np.frombuffer(self._mm, np.uint8, need) if need else np.zeros(0, np.uint8)
        rec = raw.reshape(self.total_number, self.record_size)
        self._len_be = rec[:, :4]
        self.ids = rec[:, 4:].view(self.dtype)       # [N, L], zero-copy

    def __len__(self):
        return self.total_number

    def lengths(self, idx=None):
        b = self._len_be if idx is None else self._len_be[idx]
        return (b.astype(np.int64) * _BE32).sum(-1)

    def close(self):
        self.ids = self._len_be = None
        if self._mm is not None:
            self._mm.close()
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def shard_indices(total, world, rank):
    """Record i is encoded by rank i % world (StreamingDataset, utils/util.py:422-424)."""
    return np.arange(rank, total, world, dtype=np.int64)


# ---- pickle protocol-4 ndarray blocks -------------------------------------------------------------
_OPS = {op.code.encode("latin1"): op for op in pickletools.opcodes}


def _walk(buf, start=0):
    """Yield (opcode, arg, pos, next_pos) without ever touching a large payload: stops at the first
    BINBYTES / BINBYTES8 (whose arg is reported as its length)."""
    bio = io.BytesIO(buf)
    i = start
    while i < len(buf):
        op = _OPS[buf[i:i + 1]]
        if op.name == "BINBYTES":
            yield op, struct.unpack("<I", buf[i + 1:i + 5])[0], i, i + 5
            return
        if op.name == "BINBYTES8":
            yield op, struct.unpack("<Q", buf[i + 1:i + 9])[0], i, i + 9
            return
        bio.seek(i + 1)
        arg = op.arg.reader(bio) if op.arg is not None else None
        yield op, arg, i, bio.tell()
        i = bio.tell()
        if op.name == "STOP":
            return


def _encode_int(n):
    """The pickler's integer encodings (save_long)."""
    if 0 <= n <= 0xff:
        return b"K" + struct.pack("<B", n)
    if 0 <= n <= 0xffff:
        return b"M" + struct.pack("<H", n)
    if -(1 << 31) <= n < (1 << 31):
        return b"J" + struct.pack("<i", n)
    raw = n.to_bytes((n.bit_length() + 8) // 8, "little", signed=True)
    return b"\x8a" + struct.pack("<B", len(raw)) + raw


def _template(arr):
    """(head, tail) opcode bytes that surround the raw payload in ``pickle.dumps(arr, protocol=4)`` for a large
    array: derived from the pickle of a ~1 MiB array of the same dtype / trailing dims by patching the leading
    dimension, the frame length and the payload length."""
    row_bytes = max(1, arr.dtype.itemsize * int(np.prod(arr.shape[1:], dtype=np.int64)))
    rows = (1 << 20) // row_bytes + 300
    while rows in arr.shape[1:] or rows in (0, 1, 3):
        rows += 1
    blob = pickle.dumps(np.zeros((rows,) + arr.shape[1:], arr.dtype), protocol=4)
    ops = list(_walk(blob))
    op, plen, ppos, pend = ops[-1]
    assert op.name == "BINBYTES" and plen == rows * row_bytes, "unexpected pickle layout"
    assert blob[:2] == b"\x80\x04" and blob[2:3] == b"\x95", "expected a framed protocol-4 pickle"
    head = bytearray(blob[:ppos])
    for o, v, p, nxt in ops:
        if o.name in ("BININT", "BININT1", "BININT2", "LONG1") and v == rows:
            head[p:nxt] = _encode_int(int(arr.shape[0]))
            break
    else:
        raise AssertionError("leading dimension not found in the pickle header")
    head[3:11] = struct.pack("<Q", len(head) - 11)          # first frame = everything up to the payload opcode
    n = arr.nbytes
    head += (b"\x8e" + struct.pack("<Q", n)) if n > 0xffffffff else (b"B" + struct.pack("<I", n))
    return bytes(head), blob[pend + plen:]


def dump_block(path, arr, chunk_bytes=1 << 26):
    """Write ``arr`` so that the file is byte-identical to ``pickle.dump(arr, handle, protocol=4)``."""
    arr = np.asarray(arr)
    if not arr.flags.c_contiguous:
        arr = np.ascontiguousarray(arr)
    if arr.ndim == 0 or arr.nbytes < (1 << 20) or arr.dtype.hasobject:
        with open(path, "wb") as h:
            pickle.dump(arr, h, protocol=4)
        return
    head, tail = _template(arr)
    mv = memoryview(arr).cast("B")
    with open(path, "wb") as h:
        h.write(head)
        for s in range(0, arr.nbytes, chunk_bytes):
            h.write(mv[s:s + chunk_bytes])
        h.write(tail)


class _State:
    def __setstate__(self, st):
        self.state = st


class _MetaUnpickler(pickle.Unpickler):
    """Rebuilds only (shape, dtype) of a pickled ndarray; refuses anything else."""

    def find_class(self, module, name):
        if module.split(".")[0] == "numpy":
            if name == "_reconstruct":
                return lambda *a: _State()
            if name == "ndarray":
                return np.ndarray
            if name == "dtype":
                return np.dtype
        raise pickle.UnpicklingError("unexpected global %s.%s in an embedding block" % (module, name))


class BlockView:
    """Zero-copy, read-only view of the ndarray stored in a block file written by ``pickle.dump`` /
    ``dump_block``: ``.array`` is backed by an mmap of the payload (no 14.6 GB ``pickle.load`` copy)."""

    def __init__(self, path):
        self._f = open(path, "rb")
        head = self._f.read(1 << 16)
        last = None
        for last in _walk(head):
            pass
        if last is None or last[0].name not in ("BINBYTES", "BINBYTES8"):
            # small array pickled in-frame (SHORT_BINBYTES / in-frame BINBYTES handled above): just load it
            self._f.seek(0)
            self.array = pickle.load(self._f)
            self._mm = None
            return
        op, n, pos, end = last
        self._f.seek(end + n)
        tail = self._f.read()
        st = _MetaUnpickler(io.BytesIO(head[:pos] + b"C\x00" + tail)).load().state
        shape, dtype, fortran = st[1], st[2], st[3]
        if fortran:
            raise ValueError("Fortran-ordered block")
        count = int(np.prod(shape, dtype=np.int64))
        if count * dtype.itemsize != n:
            raise ValueError("payload size does not match shape/dtype")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        self.offset = end
        self.array = np.frombuffer(self._mm, dtype, count, end).reshape(shape)

    def read_rows_into(self, out, row0, row1, pool=None, parts=8, wait=True):
        """Copy rows [row0, row1) of the payload into `out` (a C-contiguous ndarray of that shape, e.g. a pinned staging
        buffer) with positioned reads straight from the file -- no page faults on the mapping, and `parts` slices in
        flight on `pool` (a concurrent.futures executor; os.preadv releases the GIL), so the copy runs at several memcpy
        streams instead of one fault-bound one (measured 4.4 GB/s through the mmap on the bench host)."""
        if self._mm is None:                        # small in-frame array: already in memory
            np.copyto(out, self.array[row0:row1])
            return []
        row_bytes = int(self.array.strides[0])
        mv = memoryview(out).cast("B")
        total = (row1 - row0) * row_bytes
        assert mv.nbytes >= total
        fd, base = self._f.fileno(), self.offset + row0 * row_bytes

        def rd(a, b):
            pos = a
            while pos < b:
                got = os.preadv(fd, [mv[pos:b]], base + pos)
                if got <= 0:
                    raise IOError("short read in block payload")
                pos += got
        if pool is None or parts <= 1 or total < (8 << 20):
            rd(0, total)
            return []
        step = (total // parts + 4095) // 4096 * 4096
        futs = [pool.submit(rd, a, min(total, a + step)) for a in range(0, total, step)]
        if not wait:
            return futs          # (the caller joins them: several chunks' reads in flight, FlatIPIndex._add_host_streamed)
        for f in futs:
            f.result()
        return []

    def close(self):
        self.array = None
        if self._mm is not None:
            self._mm.close()
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class DocEmbeddingLookup:
    """Passage embeddings of the frozen teacher by passage id, straight from the corpus blocks.

    The ranking step of the reference re-encodes its 1 positive + ``num_negatives`` documents per sample with the teacher
    (10 x 512 tokens per sample per step, /root/reference/drivers/run_convdr_train.py:118-159) although those embeddings
    already exist: ``gen_passage_embeddings.py`` wrote ``body_emb`` of every passage into
    ``passage__emb_p__data_obj_{r}.pb`` next to its record offset in ``passage__embid_p__data_obj_{r}.pb``, and the ranking
    file carries the passage ids (``doc_pos_id`` / ``doc_negs_id``, data/gen_ranking_data.py:595-600).  This class maps
    pid -> record offset (``pid2offset``, data/tokenizing.py:63-74) -> (block, row) and gathers the rows from the
    memory-mapped block payloads (BlockView): no block is ever loaded whole.

    pid2offset: dict or array pid -> offset (None: the ids ARE offsets).  n_blocks: how many block pairs to open
    (default: every ``passage__emb_p__data_obj_{r}.pb`` present, r = 0, 1, ...)."""

    def __init__(self, ann_data_dir, pid2offset=None, n_blocks=None, prefix="passage_"):
        self.views, self.embids = [], []
        r = 0
        while n_blocks is None or r < n_blocks:
            emb = os.path.join(ann_data_dir, "%s_emb_p__data_obj_%d.pb" % (prefix, r))
            ids = os.path.join(ann_data_dir, "%s_embid_p__data_obj_%d.pb" % (prefix, r))
            if not (os.path.exists(emb) and os.path.exists(ids)):
                if n_blocks is not None:
                    raise FileNotFoundError(emb)
                break
            self.views.append(BlockView(emb))
            with open(ids, "rb") as h:
                self.embids.append(np.asarray(pickle.load(h), dtype=np.int64))
            r += 1
        if not self.views:
            raise FileNotFoundError("no passage blocks under %s" % ann_data_dir)
        total = int(max(int(e.max()) for e in self.embids if len(e)) + 1)
        self._block = np.full(total, -1, np.int16)          # record offset -> block
        self._row = np.zeros(total, np.int64)               # record offset -> row inside the block
        for b, e in enumerate(self.embids):
            self._block[e] = b
            self._row[e] = np.arange(len(e), dtype=np.int64)
        self.dim = int(self.views[0].array.shape[1])
        self.pid2offset = pid2offset

    def offsets(self, pids):
        pids = np.asarray(pids, dtype=np.int64).reshape(-1)
        if self.pid2offset is None:
            return pids
        if isinstance(self.pid2offset, dict):
            return np.fromiter((self.pid2offset[int(p)] for p in pids), np.int64, len(pids))
        return np.asarray(self.pid2offset)[pids].astype(np.int64)

    def gather(self, pids, out=None):
        """-> float32 [len(pids), dim] (host), row i = stored body_emb of passage pids[i]."""
        off = self.offsets(pids)
        if len(off) and (off.min() < 0 or off.max() >= len(self._block) or (self._block[off] < 0).any()):
            raise KeyError("passage id without an embedding in the blocks")
        if out is None:
            out = np.empty((len(off), self.dim), np.float32)
        blk, row = self._block[off], self._row[off]
        for b in np.unique(blk):
            sel = np.nonzero(blk == b)[0]
            order = np.argsort(row[sel], kind="stable")      # ascending rows: sequential-ish page access on the mmap
            out[sel[order]] = self.views[int(b)].array[row[sel][order]]
        return out

    def gather_device(self, pids, device):
        """gather() into a pinned buffer and one asynchronous H2D copy -> torch float32 [len(pids), dim] on `device`."""
        import torch
        n = int(np.asarray(pids).size)
        buf = torch.empty((n, self.dim), dtype=torch.float32).pin_memory()
        self.gather(pids, out=buf.numpy())
        return buf.to(device, non_blocking=True)

    def close(self):
        for v in self.views:
            v.close()
        self.views = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
