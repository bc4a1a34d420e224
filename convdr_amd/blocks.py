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

    def close(self):
        self.array = None
        if self._mm is not None:
            self._mm.close()
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
