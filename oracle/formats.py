"""Oracle: the on-disk formats either side of the hot path (CPU, plain Python).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  * token cache ``passages`` + ``passages_meta``:
      writer /root/reference/data/tokenizing.py:41-57,116 (record = 4-byte
      big-endian length || L x int32 native-endian ids; meta json
      {"type","total_number","embedding_size"}),
      reader /root/reference/utils/util.py:355-405 (EmbeddingCache),
      record -> tensors /root/reference/data/tokenizing.py:133-161 (GetProcessingFn),
      rank sharding i % world == rank /root/reference/utils/util.py:422-424.
  * embedding blocks ``passage__emb_p__data_obj_{r}.pb`` /
    ``passage__embid_p__data_obj_{r}.pb``: ``pickle.dump(ndarray, protocol=4)``
      /root/reference/utils/util.py:108-111, read back with ``pickle.load``
      /root/reference/drivers/run_convdr_inference.py:164-175.
"""
import json
import pickle

import numpy as np


def write_token_cache(path, ids_list, max_seq_length):
    """ids_list: list of python lists of token ids (already truncated).  Pads
    with 0 (utils/util.py:146-160 pad_token=0) exactly like PassagePreprocessingFn."""
    with open(path, "wb") as f:
        for ids in ids_list:
            n = min(len(ids), max_seq_length)
            padded = (list(ids) + [0] * max_seq_length)[:max_seq_length]
            f.write(n.to_bytes(4, "big") + np.array(padded, np.int32).tobytes())
    with open(path + "_meta", "w") as f:
        json.dump({"type": "int32", "total_number": len(ids_list),
                   "embedding_size": max_seq_length}, f)


def read_token_cache(path):
    """-> (lengths int64 [N], ids int32 [N, L]) via the EmbeddingCache rules."""
    with open(path + "_meta") as f:
        meta = json.load(f)
    dt = np.dtype(meta["type"])
    N, L = meta["total_number"], int(meta["embedding_size"])
    rec = L * dt.itemsize + 4
    lens = np.zeros(N, np.int64)
    ids = np.zeros((N, L), dt)
    with open(path, "rb") as f:
        for i in range(N):
            b = f.read(rec)
            lens[i] = int.from_bytes(b[:4], "big")
            ids[i] = np.frombuffer(b[4:], dtype=dt)
    return lens, ids


def processing_fn(passage_len, passage, max_len):
    """GetProcessingFn(query=False): attention mask = [1]*len + [0]*pad."""
    pad_len = max(0, max_len - passage_len)
    return np.asarray(passage, np.int32), np.array([1] * passage_len + [0] * pad_len, bool)


def shard_indices(total, world, rank):
    """StreamingDataset: record i goes to rank i % world (util.py:422-424)."""
    return [i for i in range(total) if i % world == rank]


def dump_block(path, array):
    with open(path, "wb") as h:
        pickle.dump(array, h, protocol=4)


def load_block(path):
    with open(path, "rb") as h:
        return pickle.load(h)
