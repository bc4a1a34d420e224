"""End-to-end rate of the corpus-encode loop (token cache on disk -> mmap reader -> token-budget batcher -> pinned
staging -> encoder -> pinned fp32 block), i.e. gen_passage_embeddings.py:73-127 with the host side included."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from convdr_amd import blocks, encode

N, L = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000, 128
ragged = len(sys.argv) > 2 and sys.argv[2] == "ragged"
rs = np.random.RandomState(0)
lens = rs.randint(24, L + 1, size=N) if ragged else np.full(N, L)
rec = np.zeros((N, 4 + 4 * L), np.uint8)
rec[:, :4] = np.stack([(lens >> s) & 255 for s in (24, 16, 8, 0)], 1).astype(np.uint8)
ids = rs.randint(3, 50000, size=(N, L)).astype(np.int32)
ids[:, 0] = 0
ids[np.arange(L)[None, :] >= lens[:, None]] = 0
rec[:, 4:] = ids.view(np.uint8).reshape(N, 4 * L)
d = tempfile.mkdtemp()
base = os.path.join(d, "passages")
rec.tofile(base)
json.dump({"type": "int32", "total_number": N, "embedding_size": L}, open(base + "_meta", "w"))
model = bench.random_rdot_model(0).cuda().eval()
with blocks.TokenCache(base) as cache:
    for tb, bs in ((262144, 8192),):
        encode.encode_shard(model, cache, batch_size=bs, token_budget=tb, max_seq_length=L)   # warm-up pass (page cache, packing)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        emb, embid = encode.encode_shard(model, cache, batch_size=bs, token_budget=tb, max_seq_length=L)
        el = time.perf_counter() - t0
        print("encode_shard: %d passages (%s lengths, %.1f M real tokens) in %.2f s = %.0f passages/s, %.2f M tokens/s" % (
            N, "ragged 24-128" if ragged else "128", lens.sum() / 1e6, el, N / el, lens.sum() / el / 1e6), flush=True)
