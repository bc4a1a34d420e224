"""Latency of small-batch encodes (the evaluation loop's query batches, the frozen teacher's targets): ms per call, roberta-base."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from convdr_amd import _lib
dev = torch.device("cuda", 0)
model = bench.random_rdot_model(0).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(0)
for B, L in ((64, 64), (16, 128), (32, 256), (4, 64)):
    ids = torch.randint(3, 50000, (B, L), generator=g, device=dev); ids[:, 0] = 0
    lens = torch.randint(max(8, L // 4), L + 1, (B,), generator=g, device=dev)
    mask = (torch.arange(L, device=dev)[None, :] < lens[:, None]).long()
    ids = ids * mask
    hl = lens.cpu().numpy().astype(np.int32)
    for rep in range(2):
        for mode in (0, 1):
            _lib.lib().convdr_set_option(b"ffn2_splitk", mode)
            with torch.no_grad():
                for _ in range(5):
                    model(ids, mask, seq_lens=hl)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    model(ids, mask, seq_lens=hl)
                torch.cuda.synchronize()
            print("B=%3d L=%3d rows~%5d  ffn2_splitk=%d  %.3f ms per encode" % (B, L, int(hl.sum()), mode, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
_lib.lib().convdr_set_option(b"ffn2_splitk", 1)
