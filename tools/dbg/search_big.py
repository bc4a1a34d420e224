"""configs[3]'s per-GPU shard: 1000 queries over 4.75M x 768 (38M / 8)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from convdr_amd import _lib
from convdr_amd.search import FlatIPIndex
n, d, k, nq = 4_750_000, 768, 100, 1000
dev = torch.device("cuda")
idx = FlatIPIndex(d, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
for s in range(0, n, 250_000):
    idx.add(torch.randn(min(250_000, n - s), d, device=dev, generator=g))
Q = torch.randn(nq, d, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
for _ in range(2):
    out = idx.search_device(Q, k)
torch.cuda.synchronize()
_lib.lib().convdr_prof_enable(1)
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(5):
    out = idx.search_device(Q, k)
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 5
em, band = (t.float().mean().item() for t in idx.last_counts(nq, k))
parts = {nm: _lib.prof_collect(nm) for nm in ("ip_scan_emit", "ip_scan_sample", "ip_rescore", "ip_cut", "ip_select")}
print("n=%d nq=%d: %.2f ms = %.0f G pairs/s; uncertified %d; emitted %.0f band %.0f; %s" % (
    n, nq, ms, nq * n / ms / 1e6, int((out[2] != 0).sum()), em, band,
    {k_: round(v[0] / max(v[1], 1), 3) for k_, v in parts.items()}), flush=True)
D, I = idx.search(Q.cpu().numpy()[:50], k)
print("host API ok", D.shape, I.shape, idx.stats)
