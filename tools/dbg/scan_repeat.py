import os, sys
sys.path.insert(0, os.getcwd())
import torch
from convdr_amd.search import FlatIPIndex
dev = torch.device("cuda")
for (n, nq, k) in ((1_000_000, 1000, 100), (300_000, 257, 100), (47_104, 1000, 100)):
    g = torch.Generator(device=dev).manual_seed(n)
    idx = FlatIPIndex(768, device=dev)
    idx.add(torch.randn(n, 768, device=dev, generator=g))
    Q = torch.randn(nq, 768, device=dev, generator=g)
    D0, I0 = idx.search_tensors(Q, k)
    bad = 0
    for r in range(40):
        D, I = idx.search_tensors(Q, k)
        if not (torch.equal(D, D0) and torch.equal(I, I0)):
            bad += 1
    print("n=%d nq=%d: %d of 40 repeats differ; stats %s" % (n, nq, bad, idx.stats), flush=True)
