"""Compare gradients of the flat-arena path and the per-parameter path after one backward (debug aid)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from tests.test_train_gpu import _tiny, _batch
from convdr_amd import train as TR
rs = np.random.RandomState(5)
ids, mask = _batch(rs, 6, 48, [48, 20, 33, 5, 40, 12])
G = torch.randn(6, 768, device="cuda")
for rep in range(4):
    grads = []
    for flat in (False, True):
        m = _tiny(seed=3).cuda().train()
        if flat:
            TR.flatten_parameters(m)
        for it in range(2):
            m.zero_grad(set_to_none=False) if it else None
            e = m(ids.cuda(), mask.cuda())
            (e * G).sum().backward()
        torch.cuda.synchronize()
        grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    bad = [n for n in grads[0] if n in grads[1] and not torch.equal(grads[0][n], grads[1][n]) and "_embeddings" not in n]
    print("rep", rep, "differing non-embedding grads:", bad)
    for n in bad[:3]:
        d = (grads[0][n] - grads[1][n]).abs()
        print("    ", n, "max abs diff %.3e" % d.max().item(), "of", grads[0][n].abs().max().item(), "count", int((d > 0).sum()), "/", d.numel())
