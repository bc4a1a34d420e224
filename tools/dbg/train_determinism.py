"""Run the same forward+backward several times and report which gradients differ between runs (debug aid)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from tests.test_train_gpu import _tiny, _batch
rs = np.random.RandomState(5)
ids, mask = _batch(rs, 6, 48, [48, 20, 33, 5, 40, 12])
m = _tiny(seed=3).cuda().train()
G = torch.randn(6, 768, device="cuda")
runs = []
for r in range(6):
    m.zero_grad(set_to_none=True)
    e = m(ids.cuda(), mask.cuda())
    (e * G).sum().backward()
    torch.cuda.synchronize()
    runs.append((e.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
for r in range(1, len(runs)):
    bad = [n for n in runs[0][1] if not torch.equal(runs[0][1][n], runs[r][1][n])]
    print("run", r, "emb equal", torch.equal(runs[0][0], runs[r][0]), "differing grads:", bad[:8], len(bad))
    for n in bad[:3]:
        d = (runs[0][1][n] - runs[r][1][n]).abs()
        print("    ", n, "max abs diff %.3e" % d.max().item(), "of", runs[0][1][n].abs().max().item(), "count", int((d > 0).sum()))
