import sys, time; sys.path.insert(0, '.')
import torch
from convdr_amd import train as TR
B, N, E = 64, 5120, 768
e = torch.randn(B, E, device="cuda", requires_grad=True); d = torch.randn(N, E, device="cuda") * 0.3
pos = torch.randint(0, N, (B,), device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l = TR.ranking_loss_inbatch(e, d, pos); torch.cuda.synchronize()
    print("fwd+bwd kernel %.3f ms" % ((time.perf_counter() - t0) * 1e3))
