"""Race check at the configs[2] size: the same forward + backward of a roberta-base student (batch 64, <= 256 tokens, dropout
0.1, fixed mask seed) ten times; every gradient outside the embedding tables (fp32 atomics) must be bitwise equal between runs --
the weight-gradient stream carries the head's jobs and runs beside the whole chain, so an ordering bug shows up here."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import bench
from convdr_amd import train as TR
dev = torch.device("cuda", 0)
m = bench.random_rdot_model(0).to(dev).train()
m.config.hidden_dropout_prob = m.config.attention_probs_dropout_prob = 0.1
TR.flatten_parameters(m)
g = torch.Generator(device=dev).manual_seed(0)
B, L = 64, 256
ids = torch.randint(3, 50000, (B, L), generator=g, device=dev); ids[:, 0] = 0
lens = torch.randint(32, L + 1, (B,), generator=g, device=dev)
mask = (torch.arange(L, device=dev)[None, :] < lens[:, None]).long()
ids = ids * mask
hl = lens.cpu().numpy().astype(np.int32)
G = torch.randn(B, 768, device=dev, generator=g)
ref, bad_total = None, 0
for r in range(10):
    m.zero_grad(set_to_none=True)
    m.dropout_seed, m.__dict__["_dropout_calls"] = 77, 0
    e = m(ids, mask, seq_lens=hl)
    (e * G).sum().backward()
    torch.cuda.synchronize()
    cur = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    if ref is None:
        ref, e0 = cur, e.detach().clone()
        continue
    bad = [n for n in ref if not torch.equal(ref[n], cur[n]) and not ("embeddings." in n and "LayerNorm" not in n)]
    bad_total += len(bad)
    print("run", r, "emb equal", torch.equal(e0, e.detach()), "differing non-table grads:", bad[:6], len(bad), flush=True)
print("differences: %d" % bad_total)
