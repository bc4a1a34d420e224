"""Randomised forward / backward cases against autograd on the oracle (GPU box): hidden sizes, layer counts, ragged
batches incl. length-1 and max-length rows, with and without mean pooling.  Prints worst cosine per case."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import encoder as OE  # noqa: E402
from tests.helpers import cosine  # noqa: E402


def run(c):
    from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
    rs = np.random.RandomState(c)
    hidden = int(rs.choice([128, 256, 768]))
    heads = hidden // 64
    layers = int(rs.choice([1, 2, 3]))
    inter = int(rs.choice([256, 512, 3072, 192, 320, 704]))     # (192 / 320 / 704: ragged last feature tile of the blocked gelu' image)
    L = int(rs.choice([1, 8, 40, 130, 256]))
    B = int(rs.choice([1, 2, 5, 9]))
    lens = [int(rs.randint(1, L + 1)) for _ in range(B)]
    if rs.rand() < 0.5:
        lens[0] = L
    torch.manual_seed(c)
    cfg = RobertaConfig(vocab_size=300, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                        intermediate_size=inter, max_position_embeddings=300, hidden_dropout_prob=0.0,
                        attention_probs_dropout_prob=0.0)
    model = MSMarcoConfigDict["rdot_nll"].model_class(cfg)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.05)
    ids = rs.randint(3, 300, size=(B, L)).astype(np.int64)
    ids[:, 0] = 0
    mask = np.zeros((B, L), np.int64)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
        ids[b, n:] = 1
    ids, mask = torch.from_numpy(ids), torch.from_numpy(mask)
    G = torch.from_numpy(rs.randn(B, 768).astype(np.float32))
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    ref_emb = OE.rdot_nll_emb(sd, ids, mask, num_layers=layers, num_heads=heads)
    (ref_emb * G).sum().backward()
    ref = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    model = model.cuda().train()
    emb = model(ids.cuda(), mask.cuda())
    fc = 1 - cosine(emb.detach().cpu().numpy(), ref_emb.detach().numpy()).min()
    (emb * G.cuda()).sum().backward()
    worst, worst_n = 0.0, ""
    for n, p in model.named_parameters():
        if n in ref and not n.endswith("attention.self.key.bias"):
            g, r = p.grad.detach().cpu().double().reshape(-1), ref[n].double().reshape(-1)
            if r.norm() < 1e-12:
                continue
            cc = 1 - float((g @ r) / (g.norm() * r.norm() + 1e-300))
            if cc > worst:
                worst, worst_n = cc, n
    ok = fc < 1e-3 and worst < 5e-3
    print("[%s] case %d: H=%d layers=%d I=%d B=%d L=%d lens=%s  fwd 1-cos %.1e  worst grad 1-cos %.1e (%s)" % (
        "ok" if ok else "FAIL", c, hidden, layers, inter, B, L, lens, fc, worst, worst_n), flush=True)
    return ok


if __name__ == "__main__":
    n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    bad = 0
    for c in range(n0, n0 + (int(sys.argv[2]) if len(sys.argv) > 2 else 30)):
        try:
            bad += not run(c)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("[FAIL] case %d: %s: %s" % (c, type(e).__name__, str(e)[:300]), flush=True)
    print("failures:", bad)
