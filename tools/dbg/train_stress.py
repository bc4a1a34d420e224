"""Workspace-reuse stress for the training path: forward/backward of many different ragged shapes interleaved on one model
(with and without dropout), each compared bit for bit (embedding tables aside: fp32 atomics) with the same shape run on
its own right after.  A stale activation, a workspace handed out twice or a side-stream overrun shows up as a difference."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.test_train_gpu import _tiny, _batch  # noqa: E402


def grads_of(m, ids, mask, G, seed):
    m.zero_grad(set_to_none=True)
    m.dropout_seed = seed
    m.__dict__["_dropout_calls"] = 0          # same mask every time this shape runs
    e = m(ids, mask)
    (e * G).sum().backward()
    return e.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def main(rounds=40):
    rs = np.random.RandomState(0)
    m = _tiny(seed=3, layers=2)
    m.config = getattr(m, "config", None)
    m = m.cuda().train()
    shapes = []
    for i in range(8):
        B = int(rs.choice([1, 3, 6, 17]))
        L = int(rs.choice([8, 48, 130]))
        lens = [int(rs.randint(1, L + 1)) for _ in range(B)]
        ids, mask = _batch(rs, B, L, lens)
        shapes.append((ids.cuda(), mask.cuda(), torch.randn(B, 768, device="cuda")))
    bad = 0
    for p_drop in (0.0, 0.1):
        for mod in m.modules():
            if hasattr(mod, "config") and hasattr(mod.config, "hidden_dropout_prob"):
                mod.config.hidden_dropout_prob = p_drop
                mod.config.attention_probs_dropout_prob = p_drop
        ref = [grads_of(m, *s, seed=100 + i) for i, s in enumerate(shapes)]
        for r in range(rounds):
            order = rs.permutation(len(shapes))
            # several forwards queued before their backwards, in shuffled order
            for i in order:
                e, g = grads_of(m, *shapes[i], seed=100 + i)
                diff = [n for n in g if "embeddings.word" not in n and "embeddings.position" not in n and "token_type" not in n
                        and not torch.equal(g[n], ref[i][1][n])]
                if not torch.equal(e, ref[i][0]) or diff:
                    bad += 1
                    print("p=%.1f round %d shape %d: emb equal %s, differing grads %s" % (p_drop, r, i, torch.equal(e, ref[i][0]), diff[:4]), flush=True)
    print("differences:", bad)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
