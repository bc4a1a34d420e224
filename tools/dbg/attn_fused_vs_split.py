"""One-workgroup attention backward (attn_bwd_fused = 1) against the dQ + dK/dV kernel pair (= 0) on the same forward: per-parameter
1 - cos of the q / k / v gradients, for ragged batches around the tile edges.  Debug tool (CONVDR_HIP_LIB selects the library)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from convdr_amd import _lib  # noqa: E402
from tests.test_train_gpu import _tiny_long as _tiny, _batch  # noqa: E402

L = _lib.lib()
FUSED = int(os.environ.get("FUSED", "1"))     # 1 = the one-workgroup kernel (2 was the key-pass prototype of round 6: tools/proto/attn_bwd_keypass.hpp)
DROP = float(os.environ.get("DROPOUT", "0"))
for lens in ([130, 64, 65], [256, 255, 129, 128, 127, 1, 33], [64], [65], [128], [129], [192], [193], [200, 100]):
    rs = np.random.RandomState(1)
    model = _tiny(seed=0, layers=2).cuda().train()
    model.config.hidden_dropout_prob = model.config.attention_probs_dropout_prob = DROP
    model.dropout_seed = 77
    ids, mask = _batch(rs, len(lens), max(lens), lens)
    ids, mask = ids.cuda(), mask.cuda()
    G = torch.from_numpy(rs.randn(len(lens), 768).astype(np.float32)).cuda()
    out = {}
    for fused in (0, FUSED):
        _lib.check(L.convdr_set_option(b"attn_bwd_fused", fused), "opt")
        model.zero_grad()
        model.__dict__["_dropout_calls"] = 0
        (model(ids, mask) * G).sum().backward()
        out[fused] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    _lib.check(L.convdr_set_option(b"attn_bwd_fused", 1), "opt")
    worst = []
    assert all(torch.isfinite(v).all() for v in out[FUSED].values())
    for n in out[0]:
        if any(k in n for k in ("query", "key", "value")) and "weight" in n:
            a, b = out[0][n].double().flatten(), out[FUSED][n].double().flatten()
            c = float(a @ b / (a.norm() * b.norm() + 1e-300))
            worst.append((1 - c, n))
    worst.sort(reverse=True)
    print("lens %-40s worst 1-cos %.2e (%s)   2nd %.2e (%s)" % (lens, worst[0][0], worst[0][1][-40:], worst[1][0], worst[1][1][-40:]))
