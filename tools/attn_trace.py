"""Phase timing inside k_attention_fwd (workgroup (0, 0, b) of every sequence b): s_memtime stamps of thread 0.
Needs a trace build (make -C convdr_amd/csrc clean all TRACE=1).  Experiment tool, not part of the product."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from convdr_amd import _lib  # noqa: E402

B, L = 2048, 128
model = bench.random_rdot_model(0).cuda().eval()
ids = bench.synthetic_tokens(B, L, 0, "cuda")
lens = np.full(B, L, np.int32)
tower, head = model.roberta, (model.embeddingHead, model.norm)
with torch.no_grad():
    for _ in range(2):
        tower.embed(ids, None, head=head, seq_lens=lens)
    buf = torch.zeros(2048 * 8, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib().convdr_set_option(b"attn_trace", buf.data_ptr()), "set_option")
    tower.embed(ids, None, head=head, seq_lens=lens)
    torch.cuda.synchronize()
    _lib.lib().convdr_set_option(b"attn_trace", 0)
t = buf.cpu().numpy().reshape(2048, 8).astype(np.float64)
names = ["prologue scalars", "Q loads issued + both K/V tiles issued + wait + barrier", "tile 0 compute + wait + barrier",
         "tile 1 compute", "normalise + stores issued"]
for i, n in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print("%-58s median %7.0f cycles  p10 %7.0f  p90 %7.0f" % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
d = t[:, 5] - t[:, 0]
print("%-58s median %7.0f cycles" % ("whole workgroup", np.median(d)))
