"""Phase timing inside k_gemm_resid_ln (attention-output projection launches): s_memtime stamps of thread 0 of every
workgroup; prints median shader cycles per phase.  Needs a trace build (make -C convdr_amd/csrc clean all TRACE=1).
Experiment tool, not part of the product."""
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
    buf = torch.zeros(4096 * 16 + 256 * 64, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib().convdr_set_option(b"gemm_trace_ln", buf.data_ptr()), "set_option")
    tower.embed(ids, None, head=head, seq_lens=lens)
    torch.cuda.synchronize()
    _lib.lib().convdr_set_option(b"gemm_trace_ln", 0)
raw = buf.cpu().numpy().astype(np.float64)
t = raw[:4096 * 16].reshape(4096, 16)[:2048]
names = ["mainloop", "barrier + stage LN params + barrier", "bias + residual + row sums", "barrier",
         "mean reduce + barrier + centred squares", "var reduce + 2 barriers", "normalise + store"]
for i, n in enumerate(names[:7]):
    d = t[:, i + 1] - t[:, i]
    print("%-42s median %8.0f cycles   p10 %8.0f  p90 %8.0f" % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
for n, (i, j) in (("half 0: DMA issue", (2, 8)), ("half 0: DMA wait", (8, 9)), ("half 0: barrier", (9, 10)), ("half 0: consume", (10, 11)),
                  ("half 0: normalise + park", (6, 12)), ("half 0: barrier", (12, 13)), ("half 0: store issue", (13, 14))):
    d = t[:, j] - t[:, i]
    print("%-42s median %8.0f cycles   p10 %8.0f  p90 %8.0f" % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
d = t[:, 7] - t[:, 0]
print("%-42s median %8.0f cycles" % ("whole workgroup", np.median(d)))

st = raw[2048 * 16:2048 * 16 + 256 * 64].reshape(256, 8, 8)
print("per-wave K step (cycles, median over workgroups): wave | dma-wait | barrier | dma-issue | reads+mfma issue | step period")
for wv in range(8):
    r = st[st[:, wv, 5] > 0, wv]
    print("  wave %d | %6.0f | %6.0f | %6.0f | %6.0f | %6.0f" % (wv, np.median(r[:, 1] - r[:, 0]), np.median(r[:, 2] - r[:, 1]),
          np.median(r[:, 3] - r[:, 2]), np.median(r[:, 4] - r[:, 3]), np.median(r[:, 5] - r[:, 0])))
