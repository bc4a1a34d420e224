"""Phase timing inside the FFN1 GEMM (k_gemm<EPI_GELU_BF16>): s_memtime stamps of wave 0 of every workgroup at the
phase boundaries of each tile; prints mean microseconds per phase.  Experiment tool, not part of the product."""
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
    buf = torch.zeros(512 * 64 * 16, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib().convdr_set_option(b"gemm_trace", buf.data_ptr()), "set_option")
    tower.embed(ids, None, head=head, seq_lens=lens)
    torch.cuda.synchronize()
    _lib.lib().convdr_set_option(b"gemm_trace", 0)
t = buf.cpu().numpy().reshape(512, 64, 16).astype(np.float64)[:256]
ntile = int((t[0, :, 0] > 0).sum())
print("tiles per workgroup:", ntile)
t = t[:, :ntile]
tick = 1e-2  # s_memtime ticks at 100 MHz on gfx9
names = ["mainloop", "prefetch+bias+barrier", "pass0 compute", "pass0 barrier", "pass0 dma-wait", "pass0 store issue",
         "pass1 compute", "pass1 barrier", "pass1 dma-wait", "pass1 store issue"]
ph = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10)]
for n, (a, b) in zip(names, ph):
    d = (t[:, :, b] - t[:, :, a]) * tick
    print("%-24s mean %7.2f us   p10 %7.2f  p90 %7.2f" % (n, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
tile = (t[:, 1:, 0] - t[:, :-1, 0]) * tick
print("tile period               mean %7.2f us" % tile.mean())
print("first tile start spread   %.2f us" % ((t[:, 0, 0].max() - t[:, 0, 0].min()) * tick))
