"""Phase timing inside the FFN1 GEMM (k_gemm<EPI_GELU_BF16>): s_memtime stamps of wave 0 of every workgroup at the
phase boundaries of each tile; prints median shader cycles per phase.  Needs a trace build
(make -C convdr_amd/csrc clean all TRACE=1).  Experiment tool, not part of the product."""
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
names = ["mainloop", "prefetch+bias+barrier", "pass0 compute", "pass0 barrier", "pass0 dma-wait", "pass0 store issue",
         "pass1 compute", "pass1 barrier", "pass1 dma-wait", "pass1 store issue"]
for i, n in enumerate(names):
    d = t[:, :, i + 1] - t[:, :, i]
    print("%-24s median %8.0f cycles   p10 %8.0f  p90 %8.0f" % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
tile = t[:, 1:, 0] - t[:, :-1, 0]
print("tile period              median %8.0f cycles" % np.median(tile))

# K step 6 of tile 8, per wave (rows 56..59 of the workgroup's stamp block): [wave][0..5]
raw = buf.cpu().numpy().reshape(512, 64, 16).astype(np.float64)[:256, 56:60].reshape(256, 8, 8)
ok = raw[:, :, 5] > 0
print("per-wave K step (cycles, median over workgroups): wave | dma-wait | barrier | dma-issue | reads+mfma issue | step period")
for wv in range(8):
    r = raw[ok[:, wv], wv]
    print("  wave %d | %6.0f | %6.0f | %6.0f | %6.0f | %6.0f" % (wv, np.median(r[:, 1] - r[:, 0]), np.median(r[:, 2] - r[:, 1]),
          np.median(r[:, 3] - r[:, 2]), np.median(r[:, 4] - r[:, 3]), np.median(r[:, 5] - r[:, 0])))
