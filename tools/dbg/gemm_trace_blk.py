"""Phase anatomy of the FFN1 tile in the product configuration (k_gemm<EPI_GELU_BLK>, R3 K step): s_memtime stamps of
thread 0 of every workgroup: 0 tile start, 1 main loop done, 2 barrier passed (next tile's prologue issued before it),
6 last store issued.  Needs the trace build: make -C convdr_amd/csrc TRACE=1; run with
CONVDR_HIP_LIB=convdr_amd/libconvdr_hip_trace.so CONVDR_TRACE_EPI=8."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import bench
from convdr_amd import _lib
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
t = t[:, :ntile]
print("tiles per workgroup (last launch traced = last layer's FFN1):", ntile)
for name, a, b in (("main loop", 0, 1), ("bias park + next prologue issue + barrier", 1, 2), ("epilogue (GELU, pack, stores)", 2, 6)):
    d = t[:, :, b] - t[:, :, a]
    print("%-44s median %8.0f  p10 %8.0f  p90 %8.0f" % (name, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
per = t[:, 1:, 0] - t[:, :-1, 0]
print("%-44s median %8.0f" % ("tile period", np.median(per)))
print("%-44s median %8.0f" % ("store issue -> next tile start", np.median(t[:, 1:, 0] - t[:, :-1, 6])))

# K step 6 of tile 8, per wave: stamps 0 top, 1 DMA landed (counted vmcnt), 2 barrier passed, 3 first fragments requested + L chunk
# issued, 4 all MFMAs issued, 5 R chunk issued, 6 top of step 7
raw = buf.cpu().numpy().reshape(512, 64, 16).astype(np.float64)[:256, 56:60].reshape(256, 8, 8)
ok = raw[:, :, 6] > 0
print("K step per wave (median cycles): wave | vmcnt wait | barrier | frags + L issue | MFMA issue | R issue | to next top | step")
for wv in range(8):
    r = raw[ok[:, wv], wv]
    if len(r) == 0: continue
    print("  wave %d | %6.0f | %6.0f | %6.0f | %6.0f | %6.0f | %6.0f | %6.0f" % (
        wv, *(np.median(r[:, i + 1] - r[:, i]) for i in range(6)), np.median(r[:, 6] - r[:, 0])))
