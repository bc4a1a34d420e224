"""Phase timing inside k_attention_bwd_fused at the configs[2] training shape (layer 5 of one student forward + backward):
s_memtime stamps of thread 0 (wave 0: owns keys 0..31) and thread 256 (wave 4: a dQ wave) of EVERY workgroup, with the
sequence length and HW_ID, so that the launch can be laid out per CU.  Needs the trace library (make -C convdr_amd/csrc TRACE=1;
CONVDR_HIP_LIB=convdr_amd/libconvdr_hip_trace.so).  Experiment tool, not part of the product."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from convdr_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
B, Ls = 64, 256
drop = float(os.environ.get("DROPOUT", "0.1"))
student = bench.random_rdot_model(0).to(dev).train()
student.config.hidden_dropout_prob = student.config.attention_probs_dropout_prob = drop
g = torch.Generator(device=dev).manual_seed(0)
ids = torch.randint(3, 50000, (B, Ls), generator=g, device=dev)
ids[:, 0] = 0
lens = torch.randint(32, Ls + 1, (B,), generator=g, device=dev)
mask = (torch.arange(Ls, device=dev)[None, :] < lens[:, None]).long()
ids = ids * mask
lens_h = lens.cpu().numpy().astype(np.int32)
G = torch.randn(B, 768, device=dev)
L = _lib.lib()
for _ in range(3):
    student.zero_grad()
    (student(ids, mask, seq_lens=lens_h) * G).sum().backward()
torch.cuda.synchronize()
nwg = 12 * B
buf = torch.zeros(nwg * 32, dtype=torch.int64, device=dev)
_lib.check(L.convdr_set_option(b"attn_trace", buf.data_ptr()), "set_option")
student.zero_grad()
(student(ids, mask, seq_lens=lens_h) * G).sum().backward()
torch.cuda.synchronize()
L.convdr_set_option(b"attn_trace", 0)
t = buf.cpu().numpy().reshape(nwg, 2, 16).astype(np.int64)
w0, w4 = t[:, 0, :], t[:, 1, :]
ln = w0[:, 14]
hw = w0[:, 15]
start = w0[:, 0].min()
print("launch span (first start .. last end): %d cycles" % (w0[:, 13].max() - start))
cu_key = (hw >> 8) & 0xfff           # cu_id [11:8], sh_id [12], se_id [15:13]  (+ xcc in other register: approximate key)
print("workgroups %d, distinct (se, sh, cu) keys %d" % (nwg, len(set(cu_key.tolist()))))
for lo, hi_ in ((1, 64), (65, 128), (129, 192), (193, 256)):
    sel = (ln >= lo) & (ln <= hi_)
    if not sel.any():
        continue
    a = w0[sel]
    b = w4[sel]
    nst = int(np.ceil(hi_ / 64))
    print("--- len %3d..%3d: %4d workgroups, whole workgroup median %6.0f cycles (p10 %6.0f, p90 %6.0f)" % (
        lo, hi_, sel.sum(), np.median(a[:, 13] - a[:, 0]), np.percentile(a[:, 13] - a[:, 0], 10), np.percentile(a[:, 13] - a[:, 0], 90)))
    print("    prologue issue (loads, D, dS zero)            %6.0f" % np.median(a[:, 1] - a[:, 0]))
    print("    first wait + barrier (tiles landed)           %6.0f" % np.median(a[:, 2] - a[:, 1]))
    for it in range(nst):
        full = a[:, 2 + 2 * it] > 0
        nxt = a[:, 4 + 2 * it] if it + 1 < nst else a[:, 10]
        nxt = np.where(a[:, 4 + 2 * it] > 0, a[:, 4 + 2 * it], a[:, 10]) if it + 1 < 4 else a[:, 10]
        if full.any():
            print("    step %d: wave 0 stage+body %6.0f | to next barrier passed %6.0f   || wave 4: dq phase %6.0f, body %6.0f" % (
                it, np.median((a[:, 10] if it + 1 == nst else a[:, 3 + 2 * it])[full] - a[:, 2 + 2 * it][full]) if it + 1 < nst else np.median(a[full, 10] - a[full, 2 + 2 * it]),
                np.median(nxt[full] - a[:, 2 + 2 * it][full]),
                np.median(b[full, 3 + 2 * it] - b[full, 2 + 2 * it]), np.median((np.where(b[:, 4 + 2 * it] > 0, b[:, 4 + 2 * it], b[:, 10]) if it + 1 < 4 else b[:, 10])[full] - b[full, 3 + 2 * it])))
    print("    last barrier                                  %6.0f" % np.median(a[:, 11] - a[:, 10]))
    print("    dK / dV park + stores (wave 0)                %6.0f" % np.median(a[:, 12] - a[:, 11]))
    print("    final dq phase after its own stores (wave 4)  %6.0f" % np.median(b[:, 13] - b[:, 12]))
# per-CU layout: the workgroups of the busiest and of a median CU
order = np.argsort(w0[:, 0])
by = {}
for i in order:
    by.setdefault(int(hw[i]) & 0xfffff00, []).append(i)
spans = sorted(by.items(), key=lambda kv: max(w0[j, 13] for j in kv[1]))
for name, (k, wl) in (("last CU to finish", spans[-1]), ("median CU", spans[len(spans) // 2])):
    print("%s (hw %x): " % (name, k) + "  ".join("[len %d: %d..%d]" % (ln[j], w0[j, 0] - start, w0[j, 13] - start) for j in wl))
