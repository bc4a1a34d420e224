"""Solo time of convdr_pack_transposed for the 49 weight matrices of a roberta-base student (0.34 GB fp32 read, 0.17 GB bf16 written)."""
import ctypes as C
import numpy as np
import torch
from convdr_amd import _lib

L = _lib.lib()
H, I, NL = 768, 3072, 12
shapes = []
for _ in range(NL):
    shapes += [(3 * H, H), (H, H), (I, H), (H, I)]
shapes.append((H, H))
src_off, tot = [], 0
for n, k in shapes:
    src_off.append(tot)
    tot += n * k + 768 * 4
base = torch.randn(tot, device="cuda")
dst = np.concatenate([[0], np.cumsum([n * k for n, k in shapes])]).astype(np.int64)
out = torch.empty(int(dst[-1]), dtype=torch.bfloat16, device="cuda")
cnt = len(shapes)
args = (_lib.ptr(base), cnt, (C.c_int64 * cnt)(*src_off), (C.c_int32 * cnt)(*[n for n, _ in shapes]), (C.c_int32 * cnt)(*[k for _, k in shapes]),
        (C.c_int64 * cnt)(*dst[:-1].tolist()), _lib.ptr(out), _lib.stream_ptr())
for _ in range(3):
    _lib.check(L.convdr_pack_transposed(*args), "pack")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    e0.record()
    _lib.check(L.convdr_pack_transposed(*args), "pack")
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
byt = sum(n * k for n, k in shapes) * 6
print("convdr_pack_transposed, %d matrices, %.0f MB moved: min %.1f us  median %.1f us  = %.2f TB/s" % (cnt, byt / 1e6, min(ts), float(np.median(ts)), byt / np.median(ts) / 1e6))
