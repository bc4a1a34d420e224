"""Search time over a resident 1M x 768 block for several query-batch sizes (HBM-bound below ~300 queries)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from convdr_amd import _lib
from convdr_amd.search import FlatIPIndex
n, d, k = 1_000_000, 768, 100
dev = torch.device("cuda")
P = torch.randn(n, d, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
idx = FlatIPIndex(d, device=dev); idx.add(P); del P
for nq in [int(x) for x in os.environ.get('NQS', '1,16,64,100,128,250,500,1000,2000').split(',')]:
    Q = torch.randn(nq, d, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    for _ in range(3):
        out = idx.search_device(Q, k)
    torch.cuda.synchronize()
    _lib.lib().convdr_prof_enable(1)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        out = idx.search_device(Q, k)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 10
    scan = _lib.prof_collect("ip_scan_emit")
    resc = _lib.prof_collect("ip_rescore")
    other = {nm: _lib.prof_collect(nm) for nm in ("ip_scan_sample", "ip_cut", "ip_select")}
    bad = int((out[2] != 0).sum())
    print("nq=%5d  search %.3f ms  scan %.3f ms (%.2f TB/s bf16 stream, %.0f TFLOP/s)  rescore %.3f  uncertified %d" % (
        nq, ms, scan[0] / max(scan[1], 1), n * d * 2 / (scan[0] / max(scan[1], 1)) / 1e9, 2.0 * nq * n * d / (scan[0] / max(scan[1], 1)) / 1e9,
        resc[0] / max(resc[1], 1), bad) + "  " + " ".join("%s %.3f" % (nm.replace("ip_", ""), v[0] / max(v[1], 1)) for nm, v in other.items()), flush=True)
