"""Search time over a resident 1M x 768 block for several query-batch sizes (HBM-bound below ~300 queries), with the
finishing chain as ONE launch (k_ip_finish, round 5) and as three (k_ip_cut + k_ip_rescore + k_ip_select).  `search` is timed
with the per-kernel event spans OFF (they put markers between the launches); the breakdown comes from a second pass."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from convdr_amd import _lib
from convdr_amd.search import FlatIPIndex
n, d, k = 1_000_000, 768, 100
dev = torch.device("cuda")
P = torch.randn(n, d, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
idx = FlatIPIndex(d, device=dev); idx.add(P); del P
L = _lib.lib()


def timed(Q, reps=20):
    for _ in range(3):
        out = idx.search_device(Q, k)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        out = idx.search_device(Q, k)
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps, out


for nq in [int(x) for x in os.environ.get('NQS', '50,100,250,479,1000').split(',')]:
    Q = torch.randn(nq, d, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    for fused in (1, 0, 1, 0):
        _lib.check(L.convdr_set_option(b"ip_fused_finish", fused), "set_option")
        ms, out = timed(Q)
        L.convdr_prof_enable(1)
        timed(Q, reps=5)
        spans = {nm: _lib.prof_collect(nm) for nm in ("ip_scan_emit", "ip_scan_sample", "ip_finish", "ip_cut", "ip_rescore", "ip_select")}
        L.convdr_prof_enable(0)
        scan = spans["ip_scan_emit"][0] / max(spans["ip_scan_emit"][1], 1)
        bad = int((out[2] != 0).sum())
        print("nq=%5d  %s  search %.3f ms  scan %.3f ms (%.2f TB/s 16-bit stream, %.0f TFLOP/s)  uncertified %d  " % (
            nq, "one-launch finish" if fused else "three launches   ", ms, scan, n * d * 2 / scan / 1e9, 2.0 * nq * n * d / scan / 1e9, bad)
            + " ".join("%s %.3f" % (nm.replace("ip_", ""), v[0] / max(v[1], 1)) for nm, v in spans.items() if v[1] and nm != "ip_scan_emit"), flush=True)
L.convdr_set_option(b"ip_fused_finish", 1)
