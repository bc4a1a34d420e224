"""ip_scan_emit kernel time of ONE uncertified first pass (FlatIPIndex.search_device) over 1M x 768, 1k queries: the harness of
tools/dbg/ceiling.sh for the scan's timing-only knobs (with CONVDR_DBG_SCAN_NOEMIT the product search would refuse the result)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from convdr_amd import _lib  # noqa: E402
from convdr_amd.search import FlatIPIndex  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
idx = FlatIPIndex(768, device=dev)
idx.add(torch.randn(1_000_000, 768, device=dev, generator=g))
Q = torch.randn(1000, 768, device=dev, generator=g)
for _ in range(2):
    idx.search_device(Q, 100)
torch.cuda.synchronize()
_lib.lib().convdr_prof_enable(1)
for _ in range(6):
    idx.search_device(Q, 100)
torch.cuda.synchronize()
ms, cnt = _lib.prof_collect("ip_scan_emit")
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("CONVDR_DBG"))
print("[scan %s] ip_scan_emit %.4f ms" % (tag, ms / cnt))
