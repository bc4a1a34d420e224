"""The configs[2] KD step with the socket power and shader clock polled beside it (rocm-smi from a thread, every ~50 ms):
python tools/dbg/kd_power.py [steps = 400].  CONVDR_HIP_LIB selects the library.  Debug tool."""
import json
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
# CONVDR_OPTS="name=value,name=value": convdr_set_option calls before the run (e.g. attn_bwd_fused=0)
from convdr_amd import _lib  # noqa: E402
for kv in filter(None, os.environ.get("CONVDR_OPTS", "").split(",")):
    k, v = kv.split("=")
    _lib.check(_lib.lib().convdr_set_option(k.encode(), int(v)), "convdr_set_option")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
samples, stop = [], threading.Event()


def poll():
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10)
            c = json.loads(r.stdout).get("card0", {})
            w = [float(v) for k, v in c.items() if "power" in k.lower() and "max" not in k.lower()]
            f = [int(re.sub(r"\D", "", v)) for k, v in c.items() if k.lower().startswith("sclk clock speed")]
            m = [int(re.sub(r"\D", "", v)) for k, v in c.items() if k.lower().startswith("fclk clock speed") or k.lower().startswith("mclk clock speed")]
            if w and f:
                samples.append((time.perf_counter(), w[0], f[0], m))
        except Exception:
            return
        stop.wait(0.03)


d0 = bench.train_kd_measure(dev, 0, 1, False, 20, 5, 64, with_kernels=False)       # settle + warm
th = threading.Thread(target=poll, daemon=True)
t0 = time.perf_counter()
th.start()
d = bench.train_kd_measure(dev, 0, 1, False, steps, 5, 64, with_kernels=False)
stop.set()
th.join(timeout=12)
late = [s for s in samples if s[0] - t0 > 0.5 * (time.perf_counter() - t0)]
ws, fs = sorted(s[1] for s in late), sorted(s[2] for s in late)
print("[%s %s] step %.3f ms (first short run %.3f) | socket W median %.0f max %.0f | sclk median %d MHz (min %d max %d) | other clocks %s | %d samples" % (
    os.path.basename(os.environ.get("CONVDR_HIP_LIB", "libconvdr_hip.so")), os.environ.get("CONVDR_OPTS", ""), d["ms_per_step"], d0["ms_per_step"],
    ws[len(ws) // 2] if ws else -1, ws[-1] if ws else -1, fs[len(fs) // 2] if fs else -1, fs[0] if fs else -1, fs[-1] if fs else -1,
    late[len(late) // 2][3] if late else None, len(late)))
