"""Per-kernel spans of the configs[2] KD step with and without dropout (bench.train_kd_measure, ProfScope spans)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for rep in range(2):
    for p in (0.1, 0.0):
        r = bench.train_kd_measure(dev, 0, 1, False, 30, 8, 64, with_kernels=True, dropout=p)
        k = r.get("kernels", {})
        print("dropout %.1f  step %.3f ms | " % (p, r["ms_per_step"]) + " ".join("%s %.2f" % (n.replace("gemm_", ""), v["ms_per_step"]) for n, v in sorted(k.items())), flush=True)
