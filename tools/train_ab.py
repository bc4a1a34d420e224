"""A/B of convdr_set_option settings on the configs[2] KD training step, interleaved inside ONE process (one box, one clock
state): python tools/train_ab.py [--reps 3] [--steps 20] name:opt=v,opt=v name2:...   (a bare name = default options).
Prints the step time and the per-span kernel times of every run, then the per-configuration medians."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from convdr_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("configs", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = _lib.lib()
    cfgs, seen = [], set()
    for c in a.configs:
        name, _, opts = c.partition(":")
        kv = [(k, int(v)) for k, v in (o.split("=") for o in opts.split(",") if "=" in o)]
        cfgs.append((name, kv))
        seen.update(k for k, _ in kv)
    res = {n: [] for n, _ in cfgs}
    for rep in range(a.reps):
        for name, kv in cfgs:
            for k in seen:
                _lib.check(L.convdr_set_option(k.encode(), 0), "convdr_set_option")
            for k, v in kv:
                _lib.check(L.convdr_set_option(k.encode(), v), "convdr_set_option")
            d = bench.train_kd_measure(dev, 0, 1, False, a.steps, 5, 64, dropout=a.dropout)
            kern = d["kernels"]
            res[name].append(d["ms_per_step"])
            assert d["final_loss"] == d["final_loss"], "NaN loss: this configuration computes garbage (and NaN operands run faster)"
            print("[%s] step %.3f ms  %.0f samples/s  loss %.5f | " % (name, d["ms_per_step"], d["value"], d["final_loss"]) +
                  " ".join("%s %.2f" % (n.replace("gemm_", ""), kern[n]["ms_per_step"]) for n in kern), flush=True)
    for name, _ in cfgs:
        print("median[%s] %.3f ms  (min %.3f)" % (name, statistics.median(res[name]), min(res[name])))


if __name__ == "__main__":
    main()
