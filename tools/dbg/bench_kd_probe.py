"""Where inside a full bench.py run does the KD training step become slower?  Runs bench.main() with train_kd_measure probes
spliced in after the timed loop, after each extras leg (source patched at import time; experiment support)."""
import os, re, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
src = open(os.path.join(ROOT, "bench.py")).read()
probe = '''
def _kd_probe(tag, dev):
    import torch
    d = train_kd_measure(dev, 0, 1, False, 8, 3, 64, with_kernels=False, dropout=0.1)
    print("KDPROBE %s: %.3f ms" % (tag, d["ms_per_step"]), file=sys.stderr, flush=True)
'''
src = src.replace("def flop_per_passage(L):", probe + "\ndef flop_per_passage(L):", 1)
src = src.replace("    L_.convdr_set_option(b\"clock_probe\", 0)\n", "    L_.convdr_set_option(b\"clock_probe\", 0)\n    _kd_probe('after timed loop', dev)\n", 1)
src = src.replace("        # ---- (1b) a-11 the way the reference runs it", "        _kd_probe('after block_load leg', dev)\n        # ---- (1b) a-11 the way the reference runs it", 1)
src = src.replace("    out.update(extras_search(dev, index, tower, head, building, filled_rows, nq, k, d))\n    return out", "    _kd_probe('after search_one_by_one_files leg', dev)\n    out.update(extras_search(dev, index, tower, head, building, filled_rows, nq, k, d))\n    _kd_probe('after extras_search', dev)\n    return out", 1)
src = src.replace("    roof[\"power\"] = _power_state()", "    _kd_probe('before rocm-smi', dev)\n    roof[\"power\"] = _power_state()\n    _kd_probe('after rocm-smi', dev)", 1)
sys.argv = ["bench.py", "--no-cpu-baseline"]
exec(compile(src, os.path.join(ROOT, "bench.py"), "exec"), {"__name__": "__main__", "__file__": os.path.join(ROOT, "bench.py")})
