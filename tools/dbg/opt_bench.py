"""bench.py with library options taken from the environment (A/B runs): CONVDR_OPT_<NAME>=<int> -> convdr_set_option("<name>", int)
before anything runs.  Example: CONVDR_OPT_ATTN_BWD_FUSED=0 python tools/dbg/opt_bench.py --workload train_kd"""
import os, runpy, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from convdr_amd import _lib
for k, v in sorted(os.environ.items()):
    if k.startswith("CONVDR_OPT_"):
        _lib.check(_lib.lib().convdr_set_option(k[len("CONVDR_OPT_"):].lower().encode(), int(v)), k)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
