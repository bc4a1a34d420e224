"""bench.py with torch.cuda.empty_cache() turned into a no-op: does the in-bench KD step (12.0 ms vs 10.5 ms stand-alone) come
from the allocator re-growing its pools after every leg?"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
if os.environ.get("NO_EMPTY_CACHE"):
    torch.cuda.empty_cache = lambda: None
sys.argv = ["bench.py", "--no-cpu-baseline"]
src = open(os.path.join(ROOT, "bench.py")).read()
exec(compile(src, os.path.join(ROOT, "bench.py"), "exec"), {"__name__": "__main__", "__file__": os.path.join(ROOT, "bench.py")})
