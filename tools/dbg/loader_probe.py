"""Host side of the block loader, without the GPU: NUMA layout, and the rate at which reader threads move a 3 GB file out of the
page cache into a pinned buffer (os.preadv), by thread count and by the NUMA node the threads are confined to."""
import glob
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    print(n.split("/")[-2], open(n).read().strip())
for dev in glob.glob("/sys/bus/pci/devices/*"):
    try:
        if open(dev + "/vendor").read().strip() == "0x1002" and open(dev + "/class").read().strip()[:4] in ("0x03", "0x12"):
            print("gpu", dev.split("/")[-1], "numa_node", open(dev + "/numa_node").read().strip(), "class", open(dev + "/class").read().strip())
    except OSError:
        pass
print("affinity:", len(os.sched_getaffinity(0)), "cpus")
nbytes = 3 << 30
td = tempfile.mkdtemp(prefix="probe_")
path = os.path.join(td, "blk")
a = np.random.RandomState(0).randint(0, 255, size=nbytes // 8, dtype=np.int64)
a.tofile(path)
del a
fd = os.open(path, os.O_RDONLY)
pin = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
mv = memoryview(pin.numpy())


def rd(a, b):
    pos = a
    while pos < b:
        got = os.preadv(fd, [mv[pos:b]], pos)
        pos += got


def run(threads, cpus=None, piece=4 << 20):
    def init():
        if cpus is not None:
            os.sched_setaffinity(0, cpus)
    pool = ThreadPoolExecutor(max_workers=threads, initializer=init)
    best = 0
    for rep in range(3):
        t0 = time.perf_counter()
        list(pool.map(lambda s: rd(s, min(nbytes, s + piece)), range(0, nbytes, piece)))
        best = max(best, nbytes / (time.perf_counter() - t0) / 1e9)
    pool.shutdown()
    return best


nodes = {}
for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    cpus = set()
    for part in open(n).read().strip().split(","):
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    nodes[n.split("/")[-2]] = cpus & os.sched_getaffinity(0)
for t in (8, 16, 32, 64, 128):
    print("pread -> pinned, %3d threads, any cpu: %.1f GB/s" % (t, run(t)), flush=True)
for name, cpus in nodes.items():
    if cpus:
        print("pread -> pinned, 32 threads on %s: %.1f GB/s" % (name, run(32, cpus)), flush=True)
for piece in (1 << 20, 16 << 20, 64 << 20):
    print("pread -> pinned, 32 threads, pieces of %d MB: %.1f GB/s" % (piece >> 20, run(32, None, piece)), flush=True)
# memcpy out of an mmap (page cache) for comparison
mm = np.memmap(path, dtype=np.uint8, mode="r")
dst = pin.numpy()
pool = ThreadPoolExecutor(max_workers=32)
for rep in range(2):
    t0 = time.perf_counter()
    list(pool.map(lambda s: np.copyto(dst[s:s + (4 << 20)], mm[s:s + (4 << 20)]), range(0, nbytes, 4 << 20)))
    print("np.copyto mmap -> pinned, 32 threads: %.1f GB/s" % (nbytes / (time.perf_counter() - t0) / 1e9), flush=True)
d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    d.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
print("pinned -> HBM: %.1f GB/s" % (3 * nbytes / (time.perf_counter() - t0) / 1e9))
os.close(fd)
os.remove(path)
os.rmdir(td)
