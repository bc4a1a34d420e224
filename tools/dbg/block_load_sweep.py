"""Block file -> HBM rate by staging-chunk size and host threads (FlatIPIndex.add of a blocks.BlockView)."""
import sys, os, time, tempfile, shutil
sys.path.insert(0, '.')
import numpy as np, torch
from convdr_amd import blocks
from convdr_amd.search import FlatIPIndex
n, d = 1_000_000, 768
td = tempfile.mkdtemp()
try:
    path = os.path.join(td, "b.pb")
    blocks.dump_block(path, torch.randn(n, d).numpy())
    for chunk, th, nb in ((64, 16, 2), (64, 16, 3), (64, 16, 4), (64, 32, 3), (32, 16, 4), (32, 32, 4), (128, 16, 3), (128, 32, 3), (64, 8, 4)):
        if True:
            rates = []
            for rep in range(3):
                with blocks.BlockView(path) as bv:
                    idx = FlatIPIndex(d)
                    idx.host_chunk_bytes, idx.host_copy_threads, idx.host_stage_buffers = chunk << 20, th, nb
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    idx.add(bv); torch.cuda.synchronize()
                    rates.append(bv.array.nbytes / (time.perf_counter() - t0) / 1e9)
                    del idx
            print("chunk %3d MB threads/chunk %2d buffers %d: %.1f GB/s (best of 3: %.1f)" % (chunk, th, nb, rates[-1], max(rates)), flush=True)
    # pure H2D ceiling from pinned memory: wherever the allocating thread happened to run, and on the GPU's NUMA node
    from convdr_amd.search import gpu_numa_cpus, pinned_near
    print("gpu numa cpus:", len(gpu_numa_cpus("cuda:0") or ()))
    dev = torch.empty((n // 4, d), device="cuda")
    for name, pin in (("default placement", torch.empty((n // 4, d)).pin_memory()), ("near the GPU", pinned_near("cuda:0", (n // 4, d), torch.float32))):
        dev.copy_(pin, non_blocking=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): dev.copy_(pin, non_blocking=True)
        torch.cuda.synchronize(); print("pinned H2D (%s): %.1f GB/s" % (name, 4 * pin.numel() * 4 / (time.perf_counter() - t0) / 1e9))
finally:
    shutil.rmtree(td, ignore_errors=True)
