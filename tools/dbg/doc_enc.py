import sys, time; sys.path.insert(0, '/root/repo')
import torch, bench
m = bench.random_rdot_model(0).cuda().eval()
g = torch.Generator(device="cuda").manual_seed(0)
ids = torch.randint(3, 50000, (640, 512), generator=g, device="cuda"); ids[:, 0] = 0
mask = torch.ones_like(ids)
for chunk in (8, 64, 640):
    with torch.no_grad():
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            outs = [m(ids[i:i + chunk], mask[i:i + chunk], is_query=False) for i in range(0, 640, chunk)]
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("chunk", chunk, "%.1f ms" % (dt * 1e3), "TF/s %.0f" % (640 * 96.64e9 / dt / 1e12))
