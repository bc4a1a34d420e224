"""Is the encode power-capped?  Polls rocm-smi (power, sclk, mclk, temperature, perf level / power cap) every ~100 ms while a child
process runs (a) the bench's encode loop (MFMA-bound), (b) an AdamW-sized streaming loop (HBM-bound), (c) nothing (idle); `kd`: the configs[2] training step."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
mode = sys.argv[1]
dev = torch.device("cuda", 0)
if mode == "encode":
    import bench
    m = bench.random_rdot_model(0).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(0)
    ids = torch.randint(3, 50000, (2048, 128), generator=g, device=dev); ids[:, 0] = 0
    mask = torch.ones_like(ids)
    t0 = time.time()
    with torch.no_grad():
        while time.time() - t0 < 12:
            for _ in range(10):
                m.body_emb(ids, mask)
            torch.cuda.synchronize()
elif mode == "kd":
    import bench
    bench.train_kd_measure(dev, 0, 1, False, 900, 8, 64, with_kernels=False)
elif mode == "stream":
    a = torch.zeros(1 << 28, device=dev); b = torch.ones(1 << 28, device=dev)
    t0 = time.time()
    while time.time() - t0 < 8:
        for _ in range(50):
            a.add_(b, alpha=0.5)
        torch.cuda.synchronize()
else:
    torch.zeros(1, device=dev); time.sleep(4)
''' % ROOT


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--showmaxpower", "--json"], capture_output=True,
                             text=True, timeout=10).stdout
        d = json.loads(out)
        c = d.get("card0", {})
        keep = {}
        for k, v in c.items():
            kl = k.lower()
            if "power" in kl or "sclk" in kl or "mclk" in kl or "fclk" in kl or "junction" in kl or "edge" in kl:
                keep[k] = v
        return keep
    except Exception as e:
        return {"error": repr(e)}


for mode in (sys.argv[1:] or ["idle", "encode", "stream"]):
    p = subprocess.Popen([sys.executable, "-c", CHILD, mode], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = []
    t0 = time.time()
    while p.poll() is None:
        rows.append((round(time.time() - t0, 2), smi()))
        time.sleep(0.1)
    print("== %s: %d samples" % (mode, len(rows)))
    for t, r in rows[:: max(1, len(rows) // 14)]:
        print("  t=%5.2f %s" % (t, json.dumps(r)))
