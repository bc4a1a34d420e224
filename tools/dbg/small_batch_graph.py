"""Does a HIP-graph replay shorten the launch-bound small-batch encode?  (one roberta-base encode, eager vs graph replay)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
dev = torch.device("cuda", 0)
model = bench.random_rdot_model(0).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(0)
for B, L in ((4, 64), (16, 128), (64, 64)):
    ids = torch.randint(3, 50000, (B, L), generator=g, device=dev); ids[:, 0] = 0
    lens = torch.randint(max(8, L // 4), L + 1, (B,), generator=g, device=dev)
    mask = (torch.arange(L, device=dev)[None, :] < lens[:, None]).long()
    ids = ids * mask
    hl = lens.cpu().numpy().astype(np.int32)
    with torch.no_grad():
        for _ in range(5):
            ref = model(ids, mask, seq_lens=hl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            model(ids, mask, seq_lens=hl)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 100 * 1e3
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    model(ids, mask, seq_lens=hl)
            torch.cuda.current_stream().wait_stream(s)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                out = model(ids, mask, seq_lens=hl)
            torch.cuda.synchronize()
            for _ in range(5):
                gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                gr.replay()
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) / 100 * 1e3
            same = torch.equal(out, ref)
            print("B=%3d L=%3d  eager %.3f ms  graph replay %.3f ms  identical %s" % (B, L, eager, graph, same), flush=True)
        except Exception as e:
            print("B=%3d L=%3d  eager %.3f ms  graph capture failed: %r" % (B, L, eager, e), flush=True)
