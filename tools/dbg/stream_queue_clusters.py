"""Which hardware queue does each calibration stream of train._StreamSets share?  Per child process: pairwise test of the eight
candidate streams with torch.cuda._sleep (one spinning thread: no resource contention -- two streams on the same hardware queue
take twice as long), the clusters, the roles (A teacher / norms, B weight gradients, C all-reduce) of both sets by cluster, the
probe's per-k-token costs and the final step time.  Question: is the +3 % mode of a set a queue shared between two roles?"""
import os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch, bench
    from convdr_amd import train as TR
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    TR.reserve_streams(dev)
    ss = TR._stream_sets(dev)
    cands = list(ss._keep)
    main = torch.cuda.current_stream(dev)
    spin = 400_000           # ~200 us at ~2 GHz

    def pair_us(x, y):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            ev = torch.cuda.Event(); ev.record(main)
            t0 = time.perf_counter()
            for s in (x, y):
                with torch.cuda.stream(s):
                    s.wait_event(ev)
                    torch.cuda._sleep(spin)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e6)
        return best
    one = pair_us(cands[0], cands[0]) / 2.0
    n = len(cands)
    cluster = [-1] * n
    nc = 0
    for i in range(n):
        if cluster[i] >= 0:
            continue
        cluster[i] = nc
        for j in range(i + 1, n):
            if cluster[j] < 0 and pair_us(cands[i], cands[j]) > 1.6 * one:
                cluster[j] = nc
        nc += 1
    with torch.cuda.stream(cands[0]):
        pass
    main_shared = [i for i in range(n) if pair_us(main, cands[i]) > 1.6 * one]
    idx = {id(c): i for i, c in enumerate(cands)}
    roles = [[cluster[idx[id(s)]] for s in st] for st in ss.sets]
    d = bench.train_kd_measure(dev, 0, 1, False, 20, 16, 64, with_kernels=False, dropout=0.1)
    info = TR.stream_decisions(dev)
    print("%.3f ms/step | one spin %.0f us | clusters %s | shares main's queue: %s | set roles (A,B,C) by cluster: %s | active %d | %s" %
          (d["ms_per_step"], one, cluster, main_shared, roles, info["active_set"], " || ".join(x.replace("stream self-check: ", "") for x in info["decisions"])), flush=True)
else:
    for i in range(int(os.environ.get("N", "12"))):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], stderr=subprocess.DEVNULL)
