"""Does a synthetic fork / join workload predict which auxiliary stream serialises with the main stream in the real step?
For k other streams used before: pick the aux streams like train._aux_streams, time the proxy on (main, A) and (main, B), then the
real step."""
import os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import torch, bench
    from convdr_amd import train as TR
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    k = int(sys.argv[1])
    keep = []
    x = torch.zeros(16, device=dev)
    for i in range(k):
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            x.add_(1)
        keep.append(s)
    torch.cuda.synchronize()
    A, B = TR._aux_streams(dev)
    main = torch.cuda.current_stream(dev)
    small = torch.zeros(16 << 20, device=dev)      # 64 MB: ~30 us per pass
    big = torch.zeros(128 << 20, device=dev)       # 512 MB: ~250 us per pass

    def proxy(side):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(3):
            for blk in range(4):
                ev = torch.cuda.Event(); ev.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    big.mul_(1.0)
                for i in range(5):
                    small.mul_(1.0)
            fin = torch.cuda.Event(); fin.record(side); main.wait_event(fin)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3 * 1e6
    for _ in range(2):
        pa, pb = proxy(A), proxy(B)
    d = bench.train_kd_measure(dev, 0, 1, False, 20, 5, 64, with_kernels=True, dropout=0.1)
    kk = d["kernels"]
    print("k=%d proxy A %.0f us  B %.0f us | step %.3f ms  fwd(qkv) %.2f  dgrad %.2f" % (k, pa, pb, d["ms_per_step"], kk["gemm_qkv"]["ms_per_step"], kk["gemm_dgrad"]["ms_per_step"]), flush=True)
else:
    for k in (0, 4, 5, 6, 7, 5, 6):
        subprocess.run([sys.executable, os.path.abspath(__file__), str(k)], stderr=subprocess.DEVNULL)
