"""KD step time as a function of how many other streams were first-used in the process before the training step's own streams
(HIP assigns hardware queues to streams in order of first use, GPU_MAX_HW_QUEUES = 4 by default)."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import torch, bench
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
    d = bench.train_kd_measure(dev, 0, 1, False, 20, 5, 64, with_kernels=True, dropout=0.1)
    kk = d["kernels"]
    from convdr_amd import train as TR
    aux = TR._aux_streams(dev)
    print("streams used before: %d -> %.3f ms/step | " % (k, d["ms_per_step"]) + " ".join("%s %.2f" % (n.replace("gemm_", ""), kk[n]["ms_per_step"]) for n in kk)
          + " | aux scores %s" % TR._SIDE_STREAMS.get((("cuda", 0), "scores")), flush=True)
else:
    for k in [int(x) for x in os.environ.get('KS', '0,4,5,6,7,5,6,0').split(',')]:
        subprocess.run([sys.executable, os.path.abspath(__file__), str(k)], stderr=subprocess.DEVNULL)
