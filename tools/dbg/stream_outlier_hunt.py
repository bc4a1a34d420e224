"""Hunt for the ~1-in-30 slow stand-alone run of the KD step (11.1 instead of 9.8 ms): N child processes, each prints its step
time, the calibration scores of train._StreamSets and every decision its step watchdog took (round 5: the probe of both
stream sets on the real step and any later move).  CONVDR_STREAM_SELFCHECK=0 in the environment gives the round-4 behaviour
(calibration only) for comparison."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch, bench
    from convdr_amd import train as TR
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    TR.reserve_streams(dev)
    d = bench.train_kd_measure(dev, 0, 1, False, 20, 16, 64, with_kernels=False, dropout=0.1)
    info = TR.stream_decisions(dev)
    print("%.3f ms/step | calibration %s | queue groups %s | active set %d | %s" %
          (d["ms_per_step"], info["scores_us"], info.get("queue_groups"), info["active_set"], " || ".join(info["decisions"])), flush=True)
else:
    for i in range(int(os.environ.get("N", "30"))):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], stderr=subprocess.DEVNULL)
