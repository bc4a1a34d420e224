"""Hunt for the ~1-in-30 slow stand-alone run of the KD step (11.1 instead of 9.8 ms): N child processes, each prints its step
time, the calibration scores of train._aux_streams and the SAME fork / join miniature re-timed on the chosen streams after
the training steps -- does a run that is slow show streams that have become 'bad' since the calibration?"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch, bench
    from convdr_amd import train as TR
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    aux = TR.reserve_streams(dev)
    d = bench.train_kd_measure(dev, 0, 1, False, 20, 5, 64, with_kernels=False, dropout=0.1)
    main = torch.cuda.current_stream(dev)
    big = torch.zeros(128 << 20, dtype=torch.float32, device=dev); small = torch.zeros(16 << 20, dtype=torch.float32, device=dev)
    post = [round(TR._fork_join_time(main, s, big, small)) for s in aux]
    print("%.3f ms/step | calibration %s | chosen streams re-timed after the steps (A, B, C): %s" %
          (d["ms_per_step"], TR._SIDE_STREAMS.get((("cuda", 0), "scores")), post), flush=True)
else:
    for i in range(int(os.environ.get("N", "30"))):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], stderr=subprocess.DEVNULL)
