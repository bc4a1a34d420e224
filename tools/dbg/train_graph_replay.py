"""How much of the configs[2] KD step is host / launch bound?  Capture ONE whole train_step (teacher forward on the side
stream, student forward, loss, backward with its weight-gradient side stream, clip, AdamW) in a hipGraph and replay it:
the replay has no Python, no ctypes and no per-kernel launch calls, only the device-side dependencies.  A measurement,
not a product path: the graph bakes the batch's row count, the learning rate and the Adam step number in."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from types import SimpleNamespace
from convdr_amd import train as TR

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dropout = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
student = bench.random_rdot_model(0).to(dev)
teacher = bench.random_rdot_model(0).to(dev).eval()
student.config.hidden_dropout_prob = student.config.attention_probs_dropout_prob = dropout
TR.flatten_parameters(student)
targs = SimpleNamespace(learning_rate=1e-5, adam_epsilon=1e-8, max_grad_norm=1.0, ranking_task=False, no_mse=False,
                        num_negatives=9, gradient_accumulation_steps=1)
opt = TR.get_optimizer(targs, student, weight_decay=0.0)
sched = TR.get_linear_schedule_with_warmup(opt, 0, 10_000)
g = torch.Generator(device=dev).manual_seed(0)


def turns(B, L, lo):
    ids = torch.randint(3, 50000, (B, L), generator=g, device=dev)
    ids[:, 0] = 0
    lens = torch.randint(lo, L + 1, (B,), generator=g, device=dev)
    mask = (torch.arange(L, device=dev)[None, :] < lens[:, None]).long()
    return ids * mask, mask, lens.cpu().numpy().astype(np.int32)


(ci, cm, cl), (ti, tm, tl) = turns(64, 256, 32), turns(64, 64, 8)
batch = (ci, cm, ti, tm, cl, tl)


def step():
    return TR.train_step(targs, student, teacher, opt, sched, batch)[0]


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    step()
out = {"eager_ms_per_step": timed(step, 20), "student_dropout": dropout}
try:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            loss = step()
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    out["graph_replay_ms_per_step"] = timed(graph.replay, 20)
    out["host_or_launch_bound_share"] = 1.0 - out["graph_replay_ms_per_step"] / out["eager_ms_per_step"]
    out["loss_after_replays"] = float(loss)
except Exception as e:      # capture is a measurement aid; say why it failed
    import traceback
    traceback.print_exc()
    out["graph_error"] = "%s: %s" % (type(e).__name__, str(e)[:400])
print(json.dumps(out))
