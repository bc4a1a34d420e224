"""How far ahead of the GPU does the host run in the KD training loop?  Host time to ENQUEUE n steps (no synchronisation) vs the
time until the GPU has finished them."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import bench
from types import SimpleNamespace
from convdr_amd import train as TR
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
TR.reserve_streams(dev)
student = bench.random_rdot_model(0).to(dev); teacher = bench.random_rdot_model(0).to(dev).eval()
student.config.hidden_dropout_prob = student.config.attention_probs_dropout_prob = 0.1
TR.flatten_parameters(student)
targs = SimpleNamespace(learning_rate=1e-5, adam_epsilon=1e-8, max_grad_norm=1.0, ranking_task=False, no_mse=False, num_negatives=9, gradient_accumulation_steps=1)
opt = TR.get_optimizer(targs, student, weight_decay=0.0); sched = TR.get_linear_schedule_with_warmup(opt, 0, 10000)
g = torch.Generator(device=dev).manual_seed(0)
def turns(L, lo):
    i = torch.randint(3, 50000, (64, L), generator=g, device=dev); i[:, 0] = 0
    ln = torch.randint(lo, L + 1, (64,), generator=g, device=dev)
    m = (torch.arange(L, device=dev)[None, :] < ln[:, None]).long()
    return i * m, m, ln.cpu().numpy().astype(np.int32)
(ci, cm, cl), (ti, tm, tl) = turns(256, 32), turns(64, 8)
batch = (ci, cm, ti, tm, cl, tl)
for _ in range(8): TR.train_step(targs, student, teacher, opt, sched, batch)
torch.cuda.synchronize()
for n in (1, 5, 20, 40):
    t0 = time.perf_counter()
    for _ in range(n): TR.train_step(targs, student, teacher, opt, sched, batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n = %2d steps: host enqueue %.3f ms/step, GPU done after %.3f ms/step" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
