"""Per-step wall time of the KD training step right after the encode workload of bench.py (same process): how many steps does
the first training run of a process need before it reaches its steady state?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import bench
from types import SimpleNamespace
from convdr_amd import train as TR
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
model = bench.random_rdot_model().to(dev).eval()
tower, head = model.roberta, (model.embeddingHead, model.norm)
ids = bench.synthetic_tokens(2048, 128, 0, dev); lens = np.full(2048, 128, np.int32)
with torch.no_grad():
    for _ in range(int(os.environ.get("ENC_STEPS", "25"))):
        tower.embed(ids, None, head=head, seq_lens=lens)
torch.cuda.synchronize()
for rnd in range(2):
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
    ts = []
    for i in range(24):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        TR.train_step(targs, student, teacher, opt, sched, batch)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("round %d per-step ms (synced each step):" % rnd, " ".join("%.1f" % t for t in ts), flush=True)
    # unsynced steady state
    t0 = time.perf_counter()
    for i in range(20): TR.train_step(targs, student, teacher, opt, sched, batch)
    torch.cuda.synchronize(); print("round %d unsynced 20 steps: %.3f ms/step" % (rnd, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
    del student, teacher, opt
    torch.cuda.empty_cache()
