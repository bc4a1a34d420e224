import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from types import SimpleNamespace
from tests.test_train_gpu import _tiny, _batch
from convdr_amd import train as TR
rs = np.random.RandomState(5)
ids, mask = _batch(rs, 6, 48, [48, 20, 33, 5, 40, 12])
tid, tmask = _batch(rs, 6, 16, [16, 9, 4, 16, 7, 3])
batch = tuple(x.cuda() for x in (ids, mask, tid, tmask))
args = SimpleNamespace(learning_rate=1e-3, adam_epsilon=1e-8, max_grad_norm=1.0, ranking_task=False, no_mse=False,
                       num_negatives=0, gradient_accumulation_steps=1)
for rep in range(5):
    res = []
    for flat in (False, True):
        student, teacher = _tiny(seed=3).cuda(), _tiny(seed=4).cuda().eval()
        if flat:
            TR.flatten_parameters(student)
        opt = TR.get_optimizer(args, student, weight_decay=0.0)
        sched = TR.get_linear_schedule_with_warmup(opt, 0, 10)
        w0 = {k: v.detach().clone() for k, v in student.state_dict().items()}
        loss = TR.train_step(args, student, teacher, opt, sched, batch)[0].item()
        g = {n: p.grad.detach().clone() for n, p in student.named_parameters() if p.grad is not None}
        res.append((loss, {k: v.detach().clone() for k, v in student.state_dict().items()}, g, w0))
    print("rep", rep, "loss", res[0][0], res[1][0])
    for k, v in res[0][1].items():
        if not v.dtype.is_floating_point or "_embeddings" in k: continue
        d = (v - res[1][1][k]).abs().max().item()
        if d > 0:
            gd = (res[0][2][k] - res[1][2][k]).abs().max().item() if k in res[0][2] else -1
            i = (v - res[1][1][k]).abs().argmax().item()
            print("   %-60s dW %.3e  dgrad %.3e  grad at worst %.3e  step %.3e/%.3e" % (k, d, gd, res[0][2][k].reshape(-1)[i].item() if k in res[0][2] else 0,
                  (v - res[0][3][k]).reshape(-1)[i].item(), (res[1][1][k] - res[1][3][k]).reshape(-1)[i].item()))
