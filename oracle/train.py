"""Oracle: the KD + ranking training step (CPU, torch fp32 + autograd).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates /root/reference/drivers/run_convdr_train.py:101-193 (step body), :69-74 (optimizer + schedule) and
/root/reference/utils/dpr_utils.py:80-87 (get_optimizer).  The optimizer arithmetic is third-party
(transformers==2.3.0 ``AdamW``, not vendored, removed from current transformers): restated from its published
algorithm -- m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps);
then p -= lr wd p -- which differs from torch.optim.AdamW in where eps enters.  Parity for it is therefore
"unpinned at the source"; the fixture tests/golden/train_step.npz pins the reference's USE of it.
"""
import math

import torch
import torch.nn.functional as F

from . import encoder as OE


def hf_adamw_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, correct_bias=True):
    b1, b2 = betas
    m.mul_(b1).add_(g, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    denom = v.sqrt().add_(eps)
    step_size = lr
    if correct_bias:
        step_size = lr * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if weight_decay > 0.0:
        p.add_(p, alpha=-lr * weight_decay)


def linear_schedule(step, warmup, total):
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


def clip_coef(total_norm, max_norm):
    return min(1.0, max_norm / (total_norm + 1e-6))


def kd_losses(sd_student, sd_teacher, batch, *, num_layers, num_heads, docs=None, num_negatives=9, emulate_bf16=False):
    """loss1 = MSE(student(concat), teacher(target)); loss2 = CE(<student, teacher(docs)>, 0) when docs given.
    emulate_bf16: both towers through oracle/encoder.py's bf16-emulating forward."""
    concat_ids, concat_mask, target_ids, target_mask = batch
    embs = OE.rdot_nll_emb(sd_student, concat_ids, concat_mask, num_layers=num_layers, num_heads=num_heads, emulate_bf16=emulate_bf16)
    with torch.no_grad():
        t = OE.rdot_nll_emb(sd_teacher, target_ids, target_mask, num_layers=num_layers, num_heads=num_heads, emulate_bf16=emulate_bf16)
    loss1 = F.mse_loss(embs, t)
    loss2 = None
    if docs is not None:
        with torch.no_grad():
            d = OE.rdot_nll_emb(sd_teacher, docs[0], docs[1], num_layers=num_layers, num_heads=num_heads, emulate_bf16=emulate_bf16)
        d = d.view(embs.shape[0], num_negatives + 1, -1)
        logits = (embs.unsqueeze(1) * d).sum(-1)
        loss2 = F.cross_entropy(logits, torch.zeros(embs.shape[0], dtype=torch.long))
    return embs, loss1, loss2


def inbatch_rank_loss(embs, docs_all, pos):
    """In-batch-negative ranking loss -- NOT in the reference ("parity unpinned by the
    reference"); this function is its definition for convdr_inbatch_ce_fwd_bwd.  It generalises
    run_convdr_train.py:160-170 (logits = <q, own docs>, CrossEntropy against index 0) to logits over every document
    gathered from all ranks, target = row of the query's own positive.  embs [B, E], docs_all [N, E], pos int64 [B]."""
    logits = embs @ docs_all.t()
    return F.cross_entropy(logits, pos.long())
