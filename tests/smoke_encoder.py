"""One tiny encoder forward + one tiny training step on cuda:0 checked against the CPU oracle (used by
__graft_entry__.smoke; test infrastructure, not product code)."""
import numpy as np
import torch


def run():
    from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
    from oracle import encoder as OE
    torch.manual_seed(0)
    cfg = RobertaConfig(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                        intermediate_size=256, max_position_embeddings=66)
    model = MSMarcoConfigDict["rdot_nll"].model_class(cfg)
    rs = np.random.RandomState(0)
    ids = rs.randint(3, 300, size=(5, 48)).astype(np.int64)
    ids[:, 0] = 0
    mask = np.zeros_like(ids)
    for b, n in enumerate([48, 17, 33, 1, 40]):
        mask[b, :n] = 1
        ids[b, n:] = 0
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ref = OE.rdot_nll_emb(sd, torch.from_numpy(ids), torch.from_numpy(mask), num_layers=2, num_heads=2).numpy()
    model = model.cuda().eval()
    with torch.no_grad():
        emb = model(torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()).cpu().numpy()
    cos = (emb * ref).sum(1) / np.sqrt((emb * emb).sum(1) * (ref * ref).sum(1))
    assert cos.min() > 1 - 1e-3, "encoder embeddings differ from the oracle: cosine %s" % cos

    # one tiny KD step (forward + backward + clip + AdamW) through the training kernels: the loss matches the oracle's
    from types import SimpleNamespace
    from convdr_amd import train as TR
    from oracle import train as OT
    torch.manual_seed(1)
    cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = 0.0
    student = MSMarcoConfigDict["rdot_nll"].model_class(cfg)
    sd_s = {k: v.detach().clone() for k, v in student.state_dict().items()}
    t_ids, t_mask = ids[:, :16].copy(), mask[:, :16].copy()
    batch_cpu = tuple(torch.from_numpy(x) for x in (ids, mask, t_ids, t_mask))
    _, l1_ref, _ = OT.kd_losses(sd_s, sd, batch_cpu, num_layers=2, num_heads=2)
    args = SimpleNamespace(learning_rate=1e-4, adam_epsilon=1e-8, max_grad_norm=1.0, ranking_task=False, no_mse=False,
                           num_negatives=9, gradient_accumulation_steps=1)
    student = student.cuda()
    opt = TR.get_optimizer(args, student)
    sched = TR.get_linear_schedule_with_warmup(opt, 0, 10)
    w0 = student.embeddingHead.weight.detach().clone()
    loss, l1, _ = TR.train_step(args, student, model, opt, sched, tuple(t.cuda() for t in batch_cpu))
    assert abs(l1.item() - l1_ref.item()) < 1e-2 * l1_ref.item() + 1e-4, (l1.item(), l1_ref.item())
    assert not torch.equal(student.embeddingHead.weight, w0)
