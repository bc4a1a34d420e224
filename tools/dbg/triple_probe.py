import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import encoder as OE
from tests.test_train_gpu import _tiny, _batch
rs = np.random.RandomState(12)
model = _tiny()
q = _batch(rs, 4, 24, [24, 9, 17, 3])
a = _batch(rs, 4, 40, [40, 33, 12, 25])
b = _batch(rs, 4, 40, [22, 40, 31, 8])
def ref_grads(emu):
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    e = [OE.rdot_nll_emb(sd, i, m, num_layers=2, num_heads=2, emulate_bf16=emu) for i, m in (q, a, b)]
    for t in e: t.retain_grad()
    loss = OE.pairwise_nll(*e); loss.backward()
    return loss.item(), {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}, [t.grad for t in e], [t.detach() for t in e]
l_e, g_e, de_e, e_e = ref_grads(True)
l_f, g_f, de_f, e_f = ref_grads(False)
m = model.cuda().train()
embs = [m.query_emb(q[0].cuda(), q[1].cuda()), m.body_emb(a[0].cuda(), a[1].cuda()), m.body_emb(b[0].cuda(), b[1].cuda())]
for t in embs: t.retain_grad()
from convdr_amd.model.models import _pairwise_nll
loss = _pairwise_nll(*embs); loss.backward()
print("loss hip %.6f emu %.6f fp32 %.6f" % (loss.item(), l_e, l_f))
for i in range(3):
    print("emb", i, "max|hip-emu| %.2e" % (embs[i].detach().cpu() - e_e[i]).abs().max().item(), " d_emb rel diff vs emu %.2e" % ((embs[i].grad.cpu() - de_e[i]).norm() / de_e[i].norm()).item())
rows = []
for n, p in m.named_parameters():
    if n in g_e and p.grad is not None and g_e[n].norm() > 1e-8 and not n.endswith("key.bias"):
        x, r = p.grad.cpu().double().reshape(-1), g_e[n].double().reshape(-1)
        rows.append((1 - float(x @ r / (x.norm() * r.norm())), abs(float(x.norm() / r.norm()) - 1), n))
for c, nd, n in sorted(rows, reverse=True)[:8]: print("%.2e %.2e %s" % (c, nd, n))
# ---- per-pass gradients with the SAME upstream gradients: is the residual of the sum a cancellation of per-pass rounding? ----
def hip_pass(ids_mask, d):
    m.zero_grad()
    e = m.body_emb(ids_mask[0].cuda(), ids_mask[1].cuda())
    (e * d.cuda()).sum().backward()
    return {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}
def emu_pass(ids_mask, d):
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
    e = OE.rdot_nll_emb(sd, ids_mask[0], ids_mask[1], num_layers=2, num_heads=2, emulate_bf16=True)
    (e * d).sum().backward()
    return {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
ha, hb = hip_pass(a, de_e[1]), hip_pass(b, de_e[2])
ea, eb = emu_pass(a, de_e[1]), emu_pass(b, de_e[2])
for n in ("roberta.encoder.layer.0.intermediate.dense.bias", "roberta.encoder.layer.1.attention.self.value.bias", "roberta.encoder.layer.0.intermediate.dense.weight", "roberta.embeddings.word_embeddings.weight"):
    cos = lambda x, r: 1 - float(x.double().reshape(-1) @ r.double().reshape(-1) / (x.double().norm() * r.double().norm()))
    print("%s\n   pass a 1-cos %.2e   pass b %.2e   sum %.2e   |g_a + g_b| / |g_a| = %.3f" % (n, cos(ha[n], ea[n]), cos(hb[n], eb[n]), cos(ha[n] + hb[n], ea[n] + eb[n]), float((ea[n] + eb[n]).norm() / ea[n].norm())))
