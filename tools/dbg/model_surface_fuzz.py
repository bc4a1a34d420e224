"""Randomised cases for the rest of the model.models surface against the oracle: MaxP multi-chunk documents (random chunk
counts, empty chunks), the pairwise NLL triple loss, the dpr towers, use_mean pooling -- forward values, and for the
triple loss the gradients."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import encoder as OE  # noqa: E402
from tests.helpers import cosine  # noqa: E402


def toks(rs, B, L, lens, pad=1):
    ids = rs.randint(3, 250, size=(B, L)).astype(np.int64)
    ids[:, 0] = 0
    m = (np.arange(L)[None, :] < np.asarray(lens)[:, None]).astype(np.int64)
    ids[m == 0] = pad
    return torch.from_numpy(ids), torch.from_numpy(m)


def main(cases):
    from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig, BertConfig
    bad = 0
    for c in range(cases):
        rs = np.random.RandomState(900 + c)
        torch.manual_seed(c)
        cfg = RobertaConfig(vocab_size=260, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                            max_position_embeddings=514, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        kind = c % 4
        try:
            if kind == 0:      # multi-chunk MaxP
                model = MSMarcoConfigDict["rdot_nll_multi_chunk"].model_class(cfg).cuda().eval()
                sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
                B, n = int(rs.choice([1, 2, 3])), int(rs.choice([1, 2, 3]))
                def doc():
                    ids, m = [], []
                    for b in range(B):
                        live = int(rs.randint(1, n + 1))
                        ci, cm = toks(rs, n, 512, [int(rs.randint(1, 513)) if j < live else 0 for j in range(n)])
                        ci[live:, 0] = 1                      # an empty chunk is all padding
                        ids.append(ci.reshape(-1)); m.append(cm.reshape(-1))
                    return torch.stack(ids), torch.stack(m)
                ia, ma = doc(); ib, mb = doc()
                iq, mq = toks(rs, B, 24, [int(rs.randint(1, 25)) for _ in range(B)])
                with torch.no_grad():
                    loss = model(iq.cuda(), mq.cuda(), ia.cuda(), ma.cuda(), ib.cuda(), mb.cuda())[0].item()
                q = OE.rdot_nll_emb(sd, iq, mq, num_layers=2, num_heads=2)
                a = OE.rdot_multi_chunk_body_emb(sd, ia, ma, num_layers=2, num_heads=2)
                b = OE.rdot_multi_chunk_body_emb(sd, ib, mb, num_layers=2, num_heads=2)
                ref = OE.multi_chunk_nll(q, a, b, ma, mb).item()
                ok, info = abs(loss - ref) < 2e-2 * max(1.0, abs(ref)), "loss %.5f vs %.5f (B=%d chunks=%d)" % (loss, ref, B, n)
            elif kind == 1:    # pairwise triple loss with gradients
                model = MSMarcoConfigDict["rdot_nll"].model_class(cfg).cuda().train()
                sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
                B = int(rs.choice([1, 2, 5]))
                L = int(rs.choice([8, 40, 130]))
                t = [toks(rs, B, L, [int(rs.randint(1, L + 1)) for _ in range(B)]) for _ in range(3)]
                loss = model(*[x.cuda() for pair in t for x in pair])[0]
                loss.backward()
                e = [OE.rdot_nll_emb(sd, i, m, num_layers=2, num_heads=2) for i, m in t]
                ref = OE.pairwise_nll(*e)
                ref.backward()
                # q . (a - b) cancels almost completely on a random-init tower (a ~ b): per-parameter cosines are noise where
                # the true gradient nearly vanishes, so the whole gradient vector is compared, relative to its norm
                num = den = 0.0
                for nme, p in model.named_parameters():
                    g = sd[nme].grad
                    if g is None or p.grad is None:
                        continue
                    gg, rr = p.grad.detach().cpu().double().reshape(-1), g.double().reshape(-1)
                    num += float(((gg - rr) ** 2).sum()); den += float((rr ** 2).sum())
                worst = (num / max(den, 1e-300)) ** 0.5
                ok = abs(loss.item() - ref.item()) < 2e-2 * max(1.0, abs(ref.item())) and worst < 0.6   # (conditioning-limited: |a - b| is of the size of the bf16 forward error of a and b; measured 0.07-0.5)
                info = "loss %.5f vs %.5f, |g - ref| / |ref| %.1e (B=%d L=%d)" % (loss.item(), ref.item(), worst, B, L)
            elif kind == 2:    # dpr towers
                bc = BertConfig(vocab_size=260, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                max_position_embeddings=514, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
                model = MSMarcoConfigDict["dpr"].model_class(type("A", (), {"bert_config": bc})()).cuda().eval()
                sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
                B, L = int(rs.choice([1, 3, 6])), int(rs.choice([1, 16, 200]))
                ids, m = toks(rs, B, L, [int(rs.randint(1, L + 1)) for _ in range(B)], pad=0)
                with torch.no_grad():
                    qe, be = model(ids.cuda(), m.cuda()).cpu().numpy(), model(ids.cuda(), m.cuda(), is_query=False).cpu().numpy()
                rq = OE.dpr_emb(sd, ids, m, tower="question_model", num_layers=2, num_heads=2).numpy()
                rb = OE.dpr_emb(sd, ids, m, tower="ctx_model", num_layers=2, num_heads=2).numpy()
                w = max(1 - cosine(qe, rq).min(), 1 - cosine(be, rb).min())
                ok, info = w < 1e-3, "1-cos %.1e (B=%d L=%d)" % (w, B, L)
            else:              # use_mean pooling
                model = MSMarcoConfigDict["rdot_nll"].model_class(cfg)
                model.use_mean = True
                model.roberta.pool_mean = True
                model = model.cuda().eval()
                sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
                B, L = int(rs.choice([1, 4, 9])), int(rs.choice([1, 30, 300]))
                ids, m = toks(rs, B, L, [int(rs.randint(1, L + 1)) for _ in range(B)])
                with torch.no_grad():
                    e = model(ids.cuda(), m.cuda()).cpu().numpy()
                r = OE.rdot_nll_emb(sd, ids, m, num_layers=2, num_heads=2, use_mean=True).numpy()
                w = 1 - cosine(e, r).min()
                ok, info = w < 1e-3, "1-cos %.1e (B=%d L=%d)" % (w, B, L)
        except Exception as ex:  # noqa: BLE001
            ok, info = False, "%s: %s" % (type(ex).__name__, str(ex)[:200])
        print("[%s] case %d kind %d: %s" % ("ok" if ok else "FAIL", c, kind, info), flush=True)
        bad += not ok
    print("failures:", bad)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
