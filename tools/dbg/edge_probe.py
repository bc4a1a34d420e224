"""Ad-hoc probe of odd shapes against the oracle (GPU box).  Prints one line per case; anything that fails here becomes a
test.  Not part of the product."""
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import encoder as OE  # noqa: E402
from oracle import search as OS  # noqa: E402
from tests.helpers import cosine  # noqa: E402
from tests.golden.make_golden import synth_corpus  # noqa: E402


def case(name, fn):
    try:
        r = fn()
        print("[ok  ] %-52s %s" % (name, r if r is not None else ""), flush=True)
    except Exception as e:  # noqa: BLE001
        print("[FAIL] %-52s %s: %s" % (name, type(e).__name__, str(e)[:300]), flush=True)
        traceback.print_exc(limit=3)


def small_model(layers=2, heads=12, hidden=768, inter=3072, seed=0):
    from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
    torch.manual_seed(seed)
    cfg = RobertaConfig(num_hidden_layers=layers, num_attention_heads=heads, hidden_size=hidden, intermediate_size=inter)
    return MSMarcoConfigDict["rdot_nll"].model_class(cfg).cuda().eval(), cfg


def enc_case(model, cfg, lens, L):
    B = len(lens)
    rs = np.random.RandomState(B * 131 + L)
    ids = rs.randint(3, 50000, size=(B, L)).astype(np.int64)
    ids[:, 0] = 0
    mask = np.zeros((B, L), np.int64)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
        ids[b, n:] = 1
    with torch.no_grad():
        emb = model(torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()).cpu().numpy()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref = OE.rdot_nll_emb(sd, torch.from_numpy(ids), torch.from_numpy(mask), num_layers=cfg.num_hidden_layers,
                            num_heads=cfg.num_attention_heads).numpy()
    cs = cosine(emb, ref)
    assert cs.min() > 1 - 1e-3, cs
    return "1-cos max %.2e  max abs %.3g" % (1 - cs.min(), np.abs(emb - ref).max())


def search_case(n, nq, k, d, seed=3, **kw):
    from convdr_amd.search import FlatIPIndex
    P, Q = synth_corpus(seed, n, d), synth_corpus(seed + 1, nq, d)
    idx = FlatIPIndex(d, **kw)
    if n:
        idx.add(P)
    D, I = idx.search(Q, k)
    Dr, Ir = OS.flat_ip_search(Q, P, k)
    np.testing.assert_array_equal(I, Ir)
    np.testing.assert_array_equal(D, Dr)
    return "D %s" % (D.shape,)


def main():
    model, cfg = small_model()
    for lens, L in (([1], 1), ([1], 8), ([1, 1, 1], 4), ([2, 1, 9], 16), ([512], 512), ([512, 1, 3], 512), ([7] * 65, 8),
                    ([128] * 3 + [1], 128), ([33] * 300, 40), ([5], 511)):
        case("encoder lens=%s L=%d" % (str(lens)[:24], L), lambda lens=lens, L=L: enc_case(model, cfg, lens, L))
    m2, c2 = small_model(layers=1, heads=2, hidden=128, inter=512, seed=1)
    for lens, L in (([1], 1), ([16, 3], 16), ([100] * 40, 128)):
        case("encoder H=128 lens=%s L=%d" % (str(lens)[:24], L), lambda lens=lens, L=L: enc_case(m2, c2, lens, L))
    m3, c3 = small_model(layers=1, heads=16, hidden=1024, inter=4096, seed=2)
    for lens, L in (([16, 3], 16), ([100] * 40, 128)):
        case("encoder H=1024 lens=%s L=%d" % (str(lens)[:24], L), lambda lens=lens, L=L: enc_case(m3, c3, lens, L))
    for n, nq, k, d in ((1, 1, 1, 768), (1, 3, 100, 768), (63, 2, 100, 768), (64, 2, 64, 768), (65, 1, 65, 768),
                        (4096, 1, 1, 768), (4097, 1000, 100, 768), (100000, 1, 100, 768), (300, 3, 300, 768),
                        (20000, 9, 1000, 768), (20000, 9, 2048, 768), (5000, 4, 100, 32), (5000, 4, 100, 1024),
                        (5000, 4, 100, 96), (5000, 4, 100, 72), (5000, 257, 100, 768), (0, 4, 10, 768)):
        case("search n=%d nq=%d k=%d d=%d" % (n, nq, k, d), lambda a=(n, nq, k, d): search_case(*a))
    # train step, B = 1 and odd lengths
    from convdr_amd import train as T

    def train_case(B, Ls, Lt):
        from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
        torch.manual_seed(5)
        cfgt = RobertaConfig(num_hidden_layers=2)
        student = MSMarcoConfigDict["rdot_nll"].model_class(cfgt).cuda().train()
        teacher = MSMarcoConfigDict["rdot_nll"].model_class(cfgt).cuda().eval()
        args = type("A", (), dict(learning_rate=1e-5, adam_epsilon=1e-8, weight_decay=0.0, max_grad_norm=1.0,
                                  gradient_accumulation_steps=1, ranking_task=False, no_mse=False, n_gpu=1,
                                  warmup_steps=0, max_steps=10))()
        opt = T.get_optimizer(args, student, weight_decay=0.0)
        sch = T.get_linear_schedule_with_warmup(opt, 0, 10)
        rs = np.random.RandomState(B)
        def mk(L):
            ids = rs.randint(3, 50000, size=(B, L)).astype(np.int64); ids[:, 0] = 0
            lens = rs.randint(1, L + 1, size=B)
            m = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
            ids[m == 0] = 1
            return torch.from_numpy(ids).cuda(), torch.from_numpy(m).cuda()
        ci, cm = mk(Ls)
        ti, tm = mk(Lt)
        out = [T.train_step(args, student, teacher, opt, sch, (ci, cm, ti, tm), step=s) for s in range(2)]
        l = [float(o[0]) for o in out]
        assert all(np.isfinite(l)), l
        return "loss %s" % l
    for B, Ls, Lt in ((1, 1, 1), (1, 256, 64), (3, 17, 5), (64, 256, 64)):
        case("train_step B=%d Ls=%d Lt=%d" % (B, Ls, Lt), lambda a=(B, Ls, Lt): train_case(*a))


if __name__ == "__main__":
    main()
