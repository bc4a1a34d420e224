"""Pin the CPU oracle against fixtures produced by running the reference itself
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import torch

from oracle import encoder as OE
from oracle import search as OS
from tests.helpers import assert_topk_equivalent, cosine
from tests.golden.make_golden import synth_corpus


def _sd(z):
    return {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")}


def test_rdot_nll_embeddings_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "encoder_rdot_nll.npz"))
    cfg = json.loads(str(z["config"]))
    sd = _sd(z)
    for case in ("L16", "L64", "L510"):
        ids, mask = torch.from_numpy(z[case + "/ids"]), torch.from_numpy(z[case + "/mask"])
        emb = OE.rdot_nll_emb(sd, ids, mask, num_layers=cfg["num_hidden_layers"],
                              num_heads=cfg["num_attention_heads"], eps=cfg["layer_norm_eps"]).numpy()
        np.testing.assert_allclose(emb, z[case + "/emb"], atol=2e-5, rtol=0)
        assert cosine(emb, z[case + "/emb"]).min() > 1 - 1e-6
    hs = OE.encoder_hidden(sd, "roberta.", torch.from_numpy(z["L16/ids"]), torch.from_numpy(z["L16/mask"]),
                           kind="roberta", num_layers=2, num_heads=2, eps=1e-5, return_all=True)
    ref = z["L16/hidden_states"]
    m = z["L16/mask"].astype(bool)
    for l in range(3):  # padded query rows are don't-care (mask constant differs: -1e4 vs dtype-min)
        np.testing.assert_allclose(hs[l].numpy()[m], ref[l][m], atol=2e-5, rtol=0)


def test_rdot_nll_triple_loss_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "encoder_rdot_nll.npz"))
    sd = _sd(z)
    e = lambda i, m: OE.rdot_nll_emb(sd, torch.from_numpy(z["triple/" + i]), torch.from_numpy(z["triple/" + m]),
                                     num_layers=2, num_heads=2)
    loss = OE.pairwise_nll(e("ids_q", "m_q"), e("ids_a", "m_a"), e("ids_b", "m_b"))
    assert abs(loss.item() - float(z["triple/loss"])) < 1e-4 * max(1, abs(float(z["triple/loss"])))


def test_dpr_embeddings_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "encoder_dpr.npz"))
    sd = _sd(z)
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    q = OE.dpr_emb(sd, ids, mask, tower="question_model", num_layers=2, num_heads=2)
    b = OE.dpr_emb(sd, ids, mask, tower="ctx_model", num_layers=2, num_heads=2)
    np.testing.assert_allclose(q.numpy(), z["q_emb"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(b.numpy(), z["b_emb"], atol=2e-5, rtol=0)
    loss = OE.pairwise_nll(q, b, torch.flip(b, [0]))
    assert abs(loss.item() - float(z["pair_loss"])) < 1e-4


def _blocks_a(z):
    return [(synth_corpus(int(s), int(n), 768), np.arange(int(n), dtype=np.int64) * 3 + r)
            for r, (n, s) in enumerate(zip(z["a/sizes"], z["a/seeds"]))]


def test_search_one_by_one_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "search.npz"))
    blocks = _blocks_a(z)
    Q = synth_corpus(int(z["a/qseed"]), 16, 768)
    topN = int(z["a/topN"])
    mD, mI = OS.search_one_by_one(blocks, Q, topN)
    assert mD.shape == z["a/merged_D"].shape == (16, 2 * topN) and mD.dtype == np.float64 and mI.dtype == np.int64
    assert_topk_equivalent(z["a/merged_D"], z["a/merged_I"], mD, mI, k=topN)
    mD, mI = OS.search_one_by_one(blocks[:1], Q, topN)
    assert mD.shape == z["b/merged_D"].shape == (16, topN)
    assert_topk_equivalent(z["b/merged_D"], z["b/merged_I"], mD, mI, k=topN)


def test_search_ties_match_reference(golden_dir):
    """Exact duplicates: lower index first inside a block, earlier block first across blocks."""
    z = np.load(os.path.join(golden_dir, "search.npz"))
    base = synth_corpus(21, 300, 768)
    b0 = np.concatenate([base[:200], base[50:60]])
    b1 = np.concatenate([base[40:70], base[200:300]])
    blocks = [(b0, np.arange(len(b0), dtype=np.int64)), (b1, 1000 + np.arange(len(b1), dtype=np.int64))]
    mD, mI = OS.search_one_by_one(blocks, base[45:53] + 0.0, 10)
    k = 10
    # canonical scores of identical vectors tie exactly -> documented rule decides:
    # lower index first inside a block, earlier block first across blocks (:218 `>=`).
    # (The fp32-BLAS stand-in behind the fixture does NOT tie exactly: its scores for
    # identical rows differ in the 7th digit, so vs the fixture they are exchangeable.)
    for j, r in enumerate(range(45, 53)):
        want = [r] + ([200 + r - 50] if r >= 50 else []) + [1000 + r - 40]
        assert mI[j, :len(want)].tolist() == want
        assert len(set(mD[j, :len(want)].tolist())) == 1
    assert_topk_equivalent(z["c/merged_D"], z["c/merged_I"], mD, mI, k=k)


def test_eval_dev_query_text_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "search.npz"))
    topN = int(z["a/topN"])
    rows = OS.eval_dev_query_rows([str(q) for q in z["d/qids"]], z["a/merged_D"], z["a/merged_I"], topN,
                                  z["d/offset2pid"].tolist())
    assert "".join(OS.trec_lines(rows, topN)) == str(z["d/trec"])
    first = json.loads(str(z["d/jsonl"]).splitlines()[0])
    assert first["doc_id"] == str(rows[str(z["d/qids"][0])][0][0]) and first["label"] == 2


def test_canonical_score_is_fp64_accurate():
    rs = np.random.RandomState(0)
    Q, P = rs.randn(3, 768).astype(np.float32), rs.randn(50, 768).astype(np.float32)
    S = OS.canonical_scores(Q, P)
    ref = Q.astype(np.float64) @ P.astype(np.float64).T
    np.testing.assert_allclose(S, ref, rtol=1e-13, atol=1e-11)


def test_multi_chunk_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "encoder_rdot_nll.npz"))
    sd = _sd(z)
    t = lambda k: torch.from_numpy(z["mc/" + k])
    a = OE.rdot_multi_chunk_body_emb(sd, t("ids_a"), t("m_a"), num_layers=2, num_heads=2)
    live = z["mc/m_a"].reshape(2, 2, 512)[:, :, 0].astype(bool)
    np.testing.assert_allclose(a.numpy()[live], z["mc/emb_a"][live], atol=2e-5, rtol=0)   # padding chunks: don't care
    b = OE.rdot_multi_chunk_body_emb(sd, t("ids_b"), t("m_b"), num_layers=2, num_heads=2)
    q = OE.rdot_nll_emb(sd, t("ids_q"), t("m_q"), num_layers=2, num_heads=2)
    loss = OE.multi_chunk_nll(q, a, b, t("m_a"), t("m_b"))
    assert abs(loss.item() - float(z["mc/loss"])) < 1e-4 * max(1.0, abs(float(z["mc/loss"])))


def test_evaluate_loop_matches_reference(golden_dir):
    """run_convdr_inference.py:116-154 run by the reference itself (make_golden.py:gen_evaluate) vs oracle/inference.py."""
    from oracle import inference as OI
    z = np.load(os.path.join(golden_dir, "evaluate.npz"))
    sd = _sd(np.load(os.path.join(golden_dir, "encoder_rdot_nll.npz")))
    hist = json.loads(str(z["hist"]))
    emb, emb2id, raw = OI.evaluate(sd, z["ids"], z["mask"], [str(q) for q in z["qids"]], hist, int(z["batch"]),
                                   num_layers=2, num_heads=2)
    assert emb.dtype == np.float32 and emb.shape == z["embedding"].shape
    np.testing.assert_allclose(emb, z["embedding"], atol=2e-5, rtol=0)
    assert emb2id == [str(q) for q in z["embedding2id"]]
    assert raw == json.loads(str(z["raw_sequences"]))


def test_product_eval_dev_query_writes_the_reference_text(golden_dir, tmp_path):
    """convdr_amd.search.EvalDevQuery (the PRODUCT result writer, host code: no GPU needed) against the .trec and .jsonl
    text the reference's EvalDevQuery wrote for the same inputs (run_convdr_inference.py:21-113): byte-identical, incl.
    duplicate-pid removal, the positive label and json float formatting of the scores."""
    from convdr_amd.search import EvalDevQuery
    z = np.load(os.path.join(golden_dir, "search.npz"))
    topN = int(z["a/topN"])
    qids = [str(q) for q in z["d/qids"]]
    offset2pid = z["d/offset2pid"].tolist()
    with open(tmp_path / "queries.raw.tsv", "w") as f:
        for q in qids:
            f.write("%s\tquery text %s\n" % (q, q))
    with open(tmp_path / "collection.tsv", "w") as f:
        for pid in sorted(set(offset2pid)):
            f.write("%d\tpassage %d body\n" % (pid, pid))
    raw = [["hist %s" % q, "cur %s" % q] for q in qids]
    pos = {str(z["d/pos_qid"]): {int(z["d/pos_pid"]): int(z["d/pos_label"])}}
    EvalDevQuery(qids, z["a/merged_D"], pos, z["a/merged_I"], topN, str(tmp_path / "o.jsonl"), str(tmp_path / "o.trec"),
                 offset2pid, str(tmp_path), "raw", raw_sequences=raw)
    assert open(tmp_path / "o.trec").read() == str(z["d/trec"])
    assert open(tmp_path / "o.jsonl").read() == str(z["d/jsonl"])


def test_use_mean_pooling_matches_reference(golden_dir):
    """models.py:32-41 with use_mean = True, run by the reference (make_golden.py:gen_use_mean) vs the oracle."""
    z = np.load(os.path.join(golden_dir, "use_mean.npz"))
    sd = _sd(np.load(os.path.join(golden_dir, "encoder_rdot_nll.npz")))
    emb = OE.rdot_nll_emb(sd, torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"]), num_layers=2, num_heads=2, use_mean=True)
    np.testing.assert_allclose(emb.numpy(), z["emb"], atol=2e-5, rtol=0)


def test_product_eval_dev_query_equals_oracle_on_random_rankings(tmp_path):
    """The vectorised product writer against the oracle's row-by-row restatement of run_convdr_inference.py:21-113 on
    random rankings: many offsets mapping to the same pid (first occurrence kept), lists that run out of distinct pids
    before topN, repeated query ids."""
    from convdr_amd.search import EvalDevQuery
    for seed in range(12):
        rs = np.random.RandomState(seed)
        nq, topN = int(rs.randint(1, 9)), int(rs.choice([1, 5, 20]))
        n_off = int(rs.randint(topN, 200))
        n_pid = int(rs.randint(1, n_off + 1))
        offset2pid = rs.randint(0, n_pid, size=n_off).tolist()
        width = int(rs.choice([topN, 2 * topN]))                     # merged lists keep 2 * topN entries after >= 2 blocks
        merged_I = np.stack([rs.permutation(n_off)[:width] if n_off >= width else rs.randint(0, n_off, size=width)
                             for _ in range(nq)]).astype(np.int64)
        merged_D = -np.sort(-rs.rand(nq, width) * 100, axis=1)
        qids = ["q%d" % (i if rs.rand() < 0.85 else 0) for i in range(nq)]
        with open(tmp_path / "queries.raw.tsv", "w") as f:
            for q in sorted(set(qids)):
                f.write("%s\ttext of %s\n" % (q, q))
        with open(tmp_path / "collection.tsv", "w") as f:
            for pid in range(n_pid):
                f.write("%d\tpassage %d\n" % (pid, pid))
        out_j, out_t = str(tmp_path / ("o%d.jsonl" % seed)), str(tmp_path / ("o%d.trec" % seed))
        EvalDevQuery(qids, merged_D, {}, merged_I, topN, out_j, out_t, offset2pid, str(tmp_path), "raw",
                     raw_sequences=[["h", "c"]] * nq)
        rows = OS.eval_dev_query_rows(qids, merged_D, merged_I, topN, offset2pid)
        assert open(out_t).read() == "".join(OS.trec_lines(rows, topN)), seed


import pytest


@pytest.mark.parametrize("fixture", ["train_step.npz", "train_step_b.npz"])
def test_oracle_training_step_replays_the_reference_run(golden_dir, fixture):
    """oracle/train.py (step body run_convdr_train.py:101-193, clip :188-189, HF AdamW utils/dpr_utils.py:80-87, linear
    schedule :69-74) against the four optimizer steps the reference's own train() ran: same batches and sampled
    documents -> the same losses, the same gradient norms, the same parameters after the last step."""
    from oracle import train as OT
    z = np.load(os.path.join(golden_dir, fixture))
    cfg, hp = json.loads(str(z["config"])), json.loads(str(z["hyper"]))
    NL, NH = cfg["num_hidden_layers"], cfg["num_attention_heads"]
    sd = {k[3:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("w0/")}
    sdt = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("wt/")} or {k: v.clone() for k, v in sd.items()}
    w1 = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w1/")}
    # the parameters the reference model trains: everything the hot path reaches (pooler / classifier get no gradient)
    names = [k for k, v in sd.items() if v.dtype.is_floating_point]
    no_decay = ["bias", "LayerNorm.weight"]                     # utils/dpr_utils.py:81-86 (name-based groups)
    m = {k: torch.zeros_like(sd[k]) for k in names}
    v = {k: torch.zeros_like(sd[k]) for k in names}
    K1 = hp["num_negatives"] + 1
    dptr = 0
    for step, idxs in enumerate(z["batches"]):
        g = lambda key: torch.from_numpy(np.stack([z["ex/%d/%s" % (i, key)] for i in idxs]))
        n_docs = len(idxs) * K1
        rows = z["docs"][dptr:dptr + n_docs]
        dptr += n_docs
        Ld = int((rows >= 0).sum(1).max())
        doc_ids = np.zeros((n_docs, Ld), np.int64)
        doc_mask = np.zeros((n_docs, Ld), np.int64)
        for r, row in enumerate(rows):
            n = int((row >= 0).sum())
            doc_ids[r, :n] = row[:n]
            doc_mask[r, :n] = 1
        leaf = {k: (t.clone().requires_grad_(True) if k in names else t) for k, t in sd.items()}
        _, l1, l2 = OT.kd_losses(leaf, sdt, (g("concat_ids"), g("concat_id_mask"), g("target_ids"), g("target_id_mask")),
                                 num_layers=NL, num_heads=NH, docs=(torch.from_numpy(doc_ids), torch.from_numpy(doc_mask)),
                                 num_negatives=hp["num_negatives"])
        (l1 + l2).backward()
        assert abs(l1.item() - z["loss1"][step]) < 2e-5 * max(1.0, z["loss1"][step]), (step, l1.item(), z["loss1"][step])
        assert abs(l2.item() - z["loss2"][step]) < 2e-5 * max(1.0, z["loss2"][step]), (step, l2.item(), z["loss2"][step])
        grads = {k: leaf[k].grad for k in names if leaf[k].grad is not None}
        norm = float(torch.sqrt(sum((gg.double() ** 2).sum() for gg in grads.values())))
        assert abs(norm / z["grad_norm"][step] - 1) < 1e-4, (step, norm, z["grad_norm"][step])
        coef = OT.clip_coef(norm, hp["max_grad_norm"])
        lr = hp["lr"] * OT.linear_schedule(step, hp["warmup"], hp["t_total"])
        with torch.no_grad():
            for k, gg in grads.items():
                wd = 0.0 if any(nd in k for nd in no_decay) else hp["weight_decay"]
                OT.hf_adamw_step(sd[k], gg * coef, m[k], v[k], step + 1, lr, eps=hp["eps"], weight_decay=wd)
    for k in names:
        if k in w1:
            # (Adam turns an element whose true gradient is rounding noise into a +-lr = 2e-4 step of arbitrary sign; the
            #  two summation orders agree to <= 5e-6 on every element here: 2.5 % of one step)
            np.testing.assert_allclose(sd[k].numpy(), w1[k].numpy(), rtol=0, atol=2e-5, err_msg=k)
            assert (np.abs(sd[k].numpy() - w1[k].numpy()) > 3e-6).mean() < 1e-3, k
