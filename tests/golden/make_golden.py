#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING AND RUNNING the
reference (thunlp/ConvDR at /root/reference) in the build container.

Run here only (the reference never travels to the GPU box):

    python tests/golden/make_golden.py            # all groups
    python tests/golden/make_golden.py encoder    # one group

Nothing of the reference's source is stored: fixtures are inputs + outputs
(.npz) and, for text outputs, the produced lines.  The shims below are the ones
listed in SURVEY.md §8(c); they replace *absent third-party packages*, never
reference code:
  faiss / pytrec_eval / tensorboardX  -> empty module stubs
  transformers.AdamW (removed from HF) -> ``HFAdamW230`` below: a restatement of
      the pinned transformers==2.3.0 optimizer from its published algorithm
      (the arithmetic itself is therefore "parity unpinned"; what the fixture
      pins is the reference's *use* of it: param groups, lr, eps, clip, schedule)
  RobertaConfig.pretrained_config_archive_map -> {}
  configs built with return_dict=False (models.py:39 asserts a tuple)
  HFBertEncoder.init_encoder -> random-init tiny BERT (no network)
  FlatIP stand-in for faiss.IndexFlatIP: exact fp32 ``q @ x.T`` + stable
      descending argsort
"""
import importlib.machinery
import json
import logging
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _stub(name):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m
    return m


def import_reference():
    for n in ("faiss", "pytrec_eval", "tensorboardX"):
        _stub(n)
    sys.modules["tensorboardX"].SummaryWriter = object
    import torch
    from transformers import (RobertaConfig, RobertaModel, BertModel, BertConfig,  # noqa: F401
                              RobertaForSequenceClassification)
    sys.modules["transformers"].AdamW = HFAdamW230()
    RobertaConfig.pretrained_config_archive_map = {}
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import model.models as M
    import utils.util as U
    import utils.dpr_utils as DU
    import data.tokenizing as T
    return M, U, DU, T


def HFAdamW230():
    import math
    import torch

    class AdamW(torch.optim.Optimizer):
        """transformers==2.3.0 optimization.AdamW, restated."""

        def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
            super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                          correct_bias=correct_bias))

        def step(self, closure=None):
            for group in self.param_groups:
                for p in group["params"]:
                    if p.grad is None:
                        continue
                    grad = p.grad.data
                    state = self.state[p]
                    if len(state) == 0:
                        state["step"] = 0
                        state["exp_avg"] = torch.zeros_like(p.data)
                        state["exp_avg_sq"] = torch.zeros_like(p.data)
                    m, v = state["exp_avg"], state["exp_avg_sq"]
                    b1, b2 = group["betas"]
                    state["step"] += 1
                    m.mul_(b1).add_(grad, alpha=1.0 - b1)
                    v.mul_(b2).addcmul_(grad, grad, value=1.0 - b2)
                    denom = v.sqrt().add_(group["eps"])
                    step_size = group["lr"]
                    if group["correct_bias"]:
                        step_size = step_size * math.sqrt(1.0 - b2 ** state["step"]) / (1.0 - b1 ** state["step"])
                    p.data.addcdiv_(m, denom, value=-step_size)
                    if group["weight_decay"] > 0.0:
                        p.data.add_(p.data, alpha=-group["lr"] * group["weight_decay"])
    return AdamW


# ----------------------------------------------------------------------------
TINY = dict(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
            intermediate_size=256, max_position_embeddings=514)


def tiny_roberta_config(dropout=0.0):
    from transformers import RobertaConfig
    return RobertaConfig(type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5, return_dict=False,
                         hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout,
                         num_labels=2, **TINY)


def tiny_bert_config():
    from transformers import BertConfig
    return BertConfig(type_vocab_size=2, pad_token_id=0, layer_norm_eps=1e-12, return_dict=False,
                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **TINY)


def synth_ids(rng, B, L, lens, bos=0, vocab=200, pad1_at=None):
    """right-padded with id 0 / mask 0 like utils/util.py:163-185."""
    ids = rng.randint(3, vocab, size=(B, L)).astype(np.int64)
    ids[:, 0] = bos
    mask = np.zeros((B, L), np.int64)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
        ids[b, n:] = 0
    if pad1_at is not None:          # a real token equal to RoBERTa's pad id 1
        ids[pad1_at[0], pad1_at[1]] = 1
    return ids, mask


def sd_to_np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def gen_encoder():
    import torch
    M, U, DU, T = import_reference()
    torch.manual_seed(0)
    rng = np.random.RandomState(0)
    # ---- rdot_nll -----------------------------------------------------------
    cfg = tiny_roberta_config()
    model = M.MSMarcoConfigDict["rdot_nll"].model_class(cfg)
    # non-trivial LayerNorm/bias values so the fixture exercises them
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.05)
            elif "LayerNorm.weight" in n or n == "norm.weight":
                p.add_(torch.randn_like(p) * 0.1)
    model.eval()
    out = {"config": json.dumps({**TINY, "layer_norm_eps": 1e-5, "type_vocab_size": 1, "pad_token_id": 1})}
    for k, v in sd_to_np(model.state_dict()).items():
        out["w/" + k] = v
    cases = {
        "L16": (4, 16, [16, 9, 1, 12], (3, 5)),
        "L64": (3, 64, [64, 33, 50], None),
        "L510": (2, 510, [510, 77], None),
    }
    for name, (B, L, lens, pad1) in cases.items():
        ids, mask = synth_ids(rng, B, L, lens, pad1_at=pad1)
        with torch.no_grad():
            tid, tm = torch.from_numpy(ids), torch.from_numpy(mask)
            q = model(tid, tm)                              # NLL.forward -> query_emb
            b = model(tid, tm, is_query=False)              # -> body_emb
            hs = model.roberta(input_ids=tid, attention_mask=tm, output_hidden_states=True)
        assert torch.equal(q, b)
        out[name + "/ids"], out[name + "/mask"] = ids, mask
        out[name + "/emb"] = q.numpy()
        out[name + "/cls_last"] = hs[0][:, 0].numpy()
        if name == "L16":
            hidden = hs[-1] if isinstance(hs[-1], (tuple, list)) else hs[2]
            out[name + "/hidden_states"] = np.stack([h.numpy() for h in hidden])
    # triple-loss entry (models.py:66-75)
    ids_q, m_q = synth_ids(rng, 4, 12, [12, 7, 9, 12])
    ids_a, m_a = synth_ids(rng, 4, 20, [20, 15, 20, 3])
    ids_b, m_b = synth_ids(rng, 4, 20, [11, 20, 18, 20])
    with torch.no_grad():
        loss = model(*(torch.from_numpy(x) for x in (ids_q, m_q, ids_a, m_a, ids_b, m_b)))[0]
    for n, v in dict(ids_q=ids_q, m_q=m_q, ids_a=ids_a, m_a=m_a, ids_b=ids_b, m_b=m_b).items():
        out["triple/" + n] = v
    out["triple/loss"] = np.array(loss.item(), np.float64)
    # rdot_nll_multi_chunk (models.py:159-188, 78-126): same weights, documents of 2 x 512 tokens, one pure-padding chunk
    mc = M.MSMarcoConfigDict["rdot_nll_multi_chunk"].model_class(cfg)
    mc.load_state_dict(model.state_dict())
    mc.eval()
    def chunked(lens_per_chunk):
        ids = np.zeros((len(lens_per_chunk), 1024), np.int64)
        mask = np.zeros((len(lens_per_chunk), 1024), np.int64)
        for b, lens in enumerate(lens_per_chunk):
            for c, n in enumerate(lens):
                if n:
                    ids[b, 512 * c:512 * c + n] = rng.randint(3, 200, size=n)
                    ids[b, 512 * c] = 0
                    mask[b, 512 * c:512 * c + n] = 1
        return ids, mask
    ids_a, m_a = chunked([[512, 100], [30, 0]])
    ids_b, m_b = chunked([[200, 512], [512, 7]])
    ids_q2, m_q2 = synth_ids(rng, 2, 12, [12, 5])
    with torch.no_grad():
        emb_a = mc.body_emb(torch.from_numpy(ids_a), torch.from_numpy(m_a))
        loss_mc = mc(*(torch.from_numpy(x) for x in (ids_q2, m_q2, ids_a, m_a, ids_b, m_b)))[0]
    for n, v in dict(ids_a=ids_a, m_a=m_a, ids_b=ids_b, m_b=m_b, ids_q=ids_q2, m_q=m_q2).items():
        out["mc/" + n] = v
    out["mc/emb_a"] = emb_a.numpy()
    out["mc/loss"] = np.array(loss_mc.item(), np.float64)
    np.savez_compressed(os.path.join(HERE, "encoder_rdot_nll.npz"), **out)

    # ---- dpr ----------------------------------------------------------------
    bcfg = tiny_bert_config()
    M.HFBertEncoder.init_encoder = classmethod(lambda cls, args, dropout=0.1: cls(bcfg))
    torch.manual_seed(1)
    bi = M.MSMarcoConfigDict["dpr"].model_class(None)
    with torch.no_grad():
        for n, p in bi.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.05)
            elif "LayerNorm.weight" in n:
                p.add_(torch.randn_like(p) * 0.1)
    bi.eval()
    out = {"config": json.dumps({**TINY, "layer_norm_eps": 1e-12, "type_vocab_size": 2, "pad_token_id": 0})}
    for k, v in sd_to_np(bi.state_dict()).items():
        out["w/" + k] = v
    ids, mask = synth_ids(rng, 4, 24, [24, 10, 17, 2], bos=101)
    with torch.no_grad():
        tid, tm = torch.from_numpy(ids), torch.from_numpy(mask)
        out["q_emb"] = bi(tid, tm).numpy()
        out["b_emb"] = bi(tid, tm, is_query=False).numpy()
        qa = bi(tid, tm, tid, tm)
        assert isinstance(qa, tuple) and len(qa) == 2
        out["pair_loss"] = np.array(bi(tid, tm, tid, tm, torch.flip(tid, [0]), torch.flip(tm, [0]))[0].item())
    out["ids"], out["mask"] = ids, mask
    np.savez_compressed(os.path.join(HERE, "encoder_dpr.npz"), **out)
    print("encoder fixtures written")


# ----------------------------------------------------------------------------
class FlatIPStandIn:
    """SURVEY.md §8(c) shim 6."""

    def __init__(self, d):
        self.x = None

    def add(self, x):
        self.x = np.asarray(x, np.float32)

    def search(self, q, k):
        s = np.asarray(q, np.float32) @ self.x.T
        I = np.argsort(-s, axis=1, kind="stable")[:, :k]
        return np.take_along_axis(s, I, 1), I.astype(np.int64)

    def reset(self):
        self.x = None


def synth_corpus(seed, n, d):
    return np.random.RandomState(seed).randn(n, d).astype(np.float32)


def gen_search():
    import contextlib
    import io
    sys.argv = sys.argv[:1]
    M, U, DU, T = import_reference()
    # the driver module imports fine once faiss is stubbed
    sys.path.insert(0, os.path.join(REF, "drivers"))
    import run_convdr_inference as R

    def run(case, blocks, Q, topN):
        with tempfile.TemporaryDirectory() as td:
            for b, (emb, embid) in enumerate(blocks):
                for pre, arr in (("passage__emb_p_", emb), ("passage__embid_p_", embid)):
                    with open(os.path.join(td, "%s_data_obj_%d.pb" % (pre, b)), "wb") as h:
                        pickle.dump(arr, h, protocol=4)      # == utils/util.py:108-111
            with contextlib.redirect_stdout(io.StringIO()):
                mD, mI = R.search_one_by_one(td, FlatIPStandIn(768), Q, topN)
        # record the smallest adjacent score gap: the parity tests treat ranks whose
        # reference scores differ by < 1e-3 as exchangeable (SURVEY.md §7 hard part 1)
        gaps = []
        for emb, _ in blocks:
            s = np.sort((Q.astype(np.float64) @ emb.astype(np.float64).T), axis=1)[:, ::-1][:, :topN + 1]
            gaps.append(np.min(s[:, :-1] - s[:, 1:]))
        print("case", case, "min adjacent gap", min(gaps))
        return mD, mI

    out = {}
    # (a) three ragged blocks, round-robin ids like a 3-rank encode run (util.py:422-424)
    d, topN, nq = 768, 100, 16
    sizes, seeds = [1500, 1000, 700], [11, 12, 13]
    total = sum(sizes)
    blocks = []
    for r, (n, s) in enumerate(zip(sizes, seeds)):
        blocks.append((synth_corpus(s, n, d), (np.arange(n, dtype=np.int64) * 3 + r)))
    Q = synth_corpus(1234, nq, d)
    mD, mI = run("a", blocks, Q, topN)
    out["a/sizes"], out["a/seeds"], out["a/qseed"] = np.array(sizes), np.array(seeds), np.array(1234)
    out["a/topN"], out["a/merged_D"], out["a/merged_I"] = np.array(topN), mD, mI
    # (b) one block -> [nq, topN]
    mD, mI = run("b", blocks[:1], Q, topN)
    out["b/merged_D"], out["b/merged_I"] = mD, mI
    # (c) exact ties: duplicated vectors inside a block and across blocks, small k
    base = synth_corpus(21, 300, d)
    b0 = np.concatenate([base[:200], base[50:60]])            # dup rows 50..59 at 200..209
    b1 = np.concatenate([base[40:70], base[200:300]])         # rows 40..69 again in the next block
    blocks_c = [(b0, np.arange(len(b0), dtype=np.int64)), (b1, 1000 + np.arange(len(b1), dtype=np.int64))]
    Qc = base[45:53] + 0.0                                     # queries = corpus rows -> top hits are the dups
    mD, mI = run("ties", blocks_c, Qc, 10)
    out["c/merged_D"], out["c/merged_I"], out["c/topN"] = mD, mI, np.array(10)
    # (d) EvalDevQuery text outputs for case (a) with duplicate pids
    with tempfile.TemporaryDirectory() as td:
        rs = np.random.RandomState(5)
        offset2pid = (rs.permutation(3 * max(sizes)) // 2 * 2 + 7).tolist()       # every pid appears twice
        qids = ["%d_%d" % (31 + i // 4, 1 + i % 4) for i in range(nq)]
        with open(os.path.join(td, "queries.raw.tsv"), "w") as f:
            for q in qids:
                f.write("%s\tquery text %s\n" % (q, q))
        with open(os.path.join(td, "collection.tsv"), "w") as f:
            for pid in sorted(set(offset2pid)):
                f.write("%d\tpassage %d body\n" % (pid, pid))
        raw = [["hist %s" % q, "cur %s" % q] for q in qids]
        pos = {qids[0]: {offset2pid[int(out["a/merged_I"][0, 0])]: 2}}
        R.EvalDevQuery(qids, out["a/merged_D"], pos, out["a/merged_I"], topN,
                       os.path.join(td, "o.jsonl"), os.path.join(td, "o.trec"), offset2pid, td, "raw",
                       raw_sequences=raw)
        out["d/offset2pid"] = np.array(offset2pid, np.int64)
        out["d/qids"] = np.array(qids)
        out["d/pos_qid"], out["d/pos_pid"], out["d/pos_label"] = np.array(qids[0]), np.array(list(pos[qids[0]])[0]), np.array(2)
        out["d/trec"] = np.array(open(os.path.join(td, "o.trec")).read())
        out["d/jsonl"] = np.array(open(os.path.join(td, "o.jsonl")).read())
    np.savez_compressed(os.path.join(HERE, "search.npz"), **out)
    print("search fixtures written")


# ----------------------------------------------------------------------------
def gen_encode_loop():
    """Drive gen_passage_embeddings.StreamInferenceDoc under a 1-rank gloo group
    (SURVEY.md §8(c) shim 7) -> real block files; keep their bytes' content."""
    import torch
    import torch.distributed as dist
    sys.argv = sys.argv[:1]
    M, U, DU, T = import_reference()
    sys.path.insert(0, os.path.join(REF, "drivers"))
    import gen_passage_embeddings as G
    torch.manual_seed(3)
    model = M.MSMarcoConfigDict["rdot_nll"].model_class(tiny_roberta_config())
    model.eval()
    rng = np.random.RandomState(3)
    N, L = 37, 16
    lens = rng.randint(1, L + 1, size=N)
    lens[0] = L
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "passages")
        rows = []
        with open(path, "wb") as f:           # byte layout of tokenizing.py:116 minus the 8-byte pid prefix (:44)
            for n in lens:
                ids = [0] + rng.randint(3, 200, size=n - 1).tolist()
                rows.append(ids)
                f.write(int(n).to_bytes(4, "big") + np.array((ids + [0] * L)[:L], np.int32).tobytes())
        with open(path + "_meta", "w") as f:
            json.dump({"type": "int32", "total_number": N, "embedding_size": L}, f)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        dist.init_process_group("gloo", rank=0, world_size=1)
        args = types.SimpleNamespace(per_gpu_eval_batch_size=8, local_rank=0, rank=0, world_size=1,
                                     device=torch.device("cpu"), output_dir=os.path.join(td, "out"),
                                     max_seq_length=L, max_query_length=L)
        wrapped = types.SimpleNamespace(module=model, eval=model.eval)
        cache = U.EmbeddingCache(path)
        with cache as emb:
            G.StreamInferenceDoc(args, wrapped, T.GetProcessingFn(args, query=False), "passage_", emb,
                                 is_query_inference=False, merge=False)
        dist.destroy_process_group()
        e = pickle.load(open(os.path.join(td, "out", "passage__emb_p__data_obj_0.pb"), "rb"))
        i = pickle.load(open(os.path.join(td, "out", "passage__embid_p__data_obj_0.pb"), "rb"))
        token_bytes = open(path, "rb").read()
    out = {"token_cache": np.frombuffer(token_bytes, np.uint8), "N": np.array(N), "L": np.array(L),
           "lens": lens.astype(np.int64), "emb": e, "embid": i,
           "emb_dtype": np.array(str(e.dtype)), "embid_dtype": np.array(str(i.dtype))}
    for k, v in sd_to_np(model.state_dict()).items():
        out["w/" + k] = v
    np.savez_compressed(os.path.join(HERE, "encode_loop.npz"), **out)
    print("encode-loop fixture written", e.shape, i[:5])


def gen_train(independent_teacher=False):
    """Drive the reference's own `train()` (run_convdr_train.py:41-252) for 4 optimizer steps on a synthetic dataset:
    KD (MSE) + ranking task, dropout 0, and record what it computed.
    independent_teacher: the drivers load teacher and student from ONE checkpoint (loss1 starts at ~3e-4: a poorly
    conditioned KD fixture); the `train_step_b.npz` variant gives the teacher its own random initialisation, so the MSE
    term is O(1) and carries gradient from the first step."""
    import random
    import torch
    sys.argv = sys.argv[:1]
    M, U, DU, T = import_reference()
    sys.path.insert(0, os.path.join(REF, "drivers"))
    import run_convdr_train as R
    torch.manual_seed(7)
    cfg = tiny_roberta_config(dropout=0.0)
    student = M.MSMarcoConfigDict["rdot_nll"].model_class(cfg)
    with torch.no_grad():
        for n, p in student.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.05)
            elif "LayerNorm.weight" in n or n == "norm.weight":
                p.add_(torch.randn_like(p) * 0.1)
    init_sd = {k: v.detach().clone() for k, v in student.state_dict().items()}
    teacher = M.MSMarcoConfigDict["rdot_nll"].model_class(cfg)
    teacher.load_state_dict(init_sd)            # the reference loads teacher and student from the same checkpoint
    if independent_teacher:
        torch.manual_seed(1007)
        teacher = M.MSMarcoConfigDict["rdot_nll"].model_class(cfg)
        with torch.no_grad():
            for n, p in teacher.named_parameters():
                if n.endswith("bias"):
                    p.normal_(0, 0.05)
                elif "LayerNorm.weight" in n or n == "norm.weight":
                    p.add_(torch.randn_like(p) * 0.1)
    teacher_sd = {k: v.detach().clone() for k, v in teacher.state_dict().items()}

    rng = np.random.RandomState(7)
    N, Lc, Lt, K = 16, 40, 12, 3
    examples = []
    for i in range(N):
        lc, lt = rng.randint(5, Lc + 1), rng.randint(3, Lt + 1)
        c_ids, c_mask = synth_ids(rng, 1, Lc, [lc])
        t_ids, t_mask = synth_ids(rng, 1, Lt, [lt])
        docs = [" ".join(str(x) for x in rng.randint(3, 200, size=rng.randint(4, 30))) for _ in range(K + 1)]
        examples.append(dict(idx=i, concat_ids=c_ids[0], concat_id_mask=c_mask[0], target_ids=t_ids[0],
                             target_id_mask=t_mask[0], documents=docs))

    log = dict(batches=[], docs=[], loss1=[], loss2=[], norms=[], scalars=[])

    class DS:
        def __len__(self):
            return N

        def __getitem__(self, i):
            return examples[i]

        def get_collate_fn(self, args, mode):
            def fn(feats):
                log["batches"].append([f["idx"] for f in feats])
                out = {k: torch.tensor(np.stack([f[k] for f in feats]), dtype=torch.long)
                       for k in ("concat_ids", "concat_id_mask", "target_ids", "target_id_mask")}
                out["documents"] = [f["documents"] for f in feats]
                return out
            return fn

    class Tok:
        def encode(self, text, text_pair=None, add_special_tokens=True, max_length=512):
            ids = [0] + [int(x) for x in text.split()] + [2]
            log["docs"].append(ids)
            return ids[:max_length]

    class Writer:
        def add_scalar(self, tag, v, step):
            log["scalars"].append((tag, float(v), int(step)))

    mse, ce = torch.nn.MSELoss(), torch.nn.CrossEntropyLoss()

    def loss_fn(a, b):
        v = mse(a, b); log["loss1"].append(float(v)); return v

    def loss_fn_2(a, b):
        v = ce(a, b); log["loss2"].append(float(v)); return v

    orig_clip = torch.nn.utils.clip_grad_norm_

    def clip(params, max_norm):
        n = orig_clip(params, max_norm); log["norms"].append(float(n)); return n
    torch.nn.utils.clip_grad_norm_ = clip
    with tempfile.TemporaryDirectory() as td:
        args = types.SimpleNamespace(
            per_gpu_train_batch_size=4, n_gpu=0, device=torch.device("cpu"), max_steps=3, num_train_epochs=1,
            gradient_accumulation_steps=1, learning_rate=2e-4, adam_epsilon=1e-8, weight_decay=0.01, warmup_steps=1,
            max_grad_norm=1.0, log_steps=1, save_steps=-1, no_mse=False, ranking_task=True, num_negatives=K,
            model_type="rdot_nll", output_dir=td, seed=42)
        try:
            gs, _ = R.train(args, DS(), student, teacher, loss_fn, logging.getLogger("golden"), Writer(),
                            cross_validate_id=0, loss_fn_2=loss_fn_2, tokenizer=Tok())
        finally:
            torch.nn.utils.clip_grad_norm_ = orig_clip
    out = {"config": json.dumps({**TINY, "layer_norm_eps": 1e-5, "type_vocab_size": 1, "pad_token_id": 1}),
           "steps": np.array(gs), "batches": np.array(log["batches"]), "loss1": np.array(log["loss1"]),
           "loss2": np.array(log["loss2"]), "grad_norm": np.array(log["norms"]),
           "docs": np.array([(d + [-1] * 64)[:64] for d in log["docs"]], np.int32),
           "hyper": json.dumps(dict(lr=2e-4, eps=1e-8, weight_decay=0.01, warmup=1, t_total=3, max_grad_norm=1.0,
                                    num_negatives=K, batch=4))}
    for i, e in enumerate(examples):
        for k in ("concat_ids", "concat_id_mask", "target_ids", "target_id_mask"):
            out["ex/%d/%s" % (i, k)] = e[k]
    for k, v in init_sd.items():
        out["w0/" + k] = v.numpy()
    for k, v in student.state_dict().items():
        out["w1/" + k] = v.detach().numpy()
    if independent_teacher:
        for k, v in teacher_sd.items():
            out["wt/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "train_step_b.npz" if independent_teacher else "train_step.npz"), **out)
    print("train fixture written: steps", gs, "loss1", log["loss1"], "loss2", log["loss2"], "norms", log["norms"])


# ----------------------------------------------------------------------------
def gen_evaluate():
    """Query-encode loop: run_convdr_inference.py:116-154 (``evaluate``) run on a stub dataset (the reference's
    ConvSearchDataset needs the tokenizer / raw files, which are out of scope; ``evaluate`` only needs __len__, __getitem__
    and get_collate_fn(args, "inference") -> dict with qid / concat_ids / concat_id_mask / history_utterances).  Weights:
    the tiny rdot_nll model already stored in encoder_rdot_nll.npz."""
    import contextlib
    import io
    from types import SimpleNamespace
    import torch
    sys.argv = sys.argv[:1]
    M, U, DU, T = import_reference()
    sys.path.insert(0, os.path.join(REF, "drivers"))
    import run_convdr_inference as R
    z = np.load(os.path.join(HERE, "encoder_rdot_nll.npz"))
    model = M.MSMarcoConfigDict["rdot_nll"].model_class(tiny_roberta_config())
    model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")})
    rng = np.random.RandomState(77)
    nq, L = 11, 48                                   # batches of 4 -> 4 + 4 + 3 (ragged last batch)
    lens = [48, 7, 30, 1, 19, 48, 12, 33, 5, 41, 26]
    ids, mask = synth_ids(rng, nq, L, lens)
    qids = ["%d_%d" % (31 + i // 4, 1 + i % 4) for i in range(nq)]
    hist = [["utt %d of %s" % (j, q) for j in range(1 + i % 3)] for i, q in enumerate(qids)]

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return nq

        def __getitem__(self, i):
            return i

        def get_collate_fn(self, args, mode):
            assert mode == "inference"

            def fn(idx):
                return {"qid": [qids[i] for i in idx], "concat_ids": torch.from_numpy(ids[idx]),
                        "concat_id_mask": torch.from_numpy(mask[idx]), "history_utterances": [hist[i] for i in idx]}
            return fn
    args = SimpleNamespace(per_gpu_eval_batch_size=4, n_gpu=1, device=torch.device("cpu"), seed=42)
    with contextlib.redirect_stdout(io.StringIO()):
        emb, emb2id, raw = R.evaluate(args, DS(), model, logging.getLogger("golden"))
    assert emb.shape == (nq, 768) and emb2id == qids and raw == hist
    np.savez_compressed(os.path.join(HERE, "evaluate.npz"), ids=ids, mask=mask, qids=np.array(qids),
                        hist=np.array(json.dumps(hist)), batch=np.array(4), embedding=emb,
                        embedding2id=np.array(emb2id), raw_sequences=np.array(json.dumps(raw)))
    print("evaluate fixture written", emb.shape, emb.dtype)


def gen_use_mean():
    """EmbeddingMixin with use_mean = True (models.py:19-23, :32-41): masked mean over the tokens instead of the CLS row.
    No registered config sets it (models.py:295-307) and no driver passes model_argobj, so this is surface only -- pinned
    here with the tiny rdot_nll weights of encoder_rdot_nll.npz."""
    from types import SimpleNamespace
    import torch
    M, U, DU, T = import_reference()
    z = np.load(os.path.join(HERE, "encoder_rdot_nll.npz"))
    model = M.RobertaDot_NLL_LN(tiny_roberta_config(), model_argobj=SimpleNamespace(use_mean=True))
    model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")})
    model.eval()
    assert model.use_mean is True
    rng = np.random.RandomState(88)
    ids, mask = synth_ids(rng, 5, 70, [70, 9, 1, 64, 33], pad1_at=(3, 7))
    with torch.no_grad():
        emb = model(torch.from_numpy(ids), torch.from_numpy(mask))
        body = model(torch.from_numpy(ids), torch.from_numpy(mask), is_query=False)
    assert torch.equal(emb, body)
    np.savez_compressed(os.path.join(HERE, "use_mean.npz"), ids=ids, mask=mask, emb=emb.numpy())
    print("use_mean fixture written", emb.shape)


GROUPS = {"encoder": gen_encoder, "search": gen_search, "encode_loop": gen_encode_loop, "train": gen_train,
          "train_b": lambda: gen_train(independent_teacher=True),
          "evaluate": gen_evaluate, "use_mean": gen_use_mean}

if __name__ == "__main__":
    want = sys.argv[1:] or list(GROUPS)
    sys.argv = sys.argv[:1]
    for g in want:
        GROUPS[g]()
