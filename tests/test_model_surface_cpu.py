"""The reference's operator surface (model/models.py) is mirrored name for name; checkpoint I/O keeps the
reference's parameter names.  CPU only (no forward)."""
import os

import numpy as np
import pytest
import torch

from convdr_amd.model import models as M


def _cfg():
    return M.RobertaConfig(vocab_size=50, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                           max_position_embeddings=40)


def test_registry_matches_reference():
    assert set(M.MSMarcoConfigDict) == {"rdot_nll", "rdot_nll_multi_chunk", "dpr"}          # models.py:291-311
    for name, cls in (("rdot_nll", M.RobertaDot_NLL_LN), ("rdot_nll_multi_chunk", M.RobertaDot_CLF_ANN_NLL_MultiChunk),
                      ("dpr", M.BiEncoder)):
        c = M.MSMarcoConfigDict[name]
        assert c.model_class is cls and c.use_mean is False and c.name == name
    assert M.MSMarcoConfigDict["dpr"].config_class is M.BertConfig


def test_parameter_names_are_the_reference_names(golden_dir):
    z = np.load(os.path.join(golden_dir, "encoder_rdot_nll.npz"))
    ref_names = {k[2:] for k in z.files if k.startswith("w/")}
    cfg = M.RobertaConfig(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                          intermediate_size=256, max_position_embeddings=514)
    mine = set(M.RobertaDot_NLL_LN(cfg).state_dict())
    assert ref_names <= mine                                              # every reference tensor has a home
    assert all("pooler" in k for k in mine - ref_names)                  # extra: the transformers==2.3.0 pooler only
    d = np.load(os.path.join(golden_dir, "encoder_dpr.npz"))
    dpr_names = {k[2:] for k in d.files if k.startswith("w/")}
    args = type("A", (), {"bert_config": M.BertConfig(vocab_size=200, hidden_size=128, num_hidden_layers=2,
                                                       num_attention_heads=2, intermediate_size=256,
                                                       max_position_embeddings=514)})()
    assert dpr_names <= set(M.BiEncoder(args).state_dict())


def test_save_and_from_pretrained_round_trip(tmp_path):
    torch.manual_seed(0)
    m = M.RobertaDot_NLL_LN(_cfg())
    m.save_pretrained(str(tmp_path / "ckpt"))
    assert sorted(os.listdir(tmp_path / "ckpt")) == ["config.json", "pytorch_model.bin"]
    cfg = M.RobertaConfig.from_pretrained(str(tmp_path / "ckpt"), num_labels=2, finetuning_task="MSMarco")
    m2 = M.MSMarcoConfigDict["rdot_nll"].model_class.from_pretrained(str(tmp_path / "ckpt"), config=cfg)
    for (k, a), (k2, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k == k2 and torch.equal(a, b)
    # a checkpoint written by current transformers has no pooler: must still load
    sd = {k: v for k, v in m.state_dict().items() if "pooler" not in k}
    torch.save(sd, tmp_path / "ckpt" / "pytorch_model.bin")
    M.RobertaDot_NLL_LN.from_pretrained(str(tmp_path / "ckpt"))
    # ... but a missing encoder weight is an error
    sd.pop("roberta.encoder.layer.0.output.dense.weight")
    torch.save(sd, tmp_path / "ckpt" / "pytorch_model.bin")
    with pytest.raises(KeyError):
        M.RobertaDot_NLL_LN.from_pretrained(str(tmp_path / "ckpt"))


def test_resize_token_embeddings_and_optimizer_groups():
    from types import SimpleNamespace
    from convdr_amd.train import get_optimizer
    m = M.RobertaDot_NLL_LN(_cfg())
    old = m.roberta.embeddings.word_embeddings.weight.detach().clone()
    m.resize_token_embeddings(53)                                           # run_convdr_train.py:474 (<response> token)
    w = m.roberta.embeddings.word_embeddings.weight
    assert w.shape == (53, 128) and torch.equal(w[:50], old) and m.config.vocab_size == 53
    opt = get_optimizer(SimpleNamespace(learning_rate=1e-5, adam_epsilon=1e-8), m, weight_decay=0.01)
    decay, no_decay = ({id(p) for p in g["params"]} for g in opt.param_groups)
    names = dict(m.named_parameters())
    assert id(names["norm.weight"]) in decay                                # 'LayerNorm.weight' does not match 'norm.weight'
    assert id(names["roberta.embeddings.LayerNorm.weight"]) in no_decay and id(names["embeddingHead.bias"]) in no_decay
    assert opt.param_groups[0]["weight_decay"] == 0.01 and opt.param_groups[1]["weight_decay"] == 0.0


def test_abstract_and_error_conventions():
    class X(M.EmbeddingMixin):
        pass
    x = X(None)
    assert x.use_mean is False
    with pytest.raises(NotImplementedError):
        x.query_emb(None, None)
    with pytest.raises(KeyError):
        M.MSMarcoConfigDict["nope"]
    t = torch.arange(24.).view(2, 3, 4)
    mask = torch.tensor([[1, 1, 0], [1, 0, 0]])
    np.testing.assert_allclose(x.masked_mean(t, mask).numpy(), [[2, 3, 4, 5], [12, 13, 14, 15]])


def test_linear_schedule_matches_oracle():
    """run_convdr_train.py:71-74 (transformers.get_linear_schedule_with_warmup): the product's LambdaLR factor against the
    oracle's restatement, incl. the warm-up ramp, the decay and the clamp at zero past the last step."""
    import torch
    from convdr_amd import train as TR
    from oracle import train as OT
    p = torch.nn.Parameter(torch.zeros(1))
    for warm, total in ((0, 10), (3, 10), (5, 5), (1, 3)):
        opt = torch.optim.SGD([p], lr=1.0)
        sched = TR.get_linear_schedule_with_warmup(opt, warm, total)
        for step in range(total + 3):
            assert abs(opt.param_groups[0]["lr"] - OT.linear_schedule(step, warm, total)) < 1e-12, (warm, total, step)
            opt.step()
            sched.step()
