"""One tiny encoder forward on cuda:0 checked against the CPU oracle (used by __graft_entry__.smoke)."""
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
