import sys; sys.path.insert(0,'.')
import numpy as np, torch
from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
from oracle import encoder as OE
torch.manual_seed(0)
model = MSMarcoConfigDict["rdot_nll"].model_class(RobertaConfig())
with torch.no_grad():
    for n, p in model.named_parameters():
        if n.endswith("bias"): p.normal_(0, 0.02)
        elif "LayerNorm.weight" in n or n == "norm.weight": p.add_(torch.randn_like(p) * 0.05)
rs = np.random.RandomState(0)
B, L = 12, 128
lens = [128, 100, 65, 64, 63, 33, 32, 31, 17, 8, 2, 1]
ids = rs.randint(3, 50000, size=(B, L)).astype(np.int64); ids[:, 0] = 0
mask = np.zeros((B, L), np.int64)
for b, n in enumerate(lens): mask[b, :n] = 1; ids[b, n:] = 0
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
ref = OE.rdot_nll_emb(sd, torch.from_numpy(ids), torch.from_numpy(mask), num_layers=12, num_heads=12).numpy()
model = model.cuda().eval()
with torch.no_grad(): e = model.body_emb(torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()).cpu().numpy()
cos=(e*ref).sum(1)/np.sqrt((e*e).sum(1)*(ref*ref).sum(1))
print("1-cos:", np.array2string(1-cos, precision=2), "max abs err", np.abs(e-ref).max())
