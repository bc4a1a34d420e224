import sys; sys.path.insert(0,'.')
import numpy as np, torch
from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
rs = np.random.RandomState(0)
B,L=12,128
lens=[128,100,65,64,63,33,32,31,17,8,2,1]
ids = rs.randint(3, 50000, size=(B, L)).astype(np.int64); ids[:,0]=0
mask=np.zeros((B,L),np.int64)
for b,n in enumerate(lens): mask[b,:n]=1; ids[b,n:]=0
ids[1,7]=1
for layers in (1,2,12):
    torch.manual_seed(0)
    model = MSMarcoConfigDict["rdot_nll"].model_class(RobertaConfig(num_hidden_layers=layers))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"): p.normal_(0, 0.02)
            elif "LayerNorm.weight" in n or n == "norm.weight": p.add_(torch.randn_like(p) * 0.05)
    model=model.cuda().eval()
    res=[]
    for t in range(3):
        with torch.no_grad():
            e=model.body_emb(torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda())
        res.append(torch.isnan(e).any(1).int().cpu().numpy().tolist())
    print(layers, res)
