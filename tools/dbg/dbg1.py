import sys; sys.path.insert(0,'.')
import numpy as np, torch
from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
from oracle import encoder as OE
for layers,H,heads,I in [(1,768,12,3072),(1,256,4,512),(1,512,8,1024),(2,768,12,3072)]:
    torch.manual_seed(0)
    cfg = RobertaConfig(vocab_size=1000, hidden_size=H, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=I)
    model = MSMarcoConfigDict["rdot_nll"].model_class(cfg)
    rs = np.random.RandomState(0)
    B,L=12,128
    lens=[128,100,65,64,63,33,32,31,17,8,2,1]
    ids = rs.randint(3, 1000, size=(B, L)).astype(np.int64); ids[:,0]=0
    mask=np.zeros((B,L),np.int64)
    for b,n in enumerate(lens): mask[b,:n]=1; ids[b,n:]=0
    sd={k:v.detach().clone() for k,v in model.state_dict().items()}
    ref=OE.rdot_nll_emb(sd, torch.from_numpy(ids), torch.from_numpy(mask), num_layers=layers, num_heads=heads).numpy()
    model=model.cuda().eval()
    with torch.no_grad():
        emb=model.body_emb(torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()).cpu().numpy()
    cos=(emb*ref).sum(1)/np.sqrt((emb*emb).sum(1)*(ref*ref).sum(1))
    print(layers,H, "nan" if np.isnan(emb).any() else "ok", cos.min())
