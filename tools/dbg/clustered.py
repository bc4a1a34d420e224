import sys; sys.path.insert(0,'.')
import numpy as np, torch, time
from convdr_amd.search import FlatIPIndex
torch.manual_seed(0)
n,nq,d,k=1_000_000,256,768,100
g=torch.Generator(device="cuda").manual_seed(0)
for spread in (1.0, 0.5, 0.25):
    c=torch.randn(d,device="cuda",generator=g)            # shared component, norm ~27.7
    P=c[None,:]*0.9+torch.randn(n,d,device="cuda",generator=g)*spread*0.45
    Q=c[None,:]*0.9+torch.randn(nq,d,device="cuda",generator=g)*spread*0.45
    idx=FlatIPIndex(d); idx.add(P)
    try:
        D,I=idx.search(Q,k); msg="ok"
    except Exception as e:
        msg="ERR "+str(e)[:80]
    em,band=(t.float().mean().item() for t in idx.last_counts(nq,k)) if msg=="ok" else (0,0)
    print("spread",spread,"cos(p,p')~%.3f"%float(torch.nn.functional.cosine_similarity(P[:1000],P[1000:2000]).mean()), msg, idx.stats, "emitted",em,"band",band)
    del P,idx
