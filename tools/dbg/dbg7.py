import sys, json; sys.path.insert(0,'.')
import numpy as np, torch
from types import SimpleNamespace
from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
from convdr_amd import train as TR
from oracle import train as OT, encoder as OE
z=np.load("tests/golden/train_step.npz"); cfg=json.loads(str(z["config"])); hp=json.loads(str(z["hyper"]))
sd0={k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w0/")}
def build():
    m=MSMarcoConfigDict["rdot_nll"].model_class(RobertaConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **cfg)); m.load_state_dict(sd0, strict=False); return m
student, teacher = build().cuda(), build().cuda()
idxs=z["batches"][0]; K1=hp["num_negatives"]+1
g=lambda k: torch.from_numpy(np.stack([z["ex/%d/%s"%(i,k)] for i in idxs]))
n_docs=len(idxs)*K1; rows=z["docs"][:n_docs]
doc_ids=np.zeros((n_docs,512),np.int64); doc_mask=np.zeros((n_docs,512),np.int64)
for r,row in enumerate(rows):
    n=int((row>=0).sum()); doc_ids[r,:n]=row[:n]; doc_mask[r,:n]=1
batch=(g("concat_ids"),g("concat_id_mask"),g("target_ids"),g("target_id_mask"))
# oracle
sd={k:v.clone().requires_grad_(v.dtype.is_floating_point) for k,v in sd0.items()}
embs,l1,l2=OT.kd_losses(sd, sd0, batch, num_layers=2, num_heads=2, docs=(torch.from_numpy(doc_ids), torch.from_numpy(doc_mask)), num_negatives=hp["num_negatives"])
(l1+l2).backward()
ref={k:v.grad for k,v in sd.items() if v.requires_grad and v.grad is not None}
print("oracle losses", l1.item(), l2.item(), "ref total norm", float(torch.sqrt(sum((v.double()**2).sum() for v in ref.values()))))
# ours
student.train(); teacher.eval()
e=student(batch[0].cuda(), batch[1].cuda())
with torch.no_grad():
    t=teacher(batch[2].cuda(), batch[3].cuda())
    d=torch.cat([teacher(torch.from_numpy(doc_ids[i:i+8]).cuda(), torch.from_numpy(doc_mask[i:i+8]).cuda(), is_query=False) for i in range(0,n_docs,8)]).view(len(idxs),K1,-1)
L1=TR.mse_loss(e,t); L2=TR.ranking_loss(e,d)
(L1+L2).backward()
print("ours losses", L1.item(), L2.item())
tot=0
rowsout=[]
for n,p in student.named_parameters():
    if n in ref and p.grad is not None:
        a=p.grad.detach().cpu().double().reshape(-1); b=ref[n].double().reshape(-1)
        tot+=float((a**2).sum())
        rowsout.append((float(b.norm()), n, float(a.norm()/max(b.norm(),1e-30)), float(a@b/(a.norm()*b.norm()+1e-30))))
print("ours total norm", tot**0.5)
for r in sorted(rowsout, reverse=True)[:14]: print("%8.4f %-60s ratio %.4f cos %.5f"%r)
# embedding-level check of d_embs
de_ref=torch.autograd.grad
