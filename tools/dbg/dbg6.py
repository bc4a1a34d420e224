import sys; sys.path.insert(0,'.')
import ctypes as C
import numpy as np, torch
from convdr_amd import _lib
from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
rs = np.random.RandomState(0)
torch.manual_seed(0)
H,heads,I,B,L=768,12,3072,4,128
lens=[128,100,65,8]
model = MSMarcoConfigDict["rdot_nll"].model_class(RobertaConfig(vocab_size=1000,hidden_size=H,num_hidden_layers=1,num_attention_heads=heads,intermediate_size=I))
ids = rs.randint(3, 1000, size=(B, L)).astype(np.int64); ids[:,0]=0
mask=np.zeros((B,L),np.int64)
for b,n in enumerate(lens): mask[b,:n]=1; ids[b,n:]=0
model=model.cuda().eval()
rows=sum((n+7)//8*8 for n in lens)
names="tok_id tok_pos X Q K Vt ctx Hm Y cls_b cls_y cls_f head_y".split()
snaps=[]
for t in range(4):
    with torch.no_grad(): e=model.body_emb(torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda())
    torch.cuda.synchronize()
    c,w,_=model.roberta.packed((model.embeddingHead, model.norm))
    out=(C.c_int64*14)()
    _lib.lib().convdr_encoder_debug_layout(C.byref(c), rows, B, out)
    ws=model.roberta._ws
    offs=list(out)[:13]; ldt=out[13]
    sizes={"tok_id":rows*4,"tok_pos":rows*4,"X":rows*H*2,"Q":rows*H*2,"K":rows*H*2,"Vt":H*ldt*2,"ctx":rows*H*2,"Hm":rows*I*2,"Y":rows*H*4,"cls_b":B*H*2,"cls_y":B*H*4,"cls_f":B*H*4,"head_y":B*768*4}
    snap={n: ws[o:o+sizes[n]].clone() for n,o in zip(names,offs)}
    snap["out"]=e.clone()
    snaps.append(snap)
for n in names+["out"]:
    d=[int((s[n]!=snaps[0][n]).sum().item()) for s in snaps[1:]]
    print(n, "differing bytes vs run0:", d)
# where do Q diffs sit?
q0=snaps[0]["Q"].view(torch.bfloat16).view(rows,H).float(); q1=snaps[1]["Q"].view(torch.bfloat16).view(rows,H).float()
dr=(q0!=q1).nonzero()
print("Q diff positions (row,col) sample:", dr[:12].tolist(), "count", len(dr))
x0=snaps[0]["Y"].view(torch.float32).view(rows,H); x1=snaps[1]["Y"].view(torch.float32).view(rows,H)
dr=(x0!=x1).nonzero(); print("Y diffs", len(dr), dr[:8].tolist())
