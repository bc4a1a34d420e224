import sys; sys.path.insert(0,'.')
import numpy as np, torch
from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
rs = np.random.RandomState(0)
def run(layers,H,heads,I,B,L,lens,vocab=1000):
    torch.manual_seed(0)
    model = MSMarcoConfigDict["rdot_nll"].model_class(RobertaConfig(vocab_size=vocab,hidden_size=H,num_hidden_layers=layers,num_attention_heads=heads,intermediate_size=I))
    ids = rs.randint(3, vocab, size=(B, L)).astype(np.int64); ids[:,0]=0
    mask=np.zeros((B,L),np.int64)
    for b,n in enumerate(lens): mask[b,:n]=1; ids[b,n:]=0
    model=model.cuda().eval()
    outs=[]
    for t in range(6):
        with torch.no_grad(): outs.append(model.body_emb(torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()))
    d=[float((o-outs[0]).abs().max()) for o in outs[1:]]
    print(layers,H,I,B,L,"maxdiff",max(d))
lens12=[128,100,65,64,63,33,32,31,17,8,2,1]
run(1,768,12,3072,12,128,lens12)
run(1,128,2,256,12,128,lens12)
run(1,768,12,3072,1,128,[128])
run(1,768,12,3072,4,128,[128]*4)
run(1,768,12,3072,2,128,[64,64])
run(1,768,12,3072,3,128,[100,30,5])
run(12,768,12,3072,1,128,[128])
run(12,768,12,3072,12,128,lens12)
