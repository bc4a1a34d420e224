import sys; sys.path.insert(0,'.')
import numpy as np, torch, json, os
from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
z=np.load("tests/golden/encoder_rdot_nll.npz")
cfg=json.loads(str(z["config"]))
model=MSMarcoConfigDict["rdot_nll"].model_class(RobertaConfig(**cfg))
model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")}, strict=False)
model=model.cuda().eval()
for case in ("L16","L64","L510"):
    ids,mask=torch.from_numpy(z[case+"/ids"]).cuda(), torch.from_numpy(z[case+"/mask"]).cuda()
    for trial in range(4):
        with torch.no_grad():
            e=model(ids,mask)
            nan1=torch.isnan(e).any().item()
            # poison the workspace and run again
            ws=model.roberta._ws
            ws[: ws.numel()//4*4].view(torch.float32).fill_(float("nan"))
            e2=model(ids,mask)
            nan2=torch.isnan(e2).any().item()
        print(case, trial, "first", nan1, "poisoned", nan2, (e2.cpu().numpy()-z[case+"/emb"]).__abs__().max() if not nan2 else None)
