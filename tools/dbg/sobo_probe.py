import os, sys, tempfile, shutil
sys.path.insert(0, '.')
import numpy as np, torch
from convdr_amd import blocks
from convdr_amd.search import FlatIPIndex, search_one_by_one
from oracle import search as OS
n, d, nq, k = 200_000, 768, 64, 100
g = torch.Generator(device="cuda").manual_seed(0)
P = torch.randn(n, d, device="cuda", generator=g)
Q = torch.randn(nq, d, device="cuda", generator=g)
index = FlatIPIndex(d); index.add(P)
host = P.cpu().numpy()
td = tempfile.mkdtemp()
try:
    for b in range(2):
        rows = host if b == 0 else np.ascontiguousarray(host[::-1])
        blocks.dump_block(os.path.join(td, "passage__emb_p__data_obj_%d.pb" % b), rows)
        blocks.dump_block(os.path.join(td, "passage__embid_p__data_obj_%d.pb" % b), np.arange(b * n, (b + 1) * n, dtype=np.int64))
    mD, mI = search_one_by_one(td, FlatIPIndex(d), Q.cpu().numpy(), k)
    Dm, Im = index.search_tensors(Q, k)
    Im = Im.cpu().numpy(); Dm = Dm.cpu().numpy()
    print("shape", mD.shape, mI.shape)
    print("even ranks == resident:", (mI[:, 0:2 * k:2][:, :k // 2] == Im[:, :k // 2]).all())
    print("row0 merged ids", mI[0, :8], "scores", mD[0, :8])
    print("row0 resident ids", Im[0, :4], "scores", Dm[0, :4])
    oD, oI = OS.search_one_by_one([(host, np.arange(n, dtype=np.int64)), (np.ascontiguousarray(host[::-1]), np.arange(n, 2 * n, dtype=np.int64))], Q.cpu().numpy(), k)
    print("vs oracle: I equal", np.array_equal(mI, oI), "D equal", np.array_equal(mD, oD))
finally:
    shutil.rmtree(td, ignore_errors=True)
