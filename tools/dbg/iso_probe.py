import sys, os, time, json, tempfile, shutil
sys.path.insert(0, ".")
import numpy as np, torch, bench
from convdr_amd import blocks
from convdr_amd.search import FlatIPIndex, search_one_by_one
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
def kd(tag):
    d = bench.train_kd_measure(dev, 0, 1, False, 10, 3, 64, with_kernels=False, dropout=0.1)
    print(tag, "%.3f ms" % d["ms_per_step"], flush=True)
kd("fresh process")
n, d = 1_000_000, 768
P = torch.randn(n, d, device=dev)
index = FlatIPIndex(d, device=dev); index.add(P); del P
Q = torch.randn(1000, d, device=dev)
index.search_tensors(Q, 100)
kd("after resident index + search")
td = tempfile.mkdtemp()
host = index._p32.cpu().numpy()
path = os.path.join(td, "passage__emb_p__data_obj_0.pb")
blocks.dump_block(path, host)
with blocks.BlockView(path) as bv:
    fresh = FlatIPIndex(d, device=dev); fresh.add(bv); torch.cuda.synchronize(); del fresh
kd("after block_load leg")
blocks.dump_block(os.path.join(td, "passage__embid_p__data_obj_0.pb"), np.arange(n, dtype=np.int64))
blocks.dump_block(os.path.join(td, "passage__emb_p__data_obj_1.pb"), np.ascontiguousarray(host[::-1]))
blocks.dump_block(os.path.join(td, "passage__embid_p__data_obj_1.pb"), np.arange(n, 2 * n, dtype=np.int64))
del host
gi = FlatIPIndex(d, device=dev)
search_one_by_one(td, gi, Q.cpu().numpy(), 100)
del gi
shutil.rmtree(td)
kd("after search_one_by_one files")
del index
torch.cuda.empty_cache()
kd("after empty_cache")
