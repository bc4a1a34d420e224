import sys; sys.path.insert(0, '.')
import numpy as np, torch, time
from convdr_amd.search import FlatIPIndex
import bench
dev = torch.device("cuda")
d, k = 768, 100
def report(name, idx, Q):
    Dd, Id, st, tau = idx.search_device(Q, k)
    torch.cuda.synchronize()
    em, band = idx.last_counts(Q.shape[0], k)
    print(name, "n", idx.ntotal, "status hist", np.bincount(st.cpu().numpy(), minlength=4), "emitted mean/max", em.float().mean().item(), em.max().item(),
          "band mean/max", band.float().mean().item(), band.max().item(), "max_norm", idx._max_norm.item(), "qnorm", Q.norm(dim=1).mean().item())
    try:
        idx.search_tensors(Q, k); print("   certified:", idx.stats)
    except Exception as e:
        print("   ERR", str(e)[:100], idx.stats)
g = torch.Generator(device=dev).manual_seed(7)
for n in (30000, 1_000_000):
    c = torch.randn(d, device=dev, generator=g)
    Pc = 0.9 * c[None, :] + 0.12 * torch.randn(n, d, device=dev, generator=g)
    Qc = 0.9 * c[None, :] + 0.12 * torch.randn(1000, d, device=dev, generator=g)
    idx = FlatIPIndex(d, device=dev); idx.add(Pc)
    report("clustered", idx, Qc)
    del idx, Pc
model = bench.random_rdot_model().to(dev).eval()
tower, head = model.roberta, (model.embeddingHead, model.norm)
with torch.no_grad():
    embs = []
    for i in range(8):
        tok = bench.synthetic_tokens(2048, 128, 100 + i, dev)
        embs.append(tower.embed(tok, None, head=head, seq_lens=np.full(2048, 128, np.int32)))
    P = torch.cat(embs)
    qtok = bench.synthetic_tokens(1000, 32, 999, dev)
    Qe = tower.embed(qtok, None, head=head, seq_lens=np.full(1000, 32, np.int32))
print("encoded: mean pairwise cos", torch.nn.functional.cosine_similarity(P[:1000], P[1000:2000]).mean().item(), "|p|", P.norm(dim=1).mean().item())
idx = FlatIPIndex(d, device=dev); idx.add(P)
report("encoded", idx, Qe)
Pc = P - P.mean(0, keepdim=True)
print("centred |p'| mean/max", Pc.norm(dim=1).mean().item(), Pc.norm(dim=1).max().item())
S = (Qe[:8] @ Pc.T)
print("score std per query", S.std(dim=1)[:4].tolist(), "top100 gap to 100th:", (S.topk(101, dim=1).values[:, 0] - S.topk(101, dim=1).values[:, 99])[:4].tolist())

print("---- ladder at 47104 encoded passages")
with torch.no_grad():
    embs = [P]
    for i in range(8, 23):
        tok = bench.synthetic_tokens(2048, 128, 100 + i, dev)
        embs.append(tower.embed(tok, None, head=head, seq_lens=np.full(2048, 128, np.int32)))
    P = torch.cat(embs)
idx = FlatIPIndex(d, device=dev); idx.add(P)
import convdr_amd.search as S
orig = idx._certify
def spy(qt, k_, D, I, status, tau_retry, x3):
    st0 = status.cpu().numpy().copy()
    bad = orig(qt, k_, D, I, status, tau_retry, x3)
    print("   _certify x3=%s in: %s -> still bad %d" % (x3, np.bincount(st0, minlength=4), len(bad)), idx.stats)
    return bad
idx._certify = spy
for rep in range(2):
    try:
        idx.search_tensors(Qe, k); print("rep", rep, "ok", idx.stats)
    except Exception as e:
        print("rep", rep, "ERR", str(e)[:90], idx.stats)
Dd, Id, st, tau = idx.search_device(Qe, k, x3=True)
em, band = idx.last_counts(Qe.shape[0], k)
st = st.cpu().numpy()
print("x3 one pass: status", np.bincount(st, minlength=4), "emitted max", em.max().item(), "band max", band.max().item())
badq = np.nonzero(st)[0][:5]
print("bad queries", badq, "emitted", em[badq].tolist(), "band", band[badq].tolist(), "tau_retry", tau[badq].tolist())
