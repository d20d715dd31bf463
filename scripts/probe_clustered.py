"""ad-hoc probe: recall/QPS of the reference algorithm on clustered synthetic data"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parallel_hnsw_amd as ph

n = int(sys.argv[1]); dim = int(sys.argv[2]); nc = int(sys.argv[3]); noise = float(sys.argv[4])
g = torch.Generator(device="cuda"); g.manual_seed(1)
cent = torch.nn.functional.normalize(torch.rand(nc, dim, device="cuda", generator=g) * 2 - 1, dim=1)
def sample(m):
    k = torch.randint(0, nc, (m,), device="cuda", generator=g)
    nz = (torch.rand(m, dim, device="cuda", generator=g) * 2 - 1) * (noise * (3.0 / dim) ** 0.5)
    return torch.nn.functional.normalize(cent[k] + nz, dim=1).contiguous()
base = sample(n); q = sample(2000)
store = ph.VectorStore.from_device(base.data_ptr(), n, dim, dim, keepalive=base)
t = time.time()
h = ph.Hnsw.generate(store, np.arange(n), ph.BuildParameters())
print("build s", time.time() - t, flush=True)
gt = torch.topk(q @ base.T, 10, dim=1).indices.cpu().numpy()
qh = q.cpu().numpy()
for ef, pd in [(32, 2), (64, 2), (128, 2), (300, 2), (128, 8), (128, 32), (300, 32)]:
    sp = ph.SearchParameters(ef, ef, pd)
    ids, d, ln, st = h.search_batch(queries=qh, sp=sp, stats=True)
    rec = np.mean([len(set(ids[i, :10].tolist()) & set(gt[i].tolist())) / 10 for i in range(len(qh))])
    print("ef", ef, "pd", pd, "recall@10 %.4f" % rec, "ndist %.0f" % st[:, 0].mean(), "hops %.0f" % st[:, 1].mean(),
          "kernel ms %.2f" % h.kernel_ms(), flush=True)
