"""ad-hoc probe: reconstruction error and exhaustive ADC recall of the PQ codebooks at full scale"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parallel_hnsw_amd as ph
n, nq = 1_000_000, 500
noise = 0.1 * 768 ** 0.5
store = ph.VectorStore.clustered(n, 768, seed=42, n_clusters=1000, noise=noise)
q = ph.VectorStore.clustered(nq, 768, seed=42, first=2 ** 32, n_clusters=1000, noise=noise)
base = torch.as_tensor(store.read()).cuda()
qt = torch.as_tensor(q.read()).cuda()
gt = torch.topk(qt @ base.T, 10, dim=1).indices
for it in (0, 8):
    pq = ph.PqStore(store, 96, 256, seed=0, kmeans_iters=it, kmeans_sample=65536)
    cb = torch.as_tensor(pq.codebook()).cuda()          # [m, ksub, dsub]
    codes = torch.as_tensor(pq.codes().astype(np.int64)).cuda()  # [n, m]
    rec = torch.cat([cb[j][codes[:, j]] for j in range(96)], dim=1)
    mse = float(((rec - base) ** 2).sum(1).mean())
    s = qt @ rec.T
    out = []
    for R in (10, 100, 512):
        top = torch.topk(s, R, dim=1).indices
        out.append(float((top[:, :, None] == gt[:, None, :]).any(1).float().mean()))
    print("iters", it, "mse %.4f |rec| %.3f" % (mse, float(rec.norm(dim=1).mean())), "exhaustive ADC recall@10 within top-10/100/512:", out, flush=True)
