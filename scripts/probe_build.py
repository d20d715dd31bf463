"""ad-hoc probe: time the GPU build + a search sweep at a given size (not part of the tests)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallel_hnsw_amd as ph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t = time.time()
store = ph.VectorStore.synthetic(n, dim)
print("store", time.time() - t, flush=True)
t = time.time()
bp = ph.BuildParameters(max_link_rounds=rounds)
h = ph.Hnsw.generate(store, np.arange(n), bp)
bt = time.time() - t
print("build s", bt, "vec/s", n / bt, "layers", [h._layer(l).node_count() for l in range(h.layer_count())], flush=True)
q = ph.VectorStore.synthetic(2000, dim, first=2 ** 32).read()
import torch
base = torch.from_numpy(store.read()).cuda() if n <= 2000000 else None
tq = torch.from_numpy(q).cuda()
gt = torch.topk(tq @ base.T, 10, dim=1).indices.cpu().numpy()
for ef, pd in [(32, 2), (64, 2), (128, 2), (300, 2), (128, 8), (128, 32), (300, 32), (512, 64)]:
    sp = ph.SearchParameters(ef, ef, pd)
    h.search_batch(queries=q[:64], sp=sp)
    t = time.time()
    ids, d, ln, st = h.search_batch(queries=q, sp=sp, stats=True)
    dt = time.time() - t
    rec = np.mean([len(set(ids[i, :10].tolist()) & set(gt[i].tolist())) / 10 for i in range(len(q))])
    print("ef", ef, "pd", pd, "recall@10 %.4f" % rec, "ndist %.0f" % st[:, 0].mean(), "hops %.0f" % st[:, 1].mean(),
          "host qps %.0f" % (len(q) / dt), "kernel ms %.2f" % h.kernel_ms(), flush=True)
