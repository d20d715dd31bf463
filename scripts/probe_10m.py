"""ad-hoc probe: BASELINE config 4 size on ONE GPU (10M x 768 f32 = 30.7 GB): capacity + timing"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parallel_hnsw_amd as ph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = 768
t = time.time(); store = ph.VectorStore.clustered(n, dim, n_clusters=10000); print("store s", time.time() - t, flush=True)
t = time.time(); h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters()); bt = time.time() - t
print("build s %.1f vec/s %.0f" % (bt, n / bt), "layers", [h._layer(l).node_count() for l in range(h.layer_count() - 1)], flush=True)
qs = ph.VectorStore.clustered(10000, dim, first=2 ** 32, n_clusters=10000)
class D:
    def __init__(s, p, shape): s.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (p, False), "version": 2, "strides": None}
base = torch.as_tensor(D(store.rows_dev, (n, dim)), device="cuda"); q = torch.as_tensor(D(qs.rows_dev, (10000, dim)), device="cuda")
bv = torch.full((2000, 10), -2.0, device="cuda"); bi = torch.zeros((2000, 10), dtype=torch.int64, device="cuda")
for bs in range(0, n, 500000):
    sc = q[:2000] @ base[bs:bs + 500000].T
    v, i = torch.topk(sc, 10, dim=1)
    av = torch.cat([bv, v], 1); ai = torch.cat([bi, i + bs], 1); tv, ti = torch.topk(av, 10, dim=1); bv = tv; bi = torch.gather(ai, 1, ti)
nq = 10000
ids = torch.empty((nq, 512), dtype=torch.int32, device="cuda"); d = torch.empty((nq, 512), device="cuda")
ln = torch.empty(nq, dtype=torch.int32, device="cuda"); st = torch.empty((nq, 2), dtype=torch.int32, device="cuda"); status = torch.empty(nq, dtype=torch.int32, device="cuda")
for ef, pd in [(128, 2), (128, 8), (300, 8)]:
    sp = ph.SearchParameters(ef, ef, pd)
    for _ in range(3):
        h.search_batch_device(nq, sp, ids.data_ptr(), d.data_ptr(), ln.data_ptr(), status.data_ptr(), queries=qs.rows_dev, ldq=dim, out_stats=st.data_ptr())
        torch.cuda.synchronize()
    r = ids.view(-1)[: nq * ef].view(nq, ef)[:2000, :10].to(torch.int64)
    rec = float((r[:, :, None] == bi[:, None, :]).any(2).float().mean())
    ms = h.kernel_ms()
    print("ef", ef, "pd", pd, "recall@10 %.4f" % rec, "qps %.0f" % (nq / ms * 1e3), "ndist %.0f" % st[:, 0].float().mean(), "status", int(status.sum()), flush=True)

# a 100 000-query batch: split, cell-ordered descent
nq2 = 100000
qs2 = ph.VectorStore.clustered(nq2, dim, first=2 ** 33, n_clusters=10000)
ids2 = torch.empty((nq2, 300), dtype=torch.int32, device="cuda"); d2 = torch.empty((nq2, 300), device="cuda")
ln2 = torch.empty(nq2, dtype=torch.int32, device="cuda"); st2 = torch.empty((nq2, 2), dtype=torch.int32, device="cuda"); status2 = torch.empty(nq2, dtype=torch.int32, device="cuda")
for ef, pd in [(128, 8), (300, 8)]:
    sp = ph.SearchParameters(ef, ef, pd)
    for _ in range(2):
        h.search_batch_device(nq2, sp, ids2.data_ptr(), d2.data_ptr(), ln2.data_ptr(), status2.data_ptr(), queries=qs2.rows_dev, ldq=dim, out_stats=st2.data_ptr())
        torch.cuda.synchronize()
    print("100k batch ef", ef, "pd", pd, "qps %.0f" % (nq2 / h.kernel_ms() * 1e3), "ndist %.0f" % st2[:, 0].float().mean(), "status", int(status2.sum()), flush=True)
