"""ad-hoc probe: does processing similar queries together (cache locality) pay?"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parallel_hnsw_amd as ph
M64 = (1 << 64) - 1
def mix64(x):
    x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & M64; x ^= x >> 27; x = (x * 0x94D049BB133111EB) & M64; x ^= x >> 31; return x
n, dim, nq = 1000000, 768, 10000
store = ph.VectorStore.clustered(n, dim)
h = ph.Hnsw.generate(store, np.arange(n), ph.BuildParameters(promote=0))
qs = ph.VectorStore.clustered(nq, dim, first=2 ** 32)
class D:
    def __init__(s, p, shape): s.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (p, False), "version": 2, "strides": None}
q = torch.as_tensor(D(qs.rows_dev, (nq, dim)), device="cuda")
cl = np.array([((mix64(((42 + 2 ** 32 + i) * 0xA24BAED4963EE407 + 0x9FB21C651E98DF25) & M64) * 1000) >> 64) for i in range(nq)])
perm = torch.from_numpy(np.argsort(cl, kind="stable")).cuda()
qsorted = q[perm].contiguous()
ids = torch.empty((nq, 128), dtype=torch.int32, device="cuda"); d = torch.empty((nq, 128), device="cuda")
ln = torch.empty(nq, dtype=torch.int32, device="cuda"); st = torch.empty((nq, 2), dtype=torch.int32, device="cuda"); status = torch.empty(nq, dtype=torch.int32, device="cuda")
sp = ph.SearchParameters(128, 128, 8)
for name, t in (("as generated", q), ("sorted by cluster", qsorted), ("as generated", q), ("sorted by cluster", qsorted)):
    ms = []
    for _ in range(6):
        h.search_batch_device(nq, sp, ids.data_ptr(), d.data_ptr(), ln.data_ptr(), status.data_ptr(), queries=t.data_ptr(), ldq=dim, out_stats=st.data_ptr())
        torch.cuda.synchronize(); ms.append(h.kernel_ms())
    print(name, "kernel ms %.2f" % np.mean(ms[1:]), "qps %.0f" % (nq / np.mean(ms[1:]) * 1e3), flush=True)
# link-round shaped: stored queries = all nodes, natural order vs random order of a 100k sample
qid = torch.arange(0, 200000, dtype=torch.int32, device="cuda")
rnd = qid[torch.randperm(200000, device="cuda")].contiguous()
sp3 = ph.SearchParameters(300, 300, 2)
ids3 = torch.empty((200000, 300), dtype=torch.int32, device="cuda"); d3 = torch.empty((200000, 300), device="cuda")
ln3 = torch.empty(200000, dtype=torch.int32, device="cuda"); st3 = torch.empty((200000, 2), dtype=torch.int32, device="cuda"); status3 = torch.empty(200000, dtype=torch.int32, device="cuda")
# order by cluster of the stored vectors
clb = np.array([((mix64(((42 + i) * 0xA24BAED4963EE407 + 0x9FB21C651E98DF25) & M64) * 1000) >> 64) for i in range(200000)])
byc = torch.from_numpy(np.argsort(clb, kind="stable").astype(np.int32)).cuda()
for name, t in (("stored 0..200k natural", qid), ("stored random order", rnd), ("stored sorted by cluster", byc)):
    ms = []
    for _ in range(3):
        h.search_batch_device(200000, sp3, ids3.data_ptr(), d3.data_ptr(), ln3.data_ptr(), status3.data_ptr(), qids=t.data_ptr(), exclude=t.data_ptr(), out_stats=st3.data_ptr())
        torch.cuda.synchronize(); ms.append(h.kernel_ms())
    print(name, "kernel ms %.1f" % np.mean(ms[1:]), "searches/s %.0f" % (200000 / np.mean(ms[1:]) * 1e3), flush=True)
