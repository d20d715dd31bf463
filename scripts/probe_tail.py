"""experiment: how much of a 10 000-query launch is its tail?  isolated launches of several batch sizes (whole rounds of
the 3072 resident waves and not), q/s each"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import parallel_hnsw_amd as ph
n, dim, ef, pd = 1_000_000, 768, 256, 8
noise = 0.1 * 768 ** 0.5
store = ph.VectorStore.clustered(n, dim, seed=42, n_clusters=1000, noise=noise)
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
NQ = 32000
qs = ph.VectorStore.clustered(NQ, dim, seed=42, first=2 ** 32, n_clusters=1000, noise=noise)
dev = torch.device("cuda", 0)
sp = ph.SearchParameters(ef, ef, pd)
ids = torch.empty((NQ, ef), dtype=torch.int32, device=dev); d = torch.empty((NQ, ef), dtype=torch.float32, device=dev)
ln = torch.empty(NQ, dtype=torch.int32, device=dev); status = torch.empty(NQ, dtype=torch.int32, device=dev)
for nq in (3072, 6144, 9216, 10000, 12288, 15360, 18432, 20000, 24576, 30720, 32000):
    best = 1e9
    for _ in range(4):
        h.search_batch_device(nq, sp, ids.data_ptr(), d.data_ptr(), ln.data_ptr(), status.data_ptr(), queries=qs.rows_dev, ldq=qs.ld)
        torch.cuda.synchronize()
        best = min(best, h.kernel_ms())
    disp = h.dispatches()
    print("nq %5d: %.3f ms  %.0f q/s  (table %.3f, search %.3f)  waves/CU %s" % (nq, best, nq / best * 1e3, disp[0]["ms"], disp[1]["ms"],
          os.environ.get("PHNSW_WAVES_PER_CU", "default")), flush=True)
