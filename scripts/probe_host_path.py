"""the drop-in host path (phnsw_search_batch / _topk: host queries in, u64 ids out) against the device-resident
launch of the same batch, headline configuration: probe_host_path.py [EF PD]; env PHNSW_HOST_CHUNKS to tune"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import parallel_hnsw_amd as ph
ef = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pd = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n, dim = 1_000_000, 768
noise = float(os.environ.get("PROBE_NOISE", 0.1 * 768 ** 0.5))
store = ph.VectorStore.clustered(n, dim, seed=42, n_clusters=1000, noise=noise)
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
sp = ph.SearchParameters(ef, ef, pd)
dev = torch.device("cuda", 0)
qs = ph.VectorStore.clustered(100000, dim, seed=42, first=2 ** 32, n_clusters=1000, noise=noise)
qall = qs.read()
for nq in (1, 64, 1024, 10000, 100000):
    q = np.ascontiguousarray(qall[:nq])
    ids = torch.empty((nq, ef), dtype=torch.int32, device=dev); d = torch.empty((nq, ef), dtype=torch.float32, device=dev)
    ln = torch.empty(nq, dtype=torch.int32, device=dev); status = torch.empty(nq, dtype=torch.int32, device=dev)
    best_dev = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        h.search_batch_device(nq, sp, ids.data_ptr(), d.data_ptr(), ln.data_ptr(), status.data_ptr(), queries=qs.rows_dev, ldq=qs.ld)
        torch.cuda.synchronize(); best_dev = min(best_dev, time.perf_counter() - t)
    kms = h.kernel_ms()
    out = {}
    for label, k in (("top10", 10), ("whole_queue", None)):
        best = 1e9
        for _ in range(5 if nq <= 10000 else 2):
            t = time.perf_counter(); r = h.search_batch(queries=q, sp=sp, k=k); best = min(best, time.perf_counter() - t)
        out[label] = best
    print("nq %6d  device-resident %.3f ms (kernel %.3f)  host top-10 %.3f ms (%.2fx, %.0f q/s)  host whole queue %.3f ms (%.2fx)" % (
        nq, best_dev * 1e3, kms, out["top10"] * 1e3, out["top10"] / best_dev, nq / out["top10"], out["whole_queue"] * 1e3,
        out["whole_queue"] / best_dev), flush=True)
