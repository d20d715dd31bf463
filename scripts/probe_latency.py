"""small-batch latency: where a 1 / 64 / 1024-query batch spends its time (per dispatch), round-1 dataset by default.
probe_latency.py [EF PD]   env PROBE_NOISE (1.0 = round 1's tight clusters, 2.77 = SURVEY 8d)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parallel_hnsw_amd as ph
ef = int(sys.argv[1]) if len(sys.argv) > 1 else 104
pd = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 1_000_000
dev = torch.device("cuda", 0)
noise = float(os.environ.get("PROBE_NOISE", 1.0))
store = ph.VectorStore.clustered(n, 768, seed=42, n_clusters=1000, noise=noise)
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
sp = ph.SearchParameters(ef, ef, pd)
for nq in (1, 64, 1024, 4096):
    q = ph.VectorStore.clustered(nq, 768, seed=42, first=2 ** 32, n_clusters=1000, noise=noise)
    ids = torch.empty((nq, ef), dtype=torch.int32, device=dev); d = torch.empty((nq, ef), dtype=torch.float32, device=dev)
    ln = torch.empty(nq, dtype=torch.int32, device=dev); st = torch.empty((nq, 2), dtype=torch.int32, device=dev)
    status = torch.empty(nq, dtype=torch.int32, device=dev)
    best = None
    for _ in range(6):
        h.search_batch_device(nq, sp, ids.data_ptr(), d.data_ptr(), ln.data_ptr(), status.data_ptr(), queries=q.rows_dev, ldq=q.ld,
                              out_stats=st.data_ptr())
        torch.cuda.synchronize()
        ms = h.kernel_ms()
        if best is None or ms < best[0]:
            best = (ms, [(x["layers"], round(x["ms"], 3)) for x in h.dispatches()])
    print("nq %5d ef %d pd %d  kernel ms %.3f  hops %.0f (max %d) ndist %.0f  %s" % (
        nq, ef, pd, best[0], st[:, 1].float().mean(), int(st[:, 1].max()), st[:, 0].float().mean(), best[1]), flush=True)
