"""experiment: one 10 000-query launch against the same batch cut into pieces that alternate between two streams
(the index's two workspaces alternate per call): does the table pass of piece k+1 hide under the search of piece k?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import parallel_hnsw_amd as ph
n, dim, nq, ef, pd = 1_000_000, 768, 10000, 256, 8
noise = 0.1 * 768 ** 0.5
store = ph.VectorStore.clustered(n, dim, seed=42, n_clusters=1000, noise=noise)
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
qs = ph.VectorStore.clustered(nq, dim, seed=42, first=2 ** 32, n_clusters=1000, noise=noise)
dev = torch.device("cuda", 0)
sp = ph.SearchParameters(ef, ef, pd)
ids = torch.empty((nq, ef), dtype=torch.int32, device=dev); d = torch.empty((nq, ef), dtype=torch.float32, device=dev)
ln = torch.empty(nq, dtype=torch.int32, device=dev); status = torch.empty(nq, dtype=torch.int32, device=dev)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
def run(pieces, steps=10):
    b = [nq * i // pieces for i in range(pieces + 1)]
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        for i in range(pieces):
            st = (s0, s1)[i & 1] if pieces > 1 else s0
            lo, cnt = b[i], b[i + 1] - b[i]
            h.search_batch_device(cnt, sp, ids.data_ptr() + lo * ef * 4, d.data_ptr() + lo * ef * 4, ln.data_ptr() + lo * 4,
                                  status.data_ptr() + lo * 4, queries=qs.rows_dev + lo * qs.ld * 4, ldq=qs.ld, stream=st.cuda_stream)
        if pieces > 1:  # a step ends when both streams are done (what a caller's stream would wait for)
            s0.wait_stream(s1); s1.wait_stream(s0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3
ref = None
for pieces in (1, 2, 3, 4, 6, 8, 1):
    run(pieces, 3)
    ms = run(pieces)
    print("pieces %d: %.3f ms per 10 000 queries (%.0f q/s)" % (pieces, ms, nq / ms * 1e3), flush=True)
