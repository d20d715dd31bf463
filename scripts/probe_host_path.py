"""ad-hoc probe: the host-pointer entry point (phnsw_search_batch: H2D queries, search, D2H results, u32->u64)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallel_hnsw_amd as ph
n = 1000000
store = ph.VectorStore.clustered(n, 768, seed=42)
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
sp = ph.SearchParameters(104, 104, 8)
for nq in (10000, 100000):
    q = ph.VectorStore.clustered(nq, 768, seed=42, first=2 ** 32).read()
    h.search_batch(queries=q[:1000], sp=sp)
    for _ in range(2):
        t = time.time(); ids, d, ln = h.search_batch(queries=q, sp=sp); dt = time.time() - t
    print("host path nq %d: %.1f ms total (%.0f q/s), kernel %.1f ms" % (nq, dt * 1e3, nq / dt, h.kernel_ms()), flush=True)
