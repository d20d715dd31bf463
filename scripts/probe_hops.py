"""debugging build (-DPH_HOP_PROFILE): where the hops of ONE query spend their time, f32 and PQ; PHNSW_LIB_PATH=.../libprof.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import parallel_hnsw_amd as ph
n, dim = 1_000_000, 768
noise = 0.1 * 768 ** 0.5
store = ph.VectorStore.clustered(n, dim, seed=42, n_clusters=1000, noise=noise)
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
q = ph.VectorStore.clustered(64, dim, seed=42, first=2 ** 32, n_clusters=1000, noise=noise).read()
print("==== f32 ef 256 pd 8, one query", flush=True)
for _ in range(2):
    h.search_batch(queries=q[:1], sp=ph.SearchParameters(256, 256, 8))
print("==== f32 per-hop path for every layer (PHNSW_NO_TINY)", flush=True)
os.environ["PHNSW_NO_TINY"] = "1"
h.search_batch(queries=q[:1], sp=ph.SearchParameters(256, 256, 8))
del os.environ["PHNSW_NO_TINY"]
qh = ph.QuantizedHnsw(256, store, m=96, graph=h)
qh.store.set_table_mode("u8")
print("==== pq ef 448 pd 8, one query", flush=True)
for _ in range(2):
    qh.hnsw.search_batch(queries=q[:1], sp=ph.SearchParameters(448, 448, 8))
