"""ad-hoc probe: one single-GPU build of the clustered 1M x 768 store (for rocprofv3 --kernel-trace --stats)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallel_hnsw_amd as ph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
store = ph.VectorStore.clustered(n, 768, seed=42, first=0, n_clusters=1000, noise=0.1 * 768 ** 0.5)
t = time.time()
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
print("build", time.time() - t, flush=True)
