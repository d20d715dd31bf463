"""does the host path's chunk pipeline depend on how many streams the process made before it?  (HIP maps streams onto a
few hardware queues: GPU_MAX_HW_QUEUES, 4 by default.)  diag_hostpath.py N_DUMMY_STREAMS"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import parallel_hnsw_amd as ph
k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n, nq, ef = 200_000, 10000, 256
dev = torch.device("cuda", 0)
noise = 0.1 * 768 ** 0.5
store = ph.VectorStore.clustered(n, 768, seed=42, n_clusters=1000, noise=noise)
qs = ph.VectorStore.clustered(nq, 768, seed=42, first=2 ** 32, n_clusters=1000, noise=noise)
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
sp = ph.SearchParameters(ef, ef, 8)
q = np.ascontiguousarray(qs.read())
dummies = [torch.cuda.Stream(device=dev) for _ in range(k)]
for s in dummies:
    with torch.cuda.stream(s):
        torch.zeros(16, device=dev)
torch.cuda.synchronize()
best = 1e9
for _ in range(6):
    t0 = time.perf_counter(); h.search_batch(queries=q, sp=sp, k=10); best = min(best, time.perf_counter() - t0)
print("%d streams made before the first host call (GPU_MAX_HW_QUEUES=%s): host top-10 of 10000: %.3f ms" % (
    k, os.environ.get("GPU_MAX_HW_QUEUES", "default"), best * 1e3), flush=True)
