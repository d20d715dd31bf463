"""ad-hoc probe: time of the dense top-layer tile pass (ph_tiny_table_kernel) and of the table-id traversal"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parallel_hnsw_amd as ph

n, nq = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
ef, pd = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 8)
dev = torch.device("cuda", 0)
noise = 0.1 * 768 ** 0.5
store = ph.VectorStore.clustered(n, 768, seed=42, n_clusters=1000, noise=noise)
q = ph.VectorStore.clustered(nq, 768, seed=42, first=2 ** 32, n_clusters=1000, noise=noise)
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
ids = torch.empty((nq, ef), dtype=torch.int32, device=dev)
d = torch.empty((nq, ef), dtype=torch.float32, device=dev)
ln = torch.empty(nq, dtype=torch.int32, device=dev)
st = torch.empty((nq, 2), dtype=torch.int32, device=dev)
status = torch.empty(nq, dtype=torch.int32, device=dev)
sp = ph.SearchParameters(ef, ef, pd)
def run(tag):
    best = None
    for _ in range(3):
        h.search_batch_device(nq, sp, ids.data_ptr(), d.data_ptr(), ln.data_ptr(), status.data_ptr(), queries=q.rows_dev,
                              ldq=q.ld, out_stats=st.data_ptr())
        torch.cuda.synchronize()
        ds = h.dispatches()
        if best is None or h.kernel_ms() < best[0]:
            best = (h.kernel_ms(), ds)
    print(tag, "total %.2f ms | " % best[0] + " | ".join("%s %.2f ms (%d evals)" % (x["layers"], x["ms"], x["n_dist"]) for x in best[1]), flush=True)
for env in sys.argv[4:] or [""]:
    for kv in env.split(","):
        if kv:
            k, v = kv.split("=")
            os.environ[k] = v
    run(env or "default")
    for kv in env.split(","):
        if kv:
            del os.environ[kv.split("=")[0]]
