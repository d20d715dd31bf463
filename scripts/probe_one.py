"""one configuration, three launches (for rocprofv3 passes): probe_one.py f32|pq EF PD [NQ]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parallel_hnsw_amd as ph
mode, ef, pd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
n = 1_000_000
dev = torch.device("cuda", 0)
noise = float(os.environ.get("PROBE_NOISE", 0.1 * 768 ** 0.5))
store = ph.VectorStore.clustered(n, 768, seed=42, n_clusters=1000, noise=noise)
q = ph.VectorStore.clustered(nq, 768, seed=42, first=2 ** 32, n_clusters=1000, noise=noise)
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
ids = torch.empty((nq, ef), dtype=torch.int32, device=dev); d = torch.empty((nq, ef), dtype=torch.float32, device=dev)
ln = torch.empty(nq, dtype=torch.int32, device=dev); st = torch.empty((nq, 2), dtype=torch.int32, device=dev)
status = torch.empty(nq, dtype=torch.int32, device=dev)
sp = ph.SearchParameters(ef, ef, pd)
if mode == "pq":
    qh = ph.QuantizedHnsw(256, store, m=96, graph=h)
    qh.store.set_table_mode("u8")
    h = qh.hnsw
for _ in range(3):
    h.search_batch_device(nq, sp, ids.data_ptr(), d.data_ptr(), ln.data_ptr(), status.data_ptr(), queries=q.rows_dev, ldq=q.ld,
                          out_stats=st.data_ptr())
    torch.cuda.synchronize()
print(mode, ef, pd, "kernel ms %.2f" % h.kernel_ms(), "hops %.0f ndist %.0f" % (st[:, 1].float().mean(), st[:, 0].float().mean()),
      [(x["layers"], round(x["ms"], 2)) for x in h.dispatches()], flush=True)
