"""the dense top-layer table pass alone, for rocprofv3: an 88 000 x 768 index (layers ... 611, 7 333, 88 000; the
7 333-node layer is tabulated at ef 256, like the 7 331-node one of the 1M headline index), 10 000 queries, three
launches.  probe_table.py [NQ]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parallel_hnsw_amd as ph
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n = 88_000
dev = torch.device("cuda", 0)
noise = 0.1 * 768 ** 0.5
store = ph.VectorStore.clustered(n, 768, seed=42, n_clusters=1000, noise=noise)
q = ph.VectorStore.clustered(nq, 768, seed=42, first=2 ** 32, n_clusters=1000, noise=noise)
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
ef = 256
ids = torch.empty((nq, ef), dtype=torch.int32, device=dev); d = torch.empty((nq, ef), dtype=torch.float32, device=dev)
ln = torch.empty(nq, dtype=torch.int32, device=dev); st = torch.empty((nq, 2), dtype=torch.int32, device=dev)
status = torch.empty(nq, dtype=torch.int32, device=dev)
sp = ph.SearchParameters(ef, ef, 8)
for _ in range(3):
    h.search_batch_device(nq, sp, ids.data_ptr(), d.data_ptr(), ln.data_ptr(), status.data_ptr(), queries=q.rows_dev, ldq=q.ld,
                          out_stats=st.data_ptr())
    torch.cuda.synchronize()
t_layers, t_nodes, t_mfma = h.dense_top_layers(ef)
disp = h.dispatches()
flop = 2.0 * 768 * t_nodes * nq
print("table: %d layers, %d nodes, mfma %s; pass %.3f ms = %.1f TFLOP/s incl. prep/pack; ids checksum %d" % (
    t_layers, t_nodes, t_mfma, disp[0]["ms"], flop / disp[0]["ms"] / 1e9, int(ids.to(torch.int64).sum())), flush=True)
