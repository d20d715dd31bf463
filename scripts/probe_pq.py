"""ad-hoc probe: BASELINE config 5 (1M x 768, PQ m=96, 8-bit codes) timings and recall"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parallel_hnsw_amd as ph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dim, m = 768, 96
full = ph.VectorStore.clustered(n, dim)
f16 = len(sys.argv) > 2 and sys.argv[2] == 'f16'
t = time.time(); qh = ph.QuantizedHnsw(256, full, ph.BuildParameters(promote=0), m=m, table_f16=f16); torch.cuda.synchronize()  # promotion off: DESIGN section 9
print("pq create+build s", time.time() - t, flush=True)
if len(sys.argv) > 2 and sys.argv[2] == 'u8':  # graph built with the f32 table, searched with 8-bit entries
    from parallel_hnsw_amd._lib import lib, check
    qh.store.set_table_mode("u8")
qs = ph.VectorStore.clustered(10000, dim, first=2 ** 32)
class D:
    def __init__(s, p, shape): s.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (p, False), "version": 2, "strides": None}
base = torch.as_tensor(D(full.rows_dev, (n, dim)), device="cuda"); q = torch.as_tensor(D(qs.rows_dev, (10000, dim)), device="cuda")
gt = torch.topk(q[:2000] @ base.T, 10, dim=1).indices
nq = 10000
ids = torch.empty((nq, 1024), dtype=torch.int32, device="cuda"); d = torch.empty((nq, 1024), device="cuda")
ln = torch.empty(nq, dtype=torch.int32, device="cuda"); st = torch.empty((nq, 2), dtype=torch.int32, device="cuda"); status = torch.empty(nq, dtype=torch.int32, device="cuda")
for ef, pd in [(64, 2), (128, 2), (128, 8), (300, 2), (300, 8), (512, 16)]:
    sp = ph.SearchParameters(ef, ef, pd)
    for _ in range(2):
        torch.cuda.synchronize(); t = time.time()
        qh.search_batch_device(nq, sp, qs.rows_dev, qs.ld, ids.data_ptr(), d.data_ptr(), ln.data_ptr(), status.data_ptr(), st.data_ptr())
        torch.cuda.synchronize(); dt = time.time() - t
    r = ids.view(-1)[: nq * ef].view(nq, ef)[:2000, :10].to(torch.int64)
    rec = float((r[:, :, None] == gt[:, None, :]).any(2).float().mean())
    print("ef", ef, "pd", pd, "recall@10 %.4f" % rec, "qps %.0f" % (nq / dt), "search kernel ms %.2f" % qh.hnsw.kernel_ms(),
          "ndist %.0f hops %.0f" % (st[:, 0].float().mean(), st[:, 1].float().mean()), flush=True)
