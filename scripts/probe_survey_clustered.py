"""ad-hoc probe: SURVEY 8d's literal clustered variant (1000 centres, sigma = 0.1 per component
== noise norm 0.1*sqrt(768) = 2.77): build time, recall@10 / q/s over (ef, probe_depth)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parallel_hnsw_amd as ph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1 * 768 ** 0.5
nq = 10000
dev = torch.device("cuda", 0)
store = ph.VectorStore.clustered(n, 768, seed=42, n_clusters=1000, noise=noise)
q = ph.VectorStore.clustered(nq, 768, seed=42, first=2 ** 32, n_clusters=1000, noise=noise)
t = time.time()
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
torch.cuda.synchronize()
print("noise %.3f build %.1f s, self recall %.4f" % (noise, time.time() - t, h.stochastic_recall()), flush=True)
gi = torch.empty((nq, 10), dtype=torch.int32, device=dev)
gd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
store.bruteforce_topk_device(q.rows_dev, q.ld, nq, 10, gi.data_ptr(), gd.data_ptr())
gt = gi.to(torch.int64)
ids = torch.empty((nq, 1024), dtype=torch.int32, device=dev)
d = torch.empty((nq, 1024), dtype=torch.float32, device=dev)
ln = torch.empty(nq, dtype=torch.int32, device=dev)
st = torch.empty((nq, 2), dtype=torch.int32, device=dev)
status = torch.empty(nq, dtype=torch.int32, device=dev)
for ef, pd in [(32, 2), (64, 2), (128, 2), (128, 8), (256, 2), (256, 8), (512, 2), (512, 8), (512, 32), (1024, 8), (1024, 64)]:
    sp = ph.SearchParameters(ef, ef, pd)
    for _ in range(2):
        h.search_batch_device(nq, sp, ids.data_ptr(), d.data_ptr(), ln.data_ptr(), status.data_ptr(), queries=q.rows_dev,
                              ldq=q.ld, out_stats=st.data_ptr())
        torch.cuda.synchronize()
    r = ids.view(-1)[: nq * ef].view(nq, ef)[:, :10].to(torch.int64)
    rec = float(((r[:, :, None] == gt[:, None, :]).any(2).float().sum(1) / 10).mean())
    print("ef %4d pd %2d recall@10 %.4f ndist %.0f hops %.0f  %.2f ms  %.0f q/s" % (
        ef, pd, rec, st[:, 0].float().mean(), st[:, 1].float().mean(), h.kernel_ms(), nq / h.kernel_ms() * 1e3), flush=True)
