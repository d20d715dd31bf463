"""ad-hoc probe: PQ (m=96, 8-bit) on the SURVEY clustered set: k-means iterations vs recall@10 / q/s"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parallel_hnsw_amd as ph

n, nq = 1_000_000, 10_000
iters = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 8]
dev = torch.device("cuda", 0)
noise = 0.1 * 768 ** 0.5
store = ph.VectorStore.clustered(n, 768, seed=42, n_clusters=1000, noise=noise)
q = ph.VectorStore.clustered(nq, 768, seed=42, first=2 ** 32, n_clusters=1000, noise=noise)
gi = torch.empty((nq, 10), dtype=torch.int32, device=dev); gd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
store.bruteforce_topk_device(q.rows_dev, q.ld, nq, 10, gi.data_ptr(), gd.data_ptr())
gt = gi.to(torch.int64)
ids = torch.empty((nq, 1024), dtype=torch.int32, device=dev); d = torch.empty((nq, 1024), dtype=torch.float32, device=dev)
ln = torch.empty(nq, dtype=torch.int32, device=dev); st = torch.empty((nq, 2), dtype=torch.int32, device=dev)
status = torch.empty(nq, dtype=torch.int32, device=dev)
fgraph = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters()) if os.environ.get("PQ_F32_GRAPH") else None
for it in iters:
    t0 = time.time()
    qh = ph.QuantizedHnsw(256, store, ph.BuildParameters(promote=0), m=int(os.environ.get("PQ_M", "96")), kmeans_iters=it, kmeans_sample=65536, graph=fgraph)
    torch.cuda.synchronize()
    print("kmeans_iters %d: codebooks+codes+graph %.1f s" % (it, time.time() - t0), flush=True)
    for mode in ("u8",) + (("f32",) if len(sys.argv) > 2 else ()):
        qh.store.set_table_mode(mode)
        for ef, pd in [(128, 8), (192, 8), (256, 8), (384, 8), (512, 8)]:
            sp = ph.SearchParameters(ef, ef, pd)
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                qh.search_batch_device(nq, sp, q.rows_dev, q.ld, ids.data_ptr(), d.data_ptr(), ln.data_ptr(), status.data_ptr(), st.data_ptr())
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            r = ids.view(-1)[: nq * ef].view(nq, ef)[:, :10].to(torch.int64)
            rec = float(((r[:, :, None] == gt[:, None, :]).any(2).float().sum(1) / 10).mean())
            print("  %s ef %4d pd %2d recall@10 %.4f ndist %.0f hops %.0f  %.0f q/s (search kernel %.2f ms)" % (
                mode, ef, pd, rec, st[:, 0].float().mean(), st[:, 1].float().mean(), nq / dt, qh.hnsw.kernel_ms()), flush=True)
    del qh
