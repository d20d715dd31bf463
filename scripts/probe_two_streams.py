"""10 000-query steps of the headline workload: K steps on one stream against two batches alternating over two streams,
with the descent as one launch (default below 32 768 queries) and split (dense layers / layer 4 / layer 5 in launches
of their own: PHNSW_TWO_LAUNCH_MIN=1000).  probe_two_streams.py [STEPS]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parallel_hnsw_amd as ph
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, nq, ef = 1_000_000, 10000, 256
dev = torch.device("cuda", 0)
noise = 0.1 * 768 ** 0.5
store = ph.VectorStore.clustered(n, 768, seed=42, n_clusters=1000, noise=noise)
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
sp = ph.SearchParameters(ef, ef, 8)


class Lane:
    def __init__(self, first, stream):
        self.q = ph.VectorStore.clustered(nq, 768, seed=42, first=first, n_clusters=1000, noise=noise)
        self.ids = torch.empty((nq, ef), dtype=torch.int32, device=dev)
        self.d = torch.empty((nq, ef), dtype=torch.float32, device=dev)
        self.ln = torch.empty(nq, dtype=torch.int32, device=dev)
        self.status = torch.empty(nq, dtype=torch.int32, device=dev)
        self.stream = stream

    def launch(self):
        h.search_batch_device(nq, sp, self.ids.data_ptr(), self.d.data_ptr(), self.ln.data_ptr(), self.status.data_ptr(),
                              queries=self.q.rows_dev, ldq=self.q.ld, stream=self.stream)


s0 = torch.cuda.current_stream().cuda_stream
s2 = torch.cuda.Stream(device=dev)
a, b = Lane(2 ** 32, s0), Lane(2 ** 33, s2.cuda_stream)
ref = None
for split in (False, True):
    if split:
        os.environ["PHNSW_TWO_LAUNCH_MIN"] = "1000"
    else:
        os.environ.pop("PHNSW_TWO_LAUNCH_MIN", None)
    for lanes, name in (((a, a), "one stream"), ((a, b), "two streams")):
        for i in range(4):
            lanes[i & 1].launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            lanes[i & 1].launch()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        if ref is None:
            ref = a.ids.clone()
        same = bool((a.ids == ref).all()) and int(a.status.abs().sum()) == 0 and int(b.status.abs().sum()) == 0
        print("descent %s, %s: %.3f ms per step = %.0f q/s (results unchanged: %s, launches per descent %d)" % (
            "split" if split else "one launch", name, ms, nq / ms * 1e3, same, len(h.dispatches())), flush=True)
