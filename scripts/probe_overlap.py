import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parallel_hnsw_amd as ph
n, dim, nq = 300000, 768, 10000
store = ph.VectorStore.clustered(n, dim)
h = ph.Hnsw.generate(store, np.arange(n), ph.BuildParameters(promote=0))
qs = ph.VectorStore.clustered(nq, dim, first=2 ** 32)
sp = ph.SearchParameters(128, 128, 8)
def bufs():
    return (torch.empty((nq, 128), dtype=torch.int32, device="cuda"), torch.empty((nq, 128), device="cuda"),
            torch.empty(nq, dtype=torch.int32, device="cuda"), torch.empty((nq, 2), dtype=torch.int32, device="cuda"),
            torch.empty(nq, dtype=torch.int32, device="cuda"))
def launch(b, stream=0):
    h.search_batch_device(nq, sp, b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), b[4].data_ptr(), queries=qs.rows_dev, ldq=dim, out_stats=b[3].data_ptr(), stream=stream)
ref = bufs(); launch(ref); torch.cuda.synchronize()
s = [torch.cuda.Stream(), torch.cuda.Stream()]
for trial in range(3):
    bs = [bufs() for _ in range(6)]
    for i, b in enumerate(bs):
        launch(b, s[i & 1].cuda_stream)
    torch.cuda.synchronize()
    for i, b in enumerate(bs):
        same = bool((b[0] == ref[0]).all() and (b[1] == ref[1]).all() and (b[3] == ref[3]).all())
        print("trial", trial, "launch", i, "identical", same, "status", int(b[4].sum()), "diff rows", int(((b[0] != ref[0]).any(1)).sum()))
