"""ad-hoc probe: does a locality-ordered, XCD-segmented query schedule speed the search kernel up?
order = argsort(cluster id of the query) (known for the synthetic clustered generator)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import parallel_hnsw_amd as ph  # noqa: E402
from parallel_hnsw_amd._lib import lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dim = 768
M64 = (1 << 64) - 1


def mix64(x):
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & M64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & M64
    x ^= x >> 31
    return x


def cluster_of(first, count, seed=42, ncl=1000):
    out = np.empty(count, dtype=np.int64)
    for r in range(count):
        key = (seed + first + r) & M64
        out[r] = (mix64((key * 0xA24BAED4963EE407 + 0x9FB21C651E98DF25) & M64) * ncl) >> 64
    return out


store = ph.VectorStore.clustered(n, dim, seed=42, first=0, n_clusters=1000, noise=1.0)
t = time.time()
index = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
print("build %.2f s" % (time.time() - t), flush=True)
dev = torch.device("cuda", 0)
L = lib()
L.phnsw_debug_set_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]


def run(nq, sp, qids=None, qstore=None, order=None, reps=3):
    ef = sp.number_of_candidates
    ids = torch.empty((nq, ef), dtype=torch.int32, device=dev)
    d = torch.empty((nq, ef), dtype=torch.float32, device=dev)
    ln = torch.empty(nq, dtype=torch.int32, device=dev)
    st = torch.empty(nq, dtype=torch.int32, device=dev)
    stats = torch.empty((nq, 2), dtype=torch.int32, device=dev)
    L.phnsw_debug_set_order(index._h, C.c_void_p(order.data_ptr() if order is not None else None), nq if order is not None else 0)
    best = 1e9
    for _ in range(reps):
        if qids is not None:
            index.search_batch_device(nq, sp, ids.data_ptr(), d.data_ptr(), ln.data_ptr(), st.data_ptr(), qids=qids.data_ptr(),
                                      out_stats=stats.data_ptr())
        else:
            index.search_batch_device(nq, sp, ids.data_ptr(), d.data_ptr(), ln.data_ptr(), st.data_ptr(),
                                      queries=qstore.rows_dev, ldq=qstore.ld, out_stats=stats.data_ptr())
        torch.cuda.synchronize()
        best = min(best, index.kernel_ms())
    L.phnsw_debug_set_order(index._h, None, 0)
    return best, ids.clone(), float(stats[:, 0].float().mean())


# A: one link-round-like pass: every stored vector searches the whole stack (ef 300, probe 2)
cl = cluster_of(0, n)
print("clusters computed", flush=True)
sp = ph.SearchParameters(300, 300, 2)
nqa = min(n, 400000)
qids = torch.arange(nqa, dtype=torch.int32, device=dev)
ms0, ids0, nd = run(nqa, sp, qids=qids, reps=2)
order = torch.from_numpy(np.argsort(cl[:nqa], kind="stable").astype(np.int32)).to(dev)
ms1, ids1, _ = run(nqa, sp, qids=qids, order=order, reps=2)
ident = torch.arange(nqa, dtype=torch.int32, device=dev)
ms2, ids2, _ = run(nqa, sp, qids=qids, order=ident, reps=2)
pos = np.empty(n, dtype=np.uint32)
rc = L.phnsw_debug_layer_pos(C.c_void_p(index._h.value if hasattr(index._h, "value") else index._h), index.layer_count() - 1, pos.ctypes.data_as(C.c_void_p))
if rc == 0:
    o3 = torch.from_numpy(np.argsort(pos[:nqa], kind="stable").astype(np.int32)).to(dev)
    ms3, ids3, _ = run(nqa, sp, qids=qids, order=o3, reps=2)
    # how well does the hierarchical order keep clusters together?  distinct clusters per 512-query window
    oc = cl[:nqa][np.argsort(pos[:nqa], kind="stable")]
    win = [len(set(oc[i:i + 512].tolist())) for i in range(0, nqa - 512, 20000)]
    oc1 = cl[:nqa][np.argsort(cl[:nqa], kind="stable")]
    win1 = [len(set(oc1[i:i + 512].tolist())) for i in range(0, nqa - 512, 20000)]
    print("A hierarchical pos order %.1f ms; clusters per 512-query window: hierarchical %.1f, cluster-sorted %.1f" % (
        ms3, np.mean(win), np.mean(win1)), flush=True)
print("A stored queries x%d ef300 pd2: natural %.1f ms | cluster order + XCD segments %.1f ms | identity order + XCD segments %.1f ms | ndist %.0f | same results %s" % (
    nqa, ms0, ms1, ms2, nd, bool((ids0 == ids1).all() and (ids0 == ids2).all())), flush=True)
alg = nqa * nd * 768 * 4
print("   algorithmic %.1f GB -> %.2f TB/s natural, %.2f TB/s ordered" % (alg / 1e9, alg / ms0 / 1e9, alg / ms1 / 1e9), flush=True)

# B: bench-like batches of unstored queries, ef 128 probe 8
sp = ph.SearchParameters(128, 128, 8)
for nq in (10000, 100000):
    qs = ph.VectorStore.clustered(nq, dim, seed=42, first=2 ** 33, n_clusters=1000, noise=1.0)
    qc = cluster_of(2 ** 33, nq)
    ms0, ids0, nd = run(nq, sp, qstore=qs)
    order = torch.from_numpy(np.argsort(qc, kind="stable").astype(np.int32)).to(dev)
    ms1, ids1, _ = run(nq, sp, qstore=qs, order=order)
    # cluster-sorted but dealt round-robin over the segments (what a plain sort gives without XCD affinity)
    print("B unstored queries x%d ef128 pd8: natural %.2f ms (%.0f q/s) | cluster order + XCD segments %.2f ms (%.0f q/s) | ndist %.0f | same %s" % (
        nq, ms0, nq / ms0 * 1e3, ms1, nq / ms1 * 1e3, nd, bool((ids0 == ids1).all())), flush=True)
