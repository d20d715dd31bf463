"""ad-hoc probe: quality of the per-layer locality positions on the clustered set"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallel_hnsw_amd as ph
from parallel_hnsw_amd._lib import lib
from probe_locality_util import cluster_of
n = 1000000
store = ph.VectorStore.clustered(n, 768, seed=42, first=0, n_clusters=1000, noise=1.0)
index = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
cl = cluster_of(0, n)
L = lib()
L.phnsw_debug_layer_pos.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
for l in range(index.layer_count()):
    lay = index._layer(l)
    nodes = np.asarray(lay.nodes, dtype=np.int64)
    pos = np.empty(len(nodes), dtype=np.uint32)
    rc = L.phnsw_debug_layer_pos(index._h, l, pos.ctypes.data_as(C.c_void_p))
    if rc:
        print("layer", l, len(nodes), "no pos"); continue
    o = np.argsort(pos, kind="stable")
    oc = cl[nodes[o]]
    runs = 1 + int((oc[1:] != oc[:-1]).sum())
    win = [len(set(oc[i:i + 512].tolist())) for i in range(0, len(oc) - 512, max(1, len(oc) // 50))]
    print("layer", l, len(nodes), "pos range", pos.min(), pos.max(), "distinct", len(np.unique(pos)), "cluster runs", runs,
          "mean run %.1f" % (len(oc) / runs), "clusters/512 window %.1f" % np.mean(win), flush=True)
