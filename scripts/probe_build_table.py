"""A/B inside one process: builds with and without the table kept across rounds (PHNSW_NO_BUILD_TABLE), alternating"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallel_hnsw_amd as ph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
kind = sys.argv[2] if len(sys.argv) > 2 else "survey"
noise = 0.1 * 768 ** 0.5 if kind == "survey" else 1.0
store = ph.VectorStore.clustered(n, 768, seed=42, first=0, n_clusters=1000, noise=noise)
ref = None
for rep in range(3):
    for off in (False, True):
        if off:
            os.environ["PHNSW_NO_BUILD_TABLE"] = "1"
        else:
            os.environ.pop("PHNSW_NO_BUILD_TABLE", None)
        t = time.time()
        h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
        dt = time.time() - t
        nb = h._layer(h.layer_count() - 1).neighbors
        same = True if ref is None else bool(np.array_equal(nb, ref))
        ref = nb if ref is None else ref
        print("%s  build %.3f s  (%.0f vectors/s)  identical %s" % ("no table " if off else "kept table", dt, n / dt, same), flush=True)
        del h
