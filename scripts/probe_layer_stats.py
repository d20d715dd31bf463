"""ad-hoc probe: distance evaluations / hops per layer of the headline search (stats of searches cut at each layer)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallel_hnsw_amd as ph
n = 1000000
store = ph.VectorStore.clustered(n, 768, seed=42, first=0, n_clusters=1000, noise=1.0)
h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
q = ph.VectorStore.clustered(4000, 768, seed=42, first=2 ** 32, n_clusters=1000, noise=1.0).read()
for ef, pd in ((128, 8), (300, 2)):
    sp = ph.SearchParameters(ef, ef, pd)
    prev = np.zeros(2)
    for upto in range(1, h.layer_count() + 1):
        ids, d, ln, st = h.search_batch(queries=q, sp=sp, upto=upto, stats=True)
        cur = st.mean(0)
        print("ef %d pd %d layer %d (n=%d): dists %.0f hops %.1f  (cumulative %.0f / %.1f) results %.0f" % (
            ef, pd, upto - 1, h._layer(upto - 1).node_count(), cur[0] - prev[0], cur[1] - prev[1], cur[0], cur[1], ln.mean()), flush=True)
        prev = cur
