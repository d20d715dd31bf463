"""one-off: the locality schedule (cells, ordered lists, split descents) against the plain schedule on
random mid-size shapes: graphs and large-batch search results must be bit-identical"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import parallel_hnsw_amd as ph
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = []
for s in range(lo, hi):
    rng = np.random.default_rng(9000 + s)
    n = int(rng.integers(66_000, 260_000))
    dim = int(rng.choice([8, 24, 64, 100, 256, 768]))
    metric = int(rng.choice([0, 0, 1, 2]))
    clustered = bool(rng.integers(0, 2)) and metric != 2
    order = int(rng.choice([6, 12, 30]))
    nq = int(rng.integers(33_000, 70_000))
    sp = ph.SearchParameters(int(rng.choice([4, 16, 40, 130])), int(rng.choice([1, 8, 40])), int(rng.choice([1, 2, 5])))
    kw = dict(seed=s, n_clusters=int(rng.choice([50, 400, 3000]))) if clustered else dict(seed=s)
    mk = ph.VectorStore.clustered if clustered else ph.VectorStore.synthetic
    store = mk(n, dim, metric=metric, **kw)
    q = mk(nq, dim, metric=metric, first=2 ** 33, **kw).read()
    bp = ph.BuildParameters(order=order, max_link_rounds=1, seed=s, promote=int(rng.integers(0, 2)))
    bp.optimization.search.number_of_candidates = bp.optimization.search.upper_layer_candidate_count = 40
    h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), bp)
    a = h.search_batch(queries=q, sp=sp, stats=True)
    qi = rng.integers(0, n, nq).astype(np.uint64)
    a2 = h.search_batch(qids=qi, exclude=qi, sp=sp, stats=True)
    os.environ["PHNSW_NO_LOCALITY"] = "1"
    h2 = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), bp)
    b = h.search_batch(queries=q, sp=sp, stats=True)
    b2 = h.search_batch(qids=qi, exclude=qi, sp=sp, stats=True)
    del os.environ["PHNSW_NO_LOCALITY"]
    ok = h.layer_count() == h2.layer_count()
    for x, y in zip(h.layers, h2.layers):
        ok = ok and np.array_equal(x.nodes, y.nodes) and np.array_equal(x.neighbors, y.neighbors)
    for u, v in list(zip(a, b)) + list(zip(a2, b2)):
        ok = ok and np.array_equal(u.view(np.uint32) if u.dtype == np.float32 else u, v.view(np.uint32) if v.dtype == np.float32 else v)
    if not ok:
        bad.append(s)
    print("case %d n=%d dim=%d metric=%d clustered=%s nq=%d layers=%s %s  (%.0f s)" % (
        s, n, dim, metric, clustered, nq, [l.node_count() for l in h.layers], "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    del h, h2, store
print("failures:", bad)
