"""one-off: random shapes through the matrix-core distance table (csrc/tiny.hip): dims 256 / 768 / 1536, both
dot-product metrics, random index and batch sizes (partial 64 x 64 tiles on both sides, chunked tables), random
queue sizes -- the search results must equal those of the vector-unit table and of the per-hop path bit for bit"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import parallel_hnsw_amd as ph

lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = []


def run(h, q, qid, sp):
    a = h.search_batch(queries=q, sp=sp, stats=True)
    b = h.search_batch(qids=qid, sp=sp, exclude=qid, stats=True)
    return a + b


def same(x, y):
    for u, v in zip(x, y):
        if u.dtype == np.float32:
            u, v = u.view(np.uint32), v.view(np.uint32)
        if not np.array_equal(u, v):
            return False
    return True


for s in range(lo, hi):
    rng = np.random.default_rng(9000 + s)
    dim = int(rng.choice([256, 768, 768, 1536]))
    n = int(rng.integers(300, 14000 if dim < 1536 else 6000))
    metric = int(rng.choice([ph.METRIC_COSINE_HALF, ph.METRIC_ONE_MINUS_DOT]))
    nq = int(rng.choice([32, 33, 64, 65, 100, 1000, 2049, 4000]))
    ef = int(rng.choice([8, 40, 104, 129, 256, 300, 600]))
    store = ph.VectorStore.synthetic(n, dim, seed=s, metric=metric)
    h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters(seed=s, max_link_rounds=1, order=int(rng.choice([6, 12]))))
    q = ph.VectorStore.synthetic(nq, dim, seed=s + 77, first=2 ** 32, metric=metric).read()
    qid = rng.integers(0, n, nq).astype(np.uint64)
    sp = ph.SearchParameters(ef, ef, int(rng.choice([2, 5])))
    layers, nodes, mfma = h.dense_top_layers(ef)
    mc = run(h, q, qid, sp)
    os.environ["PHNSW_TINY_VALU"] = "1"
    vu = run(h, q, qid, sp)
    del os.environ["PHNSW_TINY_VALU"]
    os.environ["PHNSW_NO_TINY"] = "1"
    hop = run(h, q, qid, sp)
    del os.environ["PHNSW_NO_TINY"]
    ok = same(mc, vu) and same(mc, hop)
    if not ok:
        bad.append(s)
    print("case %d n=%d dim=%d metric=%d nq=%d ef=%d dense layers %d (%d nodes, matrix cores %s) %s" % (
        s, n, dim, metric, nq, ef, layers, nodes, mfma, "ok" if ok else "MISMATCH"), flush=True)
print("done %d cases in %.0f s" % (hi - lo, time.time() - t0))
print("failures:", bad)
