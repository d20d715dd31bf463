"""one-off: knn / threshold_nn (lib.rs:905-962) on random graphs and parameters, GPU vs oracle"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import oracle
import parallel_hnsw_amd as ph
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = []
for s in range(lo, hi):
    rng = np.random.default_rng(7000 + s)
    n = int(rng.integers(60, 3000)); dim = int(rng.choice([2, 5, 16, 33, 128])); metric = int(rng.integers(0, 3))
    dup = int(rng.choice([1, 1, 3]))
    rows = oracle.synth_rows(0, max(2, n // dup), dim, seed=s, normalize=(metric != 2))
    rows = np.repeat(rows, dup, axis=0)[:n].copy(); n = rows.shape[0]
    kw = dict(order=int(rng.choice([3, 6, 12])), neighborhood_size=int(rng.integers(2, 20)), seed=s, max_link_rounds=1)
    kw["zero_layer_neighborhood_size"] = int(rng.integers(kw["neighborhood_size"], 40))
    obp = oracle.default_build_params(**kw)
    oix = oracle.Index.generate(rows, np.arange(n), obp, dim=dim, metric=metric, sum_mode=oracle.SUM_BLOCKED64, threads=4)
    store = ph.VectorStore(rows[:, :dim], metric=metric)
    g = ph.Hnsw.from_layers(store, [oix.layer(l) for l in range(oix.layer_count)])
    k, pd = int(rng.integers(1, 40)), int(rng.choice([1, 2, 6]))
    ok = True
    ki, kd, kl = oix.knn(k, pd)
    for i, (v, got) in enumerate(g.knn(k, pd)):
        ok = ok and [x[0] for x in got] == [int(x) for x in ki[i, :int(kl[i])]]
        ok = ok and [np.float32(x[1]).view(np.uint32) for x in got] == [x.view(np.uint32) for x in kd[i, :int(kl[i])]]
    d0 = np.sort(kd[:, 0])
    thr = np.float32(d0[int(len(d0) * rng.uniform(0.2, 0.9))] * rng.uniform(1.0, 1.5) + 1e-6)
    depth = int(rng.choice([1, 2, 4, 16]))
    try:
        ti, td, tl = oix.threshold_nn(thr, pd, depth, max_out=1024)
    except RuntimeError:  # more than max_out results for some node: not a case for this comparison
        tl = np.array([1 << 20])
    if int(tl.max()) <= 512:   # the device queue doubles up to 1024 entries
        try:
            res = g.threshold_nn(float(thr), pd, depth, max_out=1024)
            for i, (v, got) in enumerate(res):
                ok = ok and [x[0] for x in got] == [int(x) for x in ti[i, :int(tl[i])]]
                ok = ok and [np.float32(x[1]).view(np.uint32) for x in got] == [x.view(np.uint32) for x in td[i, :int(tl[i])]]
        except ph.PhnswError as e:
            print("case", s, "threshold_nn refused:", str(e)[:100])
    if not ok:
        bad.append(s); print("case", s, (n, dim, metric, k, pd, float(thr), depth), "MISMATCH", flush=True)
    if (s - lo) % 20 == 19:
        print("done", s + 1 - lo, "cases in %.0f s, failures %s" % (time.time() - t0, bad), flush=True)
print("failures:", bad)
