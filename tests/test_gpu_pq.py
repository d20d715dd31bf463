"""GPU parity for the product-quantised path (reference src/pq.rs; BASELINE config 5):
codebooks, codes, the graph built over the code rows, quantised search and the
full-precision re-rank, against the oracle's definition of the same flow (oracle/orc_quant.c).
The reference pins no numeric value here ("parity unpinned"); what is checked is
bit equality with the oracle and the recall the reference's tests assert."""
import os

import numpy as np
import pytest

import oracle
import parallel_hnsw_amd as ph

pytestmark = pytest.mark.gpu


def make(n, dim, m, ksub, seed=0, metric=0, clustered=False, table_f16=False):
    rows = (oracle.synth_clustered_rows(0, n, dim, n_clusters=20) if clustered else oracle.synth_rows(0, n, dim))
    full = ph.VectorStore(rows[:, :dim], metric=metric)
    pq = ph.PqStore(full, m, ksub, seed, table_f16=table_f16)
    ocodes, ocb = oracle.pq_create(rows, dim, m, ksub, seed)
    return rows, full, pq, ocodes, ocb


@pytest.mark.parametrize("n,dim,m,ksub,iters,sample", [(3000, 64, 16, 64, 5, 0), (5000, 96, 12, 256, 3, 2000),
                                                      (900, 32, 4, 16, 8, 500), (2000, 768, 96, 32, 2, 1000)])
def test_kmeans_codebooks_equal_the_oracle(n, dim, m, ksub, iters, sample):
    """per-sub-space k-means: f64 member sums in training order, exact nearest-centroid
    assignment -- codebook and codes bit for bit the oracle's, and a lower reconstruction error than
    random_centroids (pq.rs:261-285)"""
    rows = oracle.synth_clustered_rows(0, n, dim, n_clusters=25)
    full = ph.VectorStore(rows[:, :dim])
    pq = ph.PqStore(full, m, ksub, seed=7, kmeans_iters=iters, kmeans_sample=sample)
    ocodes, ocb = oracle.pq_create(rows, dim, m, ksub, seed=7, kmeans_iters=iters, sample=sample)
    np.testing.assert_array_equal(pq.codebook().view(np.uint32), ocb.view(np.uint32))
    np.testing.assert_array_equal(pq.codes(), ocodes)
    rnd = ph.PqStore(full, m, ksub, seed=7)

    def mse(st):
        return float(((st.reconstruct(st.codes()) - rows[:, :dim]) ** 2).sum(1).mean())
    assert mse(pq) < mse(rnd)


@pytest.mark.parametrize("metric", [0, 2])
def test_quantised_distance_batch_u8_table(metric):
    """table mode 2 against a numpy restatement of its definition: 8-bit entries
    rint((T - min_row) / scale), scale = widest row range / 255, distance = bias + scale * sum"""
    n, dim, m, ksub = 600, 64, 16, 128
    rows, full, pq, ocodes, ocb = make(n, dim, m, ksub, metric=metric, table_f16=2)
    ids = np.arange(n, dtype=np.uint64)
    q = oracle.synth_rows(2 ** 32, 1, dim)[0, :dim]
    f = np.float32
    for qq, got in ((q, pq.compare_vec(ph.Unstored(q), ids)), (None, pq.compare_vec(ph.Stored(5), ids))):
        T = np.zeros((m, ksub), dtype=np.float32)
        for j in range(m):
            sub = qq[j * (dim // m):(j + 1) * (dim // m)] if qq is not None else ocb[j, ocodes[5, j]]
            for k in range(ksub):
                acc = f(0)
                for e in range(dim // m):
                    if metric == 2:
                        df = f(sub[e] - ocb[j, k, e])
                        acc = f(np.float64(df) * np.float64(df) + np.float64(acc))
                    else:
                        acc = f(np.float64(sub[e]) * np.float64(ocb[j, k, e]) + np.float64(acc))
                T[j, k] = acc
        lo = T.min(axis=1)
        widest = f(max(f(T[j].max() - lo[j]) for j in range(m)))
        bias = f(0)
        for j in range(m):
            bias = f(bias + lo[j])
        scale = f(widest / f(255))
        U = np.rint((T - lo[:, None]) / scale).astype(np.int64)
        assert U.min() >= 0 and U.max() <= 255
        exp = np.empty(n, dtype=np.float32)
        for i in range(n):
            sm = int(sum(U[j, ocodes[i, j]] for j in range(m)))
            r = f(bias + f(scale * f(sm)))
            exp[i] = {0: f((f(1) - r) / f(2)), 2: f(np.sqrt(r))}[metric]
        np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32))


@pytest.mark.parametrize("n,dim,m,ksub", [(2000, 64, 8, 256), (1500, 96, 24, 64), (3000, 768, 96, 256)])
def test_codebook_and_codes_match_oracle(n, dim, m, ksub):
    rows, full, pq, ocodes, ocb = make(n, dim, m, ksub, seed=3)
    assert (pq.m, pq.ksub, pq.dsub) == (m, ksub, dim // m)
    np.testing.assert_array_equal(pq.codebook().view(np.uint32), ocb.view(np.uint32))
    np.testing.assert_array_equal(pq.codes(), ocodes)


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_quantised_distance_batch_bit_exact(metric):
    n, dim, m, ksub = 1000, 64, 16, 128
    rows, full, pq, ocodes, ocb = make(n, dim, m, ksub, metric=metric)
    oix = oracle.Index(rows, dim=dim, metric=metric)
    oix.set_pq(ocodes, ocb)
    ids = np.arange(n, dtype=np.uint64)
    q = oracle.synth_rows(2 ** 32, 1, dim)[0, :dim]
    # raw query against codes (asymmetric) and stored code against codes (symmetric)
    got = pq.compare_vec(ph.Unstored(q), ids)
    got_s = pq.compare_vec(ph.Stored(5), ids)
    # oracle: one-layer index over everything => exhaustive search returns every distance
    oix.push_layer(np.arange(n), np.full((n, 1), oracle.EMPTY, dtype=np.uint64), 1)
    import ctypes as C
    L = oracle.lib()
    # distances through the search API: entry only is not enough, so use brute force over a flat graph
    exp = np.empty(n, dtype=np.float32)
    exp_s = np.empty(n, dtype=np.float32)
    T = np.zeros((m, ksub), dtype=np.float32)
    for (qq, out) in ((q, exp), (None, exp_s)):
        for j in range(m):
            sub = qq[j * (dim // m):(j + 1) * (dim // m)] if qq is not None else ocb[j, ocodes[5, j]]
            for k in range(ksub):
                acc = np.float32(0)
                for e in range(dim // m):
                    if metric == 2:
                        df = np.float32(sub[e] - ocb[j, k, e])
                        acc = np.float32(np.float64(df) * np.float64(df) + np.float64(acc))
                    else:
                        acc = np.float32(np.float64(sub[e]) * np.float64(ocb[j, k, e]) + np.float64(acc))
                T[j, k] = acc
        for i in range(n):
            r = np.float32(0)
            for j in range(m):
                r = np.float32(r + T[j, ocodes[i, j]])
            out[i] = {0: np.float32((np.float32(1) - r) / np.float32(2)), 1: np.float32(np.float32(1) - r),
                      2: np.float32(np.sqrt(r))}[metric]
    np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32))
    np.testing.assert_array_equal(got_s.view(np.uint32), exp_s.view(np.uint32))


# f16 column = the table mode: 0/False f32, 1/True IEEE-half entries, 2 8-bit entries with a per-query scale
@pytest.mark.parametrize("n,dim,m,ksub,f16", [(2500, 64, 16, 256, False), (1500, 768, 96, 256, False),
                                              (1500, 768, 96, 256, True), (2000, 64, 16, 128, True),
                                              (1500, 768, 96, 256, 2), (2500, 64, 16, 256, 2), (1800, 96, 24, 100, 2)])
def test_pq_index_build_and_search_parity(n, dim, m, ksub, f16):
    # mode 2 (8-bit entries) is asymmetric, hence search-only: its graph is built in mode 0
    build_mode = 0 if f16 == 2 else f16
    rows, full, pq, ocodes, ocb = make(n, dim, m, ksub, seed=1, clustered=True, table_f16=build_mode)
    bp_kw = dict(seed=2, promote=0)
    # oracle: Hnsw::generate over the code rows (QuantizedHnsw::new pq.rs:337-338)
    oix = oracle.Index(rows, dim=dim, sum_mode=oracle.SUM_BLOCKED64)
    oix.set_pq(ocodes, ocb, table_f16=build_mode)
    obp = oracle.default_build_params(**bp_kw)
    vs = oracle.shuffle(np.arange(n), obp.seed)
    sizes = oracle.calculate_partitions(n, obp.order)
    for i, sz in enumerate(sizes):
        oix.generate_layer(vs[:sz], 48 if i == len(sizes) - 1 else 24, obp)
        oix.improve_index(obp)
    g = ph.Hnsw.generate(pq, np.arange(n), ph.BuildParameters(**bp_kw))
    assert g.layer_count() == oix.layer_count
    for l in range(oix.layer_count):
        nodes, nb = oix.layer(l)
        np.testing.assert_array_equal(g._layer(l).neighbors, nb, err_msg="layer %d" % l)
    if f16 == 2:
        with pytest.raises(ph.PhnswError):  # build entry points refuse the asymmetric mode
            pq.set_table_mode("u8")
            ph.Hnsw.generate(pq, np.arange(n), ph.BuildParameters(**bp_kw))
        oracle.lib().orc_index_set_pq_table_f16(oix.h, 2)
        gs_, cs_ = g.search_batch(qids=np.arange(50), sp=ph.SearchParameters(32, 32, 2)), oix.search(qids=np.arange(50), sp=(32, 32, 2))
        np.testing.assert_array_equal(gs_[0], cs_[0])   # Stored queries in the search-only mode
        np.testing.assert_array_equal(gs_[1].view(np.uint32), cs_[1].view(np.uint32))
    # quantised search, raw query (ADC) -- ids, distances, counters
    q = oracle.synth_clustered_rows(2 ** 32, 200, dim, n_clusters=20)[:, :dim]
    sp = (64, 64, 2)
    gi, gd, gl, gs = g.search_batch(queries=q, sp=ph.SearchParameters(*sp), stats=True)
    ci, cd, cl, cs = oix.search(queries=q, sp=sp, stats=True)
    np.testing.assert_array_equal(gi, ci)
    np.testing.assert_array_equal(gd.view(np.uint32), cd.view(np.uint32))
    np.testing.assert_array_equal(gs, cs)
    # QuantizedHnsw::search: re-ranked with the full store, sorted (d, id)
    ofull = oracle.Index(rows, dim=dim, sum_mode=oracle.SUM_BLOCKED64)
    ofull.set_sum_mode(oracle.SUM_BLOCKED64)
    qh = ph.QuantizedHnsw.__new__(ph.QuantizedHnsw)
    qh.full, qh.store, qh.hnsw = full, pq, g
    for quant in (False, True):
        ri, rd, rl = qh.search_batch(q, ph.SearchParameters(*sp), quantize_query=quant)
        oi, od, ol = oix.pq_search(ofull, q, sp, quantize_query=quant)
        np.testing.assert_array_equal(rl, ol)
        np.testing.assert_array_equal(ri, oi)
        np.testing.assert_array_equal(rd.view(np.uint32), od.view(np.uint32))
        assert (np.diff(rd[:, :int(rl.min())], axis=1) >= 0).all()


def test_pq_recall_like_reference_test():
    """test_pq_recall (pq.rs:955-978) in spirit: stored vectors find themselves through the
    quantised index + full-precision re-rank"""
    n, dim = 5000, 256
    rows = oracle.synth_rows(0, n, dim)
    full = ph.VectorStore(rows[:, :dim])
    qh = ph.QuantizedHnsw(256, full, ph.BuildParameters(), m=32, seed=0)
    ids, d, ln = qh.search_batch(rows[:500, :dim], ph.SearchParameters(300, 300, 2))
    recall = np.mean(ids[:, 0] == np.arange(500))
    assert recall >= 0.9, recall
    assert np.abs(d[ids[:, 0] == np.arange(500), 0]).max() < 1e-5


def test_pq_rejects_bad_shapes():
    rows = oracle.synth_rows(0, 100, 48)
    full = ph.VectorStore(rows[:, :48])
    for m, ksub in [(5, 16), (7, 16), (8, 300), (8, 0), (8, 101)]:
        with pytest.raises(ph.PhnswError):
            ph.PqStore(full, m, ksub)
    with pytest.raises(ph.PhnswError):
        ph.PqStore(ph.PqStore(full, 8, 16), 8, 16)  # a PQ store cannot be quantised again


def test_quantize_and_reconstruct_arbitrary_vectors():
    """Quantizer::quantize / reconstruct (pq.rs:61-81) for vectors that are not in the store"""
    n, dim, m, ksub = 1200, 96, 24, 64
    rows, full, pq, ocodes, ocb = make(n, dim, m, ksub, seed=4)
    fresh = oracle.synth_rows(2 ** 33, 500, dim)
    codes = pq.quantize(fresh[:, :dim])
    np.testing.assert_array_equal(codes, oracle.pq_encode(fresh, dim, ocb))
    rec = pq.reconstruct(codes)
    want = np.concatenate([ocb[j, codes[:, j]] for j in range(m)], axis=1)
    np.testing.assert_array_equal(rec.view(np.uint32), want.view(np.uint32))
    np.testing.assert_array_equal(pq.quantize(rows[:50, :dim]), ocodes[:50])  # stored rows: their own codes


@pytest.mark.parametrize("m,ef,pd", [(96, 64, 2), (96, 300, 4), (32, 128, 3), (64, 40, 2), (128, 100, 2)])
def test_register_table_search_equals_the_other_table_placements_and_the_oracle(m, ef, pd):
    """8-bit tables of 32 / 64 / 96 / 128 sub-spaces x 256 centroids are held in VGPRs and looked up with
    ds_bpermute (DistPQR); LDS and global placements (PHNSW_PQ_TABLE) and the oracle must give the same bits"""
    n, dim = 6000, 768 if m == 96 else 4 * m
    rows = oracle.synth_clustered_rows(0, n, dim, n_clusters=30)
    full = ph.VectorStore(rows[:, :dim])
    qh = ph.QuantizedHnsw(256, full, ph.BuildParameters(promote=0, max_link_rounds=1, seed=2), m=m, kmeans_iters=2, kmeans_sample=3000)
    qh.store.set_table_mode("u8")
    q = oracle.synth_clustered_rows(2 ** 32, 200, dim, n_clusters=30)[:, :dim]
    sp = ph.SearchParameters(ef, ef, pd)
    reg = qh.hnsw.search_batch(queries=q, sp=sp, stats=True)
    for placement in ("lds", "global"):
        os.environ["PHNSW_PQ_TABLE"] = placement
        try:
            other = qh.hnsw.search_batch(queries=q, sp=sp, stats=True)
        finally:
            del os.environ["PHNSW_PQ_TABLE"]
        for a, b in zip(reg, other):
            np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                                          b.view(np.uint32) if b.dtype == np.float32 else b)
    # Stored queries (their table is that of the reconstruction)
    qid = np.arange(0, n, 11, dtype=np.uint64)
    r2 = qh.hnsw.search_batch(qids=qid, sp=sp, exclude=qid, stats=True)
    os.environ["PHNSW_PQ_TABLE"] = "global"
    try:
        o2 = qh.hnsw.search_batch(qids=qid, sp=sp, exclude=qid, stats=True)
    finally:
        del os.environ["PHNSW_PQ_TABLE"]
    for a, b in zip(r2, o2):
        np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                                      b.view(np.uint32) if b.dtype == np.float32 else b)
    # the oracle on the same codes / codebook / graph
    ocodes, ocb = oracle.pq_create(rows, dim, m, 256, seed=0, kmeans_iters=2, sample=3000)
    np.testing.assert_array_equal(qh.store.codes(), ocodes)
    oix = oracle.Index(rows, dim=dim, metric=oracle.METRIC_COSINE_HALF, sum_mode=oracle.SUM_BLOCKED64)
    for l in qh.hnsw.layers:
        oix.push_layer(l.nodes, l.neighbors, l.neighborhood_size)
    oix.set_pq(ocodes, ocb, table_f16=2)
    ci, cd, cl, cs = oix.search(queries=q, sp=(ef, ef, pd), stats=True)
    np.testing.assert_array_equal(reg[0], ci)
    np.testing.assert_array_equal(reg[1].view(np.uint32), cd.view(np.uint32))
    np.testing.assert_array_equal(reg[3], cs)


def test_reference_shaped_quantizer():
    """pq.rs's own shape (pq.rs:920-978 scaled down): one shared codebook, u16 codes, HNSW quantizer.
    * codes: the HNSW over the centroids finds (almost always) the exact nearest centroid;
    * distances over the code rows == DistF32 over the materialised reconstructions, bit for bit, so a search of
      the adopted graph over the codes equals the search over the reconstruction store;
    * QuantizedHnsw::search (quantised query, re-rank, sort) finds the true neighbours."""
    n, dim, cs, C_ = 4000, 128, 16, 500
    rows = oracle.synth_rows(0, n, dim)  # random_normed_vec, like the reference's test (pq.rs:959-968)
    full = ph.VectorStore(rows[:, :dim], metric=ph.METRIC_L2)
    qh = ph.QuantizedHnsw.reference_shaped(C_, full, cs, bp=ph.BuildParameters(seed=1),
                                           centroid_bp=ph.BuildParameters(seed=2), quantized_search=ph.SearchParameters(64, 64, 2))
    st = qh.store
    assert (st.m, st.dsub) == (dim // cs, cs) and st.ksub <= C_
    codes, cb = st.codes(), st.codebook()
    assert codes.dtype == np.uint16 and codes.shape == (n, dim // cs) and codes.max() < st.ksub
    # centroids are sub-vectors of the data, pairwise distinct (sort + dedup, pq.rs:277-278)
    assert len({c.tobytes() for c in cb}) == st.ksub
    subs = rows[:, :dim].reshape(n * (dim // cs), cs)
    d2 = ((subs[:, None, :] - cb[None, :, :]) ** 2).sum(2) if n * (dim // cs) * st.ksub < 8e6 else None
    if d2 is None:
        pick = np.random.default_rng(0).choice(len(subs), 4000, replace=False)
        d2 = ((subs[pick, None, :] - cb[None, :, :]) ** 2).sum(2)
        got = codes.reshape(-1)[pick]
    else:
        got = codes.reshape(-1)
    exact = d2.argmin(1)
    agree = (d2[np.arange(len(got)), got] <= d2[np.arange(len(got)), exact] * (1 + 1e-6)).mean()
    assert agree > 0.98, agree  # the HNSW quantizer is approximate by design (pq.rs:61-71)
    # reconstruction store == codebook[codes]
    rec = st.reconstruct_store()
    np.testing.assert_array_equal(rec.read().view(np.uint32), cb[codes].reshape(n, dim).view(np.uint32))
    # the same graph searched over the codes and over the reconstructions: identical bits
    g2 = ph.Hnsw.from_layers(rec, [(l.nodes, l.neighbors) for l in qh.hnsw.layers])
    q = oracle.synth_rows(2 ** 32, 150, dim)[:, :dim]
    for sp in (ph.SearchParameters(64, 64, 2), ph.SearchParameters(300, 100, 3), ph.SearchParameters(1000, 300, 2)):
        a = qh.hnsw.search_batch(queries=q, sp=sp, stats=True)
        b = g2.search_batch(queries=q, sp=sp, stats=True)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                                          y.view(np.uint32) if y.dtype == np.float32 else y)
    qid = np.arange(0, n, 9, dtype=np.uint64)
    a = qh.hnsw.search_batch(qids=qid, sp=ph.SearchParameters(64, 64, 2), exclude=qid)
    b = g2.search_batch(qids=qid, sp=ph.SearchParameters(64, 64, 2), exclude=qid)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    # QuantizedHnsw::search with the query quantised like the reference (pq.rs:351-352) and asymmetric
    gt = np.argsort(((q[:, None, :] - rows[None, :, :dim]) ** 2).sum(2), axis=1)[:, :10]
    for quantize in (True, False):
        ids, d, ln = qh.search_batch(q, ph.SearchParameters(128, 128, 4), quantize_query=quantize)
        assert (np.diff(d[:, :100], axis=1) >= 0).all()  # re-ranked by the full comparator, sorted
        rec10 = np.mean([len(set(ids[i, :10].tolist()) & set(gt[i].tolist())) / 10 for i in range(len(q))])
        assert rec10 > 0.1, (quantize, rec10)  # 8 sub-spaces x 500 centroids on iid data is a coarse code: the flow is what is checked
    # the reference's own assertion on this flow (test_pq_recall, pq.rs:956-978): the Hnsw over the quantised
    # vectors finds every stored vector again (self-recall of the reference's estimator)
    assert qh.hnsw.stochastic_recall() >= 0.99
    # the code rows are searched, not built on: build entry points refuse them with a status, not a fault
    with pytest.raises(ph.PhnswError) as e:
        qh.hnsw.improve_neighbors_upto(qh.hnsw.layer_count(), ph.BuildParameters(promote=0, seed=1))
    assert e.value.code == -7


def _threshold_rows(res):
    return [(v, [x[0] for x in got], [np.float32(x[1]).view(np.uint32) for x in got]) for v, got in res]


def test_threshold_nn_over_codes_with_global_memory_queues(monkeypatch):
    """threshold_nn (lib.rs:930-962) on both kinds of code stores: queues past the LDS limit go through
    ph_search_kernel_big's DistPQ / DistPQS instances.  Per-sub-space codes: rows equal the oracle's; shared-codebook
    u16 codes (no oracle form of this store): the global-memory queues alone give the rows the LDS queues give."""
    n, dim, m, ksub = 8000, 16, 4, 64
    rows, full, pq, ocodes, ocb = make(n, dim, m, ksub, seed=1)
    bp_kw = dict(seed=2, promote=0)
    oix = oracle.Index(rows, dim=dim, sum_mode=oracle.SUM_BLOCKED64)
    oix.set_pq(ocodes, ocb, table_f16=0)
    obp = oracle.default_build_params(**bp_kw)
    vs = oracle.shuffle(np.arange(n), obp.seed)
    sizes = oracle.calculate_partitions(n, obp.order)
    for i, sz in enumerate(sizes):
        oix.generate_layer(vs[:sz], 48 if i == len(sizes) - 1 else 24, obp)
        oix.improve_index(obp)
    g = ph.Hnsw.from_layers(pq, [oix.layer(l) for l in range(oix.layer_count)])
    thr = np.float32(np.quantile((1.0 - rows[:, :dim] @ rows[0, :dim]) / 2, 0.3))   # a third of the layer lies within it
    ti, td, tl = oix.threshold_nn(thr, 2, 32, max_out=n)
    assert int((tl > 1024).sum()) >= 20 and int(tl.min()) < 1024   # queues of 2 048 entries and more, and LDS ones
    got = g.threshold_nn(float(thr), 2, 32, max_out=n)
    for i, (v, ids_, bits) in enumerate(_threshold_rows(got)):
        assert ids_ == [int(x) for x in ti[i, :int(tl[i])]], i
        assert bits == [x.view(np.uint32) for x in td[i, :int(tl[i])]], i
    # shared codebook, u16 codes
    n2, dim2 = 3000, 64
    rows2 = oracle.synth_rows(0, n2, dim2)
    full2 = ph.VectorStore(rows2[:, :dim2], metric=ph.METRIC_L2)
    qh = ph.QuantizedHnsw.reference_shaped(300, full2, 16, bp=ph.BuildParameters(seed=1), centroid_bp=ph.BuildParameters(seed=2),
                                           quantized_search=ph.SearchParameters(64, 64, 2))
    dq = qh.hnsw.search_batch(qids=np.arange(1), sp=ph.SearchParameters(512, 512, 2))[1][0]
    thr2 = float(dq[40])
    lds = _threshold_rows(qh.hnsw.threshold_nn(thr2, 2, 8, max_out=1024))
    assert max(len(r[1]) for r in lds) > 16
    monkeypatch.setenv("PHNSW_THRESHOLD_ALL_BIG", "1")
    assert _threshold_rows(qh.hnsw.threshold_nn(thr2, 2, 8, max_out=1024)) == lds


@pytest.mark.parametrize("world,rank", [(2, 1), (3, 0), (8, 7)])
def test_sharded_encode_equals_single_gpu_encode(world, rank):
    """SURVEY 8e row 3 (pq.rs:326-333: the encode is one independent job per vector): both quantizers with the
    encode split over an emulated world -- every rank's vector range encoded by its own launches into its block of
    the (padded) code array -- give the codes of the single-GPU call and of the oracle, including ranges that do not
    divide n and an empty tail rank"""
    n, dim, m, ksub = 3001, 96, 24, 64
    rows, full, pq, ocodes, ocb = make(n, dim, m, ksub, seed=5)
    comm = ph.EmulatedComm(world, rank)
    spq = ph.PqStore(full, m, ksub, 5, comm=comm)
    np.testing.assert_array_equal(spq.codes(), ocodes)
    np.testing.assert_array_equal(spq.codebook().view(np.uint32), ocb.view(np.uint32))
    spk = ph.PqStore(full, m, ksub, 5, kmeans_iters=2, kmeans_sample=1000, comm=comm)
    ref = ph.PqStore(full, m, ksub, 5, kmeans_iters=2, kmeans_sample=1000)
    np.testing.assert_array_equal(spk.codes(), ref.codes())
    # the reference's own shape: shared codebook, u16 codes through the HNSW over the centroids
    f2 = ph.VectorStore(rows[:, :dim], metric=ph.METRIC_L2)
    kw = dict(centroid_bp=ph.BuildParameters(seed=2), quantized_search=ph.SearchParameters(32, 32, 2))
    a = ph.SharedPqStore(f2, 16, 300, seed=3, **kw)
    b = ph.SharedPqStore(f2, 16, 300, seed=3, comm=comm, **kw)
    np.testing.assert_array_equal(a.codes(), b.codes())
    np.testing.assert_array_equal(a.codebook().view(np.uint32), b.codebook().view(np.uint32))
