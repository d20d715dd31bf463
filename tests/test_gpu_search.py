"""GPU parity: the HIP search path (through the C ABI) against the CPU oracle on the same
seeded inputs and the same graph.  Integer results (ids, lengths, hop / distance counters)
must be identical; distances are compared bit-exactly against the oracle in the kernel's
summation order (ORC_SUM_BLOCKED64) and within 1e-5 relative against the reference's
sequential order."""
import os

import numpy as np
import pytest

import oracle
import parallel_hnsw_amd as ph
from helpers import EMPTY, load, sub, toy_vectors

pytestmark = pytest.mark.gpu

TOY = load("toy_index.json")


def build_oracle_index(n, dim, seed=0, metric=oracle.METRIC_COSINE_HALF, normalize=True, bp_kw=None, threads=8):
    rows = oracle.synth_rows(0, n, dim, seed=42, normalize=normalize)
    bp = oracle.default_build_params(seed=seed, **(bp_kw or {}))
    ix = oracle.Index.generate(rows, np.arange(n), bp, dim=dim, metric=metric, sum_mode=oracle.SUM_BLOCKED64,
                               threads=threads)
    return rows, ix


def to_gpu(rows, dim, ix, metric):
    store = ph.VectorStore(rows[:, :dim], metric=metric)
    layers = [ix.layer(l) for l in range(ix.layer_count)]
    return store, ph.Hnsw.from_layers(store, layers)


def assert_same(gpu, cpu, exact=True):
    gi, gd, gl = gpu[:3]
    ci, cd, cl = cpu[:3]
    np.testing.assert_array_equal(gl, cl)
    np.testing.assert_array_equal(gi, ci)
    if exact:
        np.testing.assert_array_equal(gd.view(np.uint32), cd.view(np.uint32))
    else:
        np.testing.assert_allclose(gd, cd, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("metric", [oracle.METRIC_COSINE_HALF, oracle.METRIC_ONE_MINUS_DOT, oracle.METRIC_L2])
@pytest.mark.parametrize("dim", [3, 32, 100, 128, 768])
def test_distance_batch_bit_exact(dim, metric):
    n = 300
    rows = oracle.synth_rows(0, n, dim, normalize=(metric != oracle.METRIC_L2))
    store = ph.VectorStore(rows[:, :dim], metric=metric)
    ix = oracle.Index(rows, dim=dim, metric=metric)
    ids = np.arange(n, dtype=np.uint64)
    q = oracle.synth_rows(2 ** 32, 1, dim, normalize=(metric != oracle.METRIC_L2))[0, :dim]
    got = store.compare_vec(ph.Unstored(q), ids)
    blocked = np.array([ix.distance(q, rows[i, :dim], oracle.SUM_BLOCKED64) for i in range(n)], dtype=np.float32)
    seq = np.array([ix.distance(q, rows[i, :dim], oracle.SUM_SEQ) for i in range(n)], dtype=np.float32)
    np.testing.assert_array_equal(got.view(np.uint32), blocked.view(np.uint32))
    np.testing.assert_allclose(got, seq, rtol=1e-5, atol=1e-6)  # north_star: distances within 1e-5 relative
    got_s = store.compare_vec(ph.Stored(7), ids)
    blocked_s = np.array([ix.distance(rows[7, :dim], rows[i, :dim], oracle.SUM_BLOCKED64) for i in range(n)],
                         dtype=np.float32)
    np.testing.assert_array_equal(got_s.view(np.uint32), blocked_s.view(np.uint32))


def test_synthetic_store_matches_oracle_generator():
    s = ph.VectorStore.synthetic(1000, 96, seed=42)
    np.testing.assert_array_equal(s.read().view(np.uint32), oracle.synth_rows(0, 1000, 96)[:, :96].view(np.uint32))
    s2 = ph.VectorStore.synthetic(50, 7, seed=9, first=2 ** 32, normalize=False)
    np.testing.assert_array_equal(s2.read().view(np.uint32),
                                  oracle.synth_rows(2 ** 32, 50, 7, seed=9, normalize=False)[:, :7].view(np.uint32))


def test_clustered_store_matches_oracle_generator():
    s = ph.VectorStore.clustered(3000, 96, seed=42, n_clusters=37, noise=1.0)
    np.testing.assert_array_equal(s.read().view(np.uint32),
                                  oracle.synth_clustered_rows(0, 3000, 96, n_clusters=37)[:, :96].view(np.uint32))
    s2 = ph.VectorStore.clustered(64, 10, seed=5, first=2 ** 32, n_clusters=3, noise=0.25)
    np.testing.assert_array_equal(
        s2.read().view(np.uint32),
        oracle.synth_clustered_rows(2 ** 32, 64, 10, seed=5, n_clusters=3, noise=0.25)[:, :10].view(np.uint32))


@pytest.mark.parametrize("n,dim,ef,upper,pd", [
    (2000, 128, 64, 64, 2),
    (2000, 128, 128, 16, 2),     # upper_layer_candidate_count < ef: take() + merge path
    (5000, 768, 128, 128, 2),    # BASELINE config 2 search parameters
    (5000, 768, 300, 300, 2),    # reference defaults (parameters.rs:10-18)
    (3000, 100, 6, 6, 2),        # initial_partition_search (parameters.rs:57-61)
    (3000, 32, 40, 40, 1),
    (3000, 32, 40, 40, 7),
    (1500, 1536, 32, 32, 2),     # reference test dimension (lib.rs:2219)
])
def test_search_parity(n, dim, ef, upper, pd):
    rows, oix = build_oracle_index(n, dim)
    store, gix = to_gpu(rows, dim, oix, oracle.METRIC_COSINE_HALF)
    q = oracle.synth_rows(2 ** 32, 257, dim)[:, :dim]
    sp = ph.SearchParameters(ef, upper, pd)
    gpu = gix.search_batch(queries=q, sp=sp, stats=True)
    cpu = oix.search(queries=q, sp=(ef, upper, pd), stats=True)
    assert_same(gpu, cpu)
    np.testing.assert_array_equal(gpu[3], cpu[3])  # distance evaluations and hops per query
    # The reference's own summation order (sequential f32, bigvec.rs:48-51) on the same graph: distances
    # within 1e-5 relative wherever the ids agree, and every slot of the top 10 whose ids differ PROVEN a
    # near-tie swap: both ids are in the other list inside one group of distances that agree
    # to 1e-5 relative -- only their order changed.  No tolerance on how many slots may differ otherwise.
    oix.set_sum_mode(oracle.SUM_SEQ)
    seq = oix.search(queries=q, sp=(ef, upper, pd))
    m = gpu[0] == seq[0]
    np.testing.assert_allclose(gpu[1][m], seq[1][m], rtol=1e-5, atol=1e-6)
    k = min(10, max(1, ef // 2))
    rep = oracle.tie_swap_report(gpu[0].astype(np.int64), gpu[1], seq[0].astype(np.int64), seq[1], k=k)
    assert rep["unexplained"] == 0, rep
    # north_star: recall@10 of the two arithmetic orders within 0.2 % (against exact brute force)
    gt, _ = oix.bruteforce(q, 10)
    rec = [np.mean([len(set(r[i, :10].tolist()) & set(gt[i].tolist())) / 10.0 for i in range(len(q))])
           for r in (gpu[0], seq[0])]
    assert abs(rec[0] - rec[1]) <= 0.002, rec


def test_search_parity_stored_and_exclude():
    n, dim = 4000, 128
    rows, oix = build_oracle_index(n, dim)
    store, gix = to_gpu(rows, dim, oix, oracle.METRIC_COSINE_HALF)
    qids = np.arange(0, n, 7, dtype=np.uint64)
    sp = ph.SearchParameters(300, 300, 2)
    # link-round shape: Stored query, exclude self (lib.rs:1112-1117)
    gpu = gix.search_batch(qids=qids, sp=sp, exclude=qids, stats=True)
    cpu = oix.search(qids=qids, sp=(300, 300, 2), exclude=qids, stats=True)
    assert_same(gpu, cpu)
    np.testing.assert_array_equal(gpu[3], cpu[3])
    assert not (gpu[0] == qids[:, None]).any() or True  # the entry-vector quirk may keep self
    # exclude == the entry vector stays in the result (search.rs:111 inserts it unfiltered)
    entry = gix.entry_vector()
    g2 = gix.search_batch(qids=[entry], sp=sp, exclude=[entry])
    c2 = oix.search(qids=[entry], sp=(300, 300, 2), exclude=[entry])
    assert_same(g2, c2)
    # no exclusion: every stored vector finds itself first (test_recall lib.rs:2166-2192)
    g3 = gix.search_batch(qids=qids, sp=sp)
    c3 = oix.search(qids=qids, sp=(300, 300, 2))
    assert_same(g3, c3)


def test_search_upto_and_l2_metric():
    n, dim = 3000, 32
    rows, oix = build_oracle_index(n, dim, metric=oracle.METRIC_L2, normalize=False)
    store, gix = to_gpu(rows, dim, oix, oracle.METRIC_L2)
    q = oracle.synth_rows(2 ** 32, 100, dim, normalize=False)[:, :dim]
    sp = ph.SearchParameters(64, 64, 2)
    assert_same(gix.search_batch(queries=q, sp=sp), oix.search(queries=q, sp=(64, 64, 2)))


def fixture_graph(entry):
    g = TOY["test_generation"]
    data = toy_vectors()
    store = ph.VectorStore(data, metric=ph.METRIC_ONE_MINUS_DOT)
    top = (np.array([entry], dtype=np.uint64), np.full((1, 3), EMPTY, dtype=np.uint64))
    bottom = (np.arange(9, dtype=np.uint64), np.array(g["neighbors"], dtype=np.uint64))
    return ph.Hnsw.from_layers(store, [top, bottom])


@pytest.mark.parametrize("entry", TOY["test_generation"]["nearness_entries"])
def test_nearness_search_golden(entry):
    """reference test_nearness_search (lib.rs:2046-2068) on the reference's own test graph"""
    h = fixture_graph(entry)
    t = TOY["test_nearness_search"]
    got = h.search(ph.Unstored(sub(t["query"])), ph.SearchParameters(*t["search"]))
    assert [g[0] for g in got] == [e[0] for e in t["expect"]]
    np.testing.assert_allclose([g[1] for g in got], [e[1] for e in t["expect"]], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("entry", [0, 5])
def test_knn_golden(entry):
    """reference test_knn (lib.rs:2358-2377)"""
    h = fixture_graph(entry)
    t = TOY["test_knn"]
    res = h.knn(t["k"], t["probe_depth"])
    for (v, got), exp in zip(res, t["expect"]):
        assert [g[0] for g in got] == [e[0] for e in exp]
        np.testing.assert_allclose([g[1] for g in got], [e[1] for e in exp], rtol=1e-5, atol=1e-7)


def test_edge_cases():
    data = toy_vectors()
    store = ph.VectorStore(data, metric=ph.METRIC_ONE_MINUS_DOT)
    # single node, single layer, empty row
    h = ph.Hnsw.from_layers(store, [(np.array([4], dtype=np.uint64), np.full((1, 6), EMPTY, dtype=np.uint64))])
    r = h.search(ph.Stored(4), ph.SearchParameters(10, 10, 2))
    assert [x[0] for x in r] == [4]
    ids, d, ln = h.search_batch(queries=np.zeros((0, 3), dtype=np.float32), sp=ph.SearchParameters(10, 10, 2))
    assert ids.shape == (0, 10)
    # layers that are not nested: get_node().unwrap() panics in the reference (lib.rs:261)
    bad = ph.Hnsw.from_layers(store, [(np.array([8], dtype=np.uint64), np.full((1, 3), EMPTY, dtype=np.uint64)),
                                      (np.arange(5, dtype=np.uint64), np.full((5, 6), EMPTY, dtype=np.uint64))])
    with pytest.raises(ph.PhnswError) as e:
        bad.search(ph.Stored(0), ph.SearchParameters(10, 10, 2))
    assert e.value.code == -4
    # invalid parameters / graphs are refused at the boundary
    with pytest.raises(ph.PhnswError):
        h.search(ph.Stored(4), ph.SearchParameters(0, 0, 2))
    with pytest.raises(ph.PhnswError):
        h.search(ph.Stored(4), ph.SearchParameters(10, 10, 0))
    with pytest.raises(ph.PhnswError):
        ph.Hnsw.from_layers(store, [(np.array([1, 0], dtype=np.uint64), np.full((2, 3), EMPTY, dtype=np.uint64))])
    # a NodeId twice in one row -- the crate's racy link step can write that (lib.rs:1123-1147): the later
    # occurrence is dropped on import (documented deviation), PHNSW_STRICT_IMPORT refuses the row
    dup_layers = [(np.array([0, 1, 2], dtype=np.uint64),
                   np.array([[1, 1, 2], [0, 2, 0], [1, EMPTY, EMPTY]], dtype=np.uint64))]
    hd = ph.Hnsw.from_layers(store, dup_layers)
    np.testing.assert_array_equal(hd._layer(0).neighbors,
                                  np.array([[1, 2, EMPTY], [0, 2, EMPTY], [1, EMPTY, EMPTY]], dtype=np.uint64))
    assert len(hd.search(ph.Stored(0), ph.SearchParameters(10, 10, 2))) == 3
    os.environ["PHNSW_STRICT_IMPORT"] = "1"
    try:
        with pytest.raises(ph.PhnswError):
            ph.Hnsw.from_layers(store, dup_layers)
    finally:
        del os.environ["PHNSW_STRICT_IMPORT"]
    with pytest.raises(ph.PhnswError):
        ph.VectorStore(np.array([[np.nan, 0, 0]], dtype=np.float32))


def test_frontier_spill_path():
    """tiny queue + large probe_depth forces pops from the spill list (visit_queue entries
    that are not in the result queue, lib.rs:211-220)"""
    n, dim = 3000, 32
    rows, oix = build_oracle_index(n, dim)
    store, gix = to_gpu(rows, dim, oix, oracle.METRIC_COSINE_HALF)
    q = oracle.synth_rows(2 ** 32, 64, dim)[:, :dim]
    for ef, pd in [(1, 30), (2, 50), (6, 200)]:
        gpu = gix.search_batch(queries=q, sp=ph.SearchParameters(ef, ef, pd), stats=True)
        cpu = oix.search(queries=q, sp=(ef, ef, pd), stats=True)
        assert_same(gpu, cpu)
        np.testing.assert_array_equal(gpu[3], cpu[3])


@pytest.mark.parametrize("n,dim,metric,k", [(5000, 768, 0, 10), (3000, 100, 1, 16), (700, 3, 0, 1), (129, 32, 0, 5)])
def test_bruteforce_mfma_exact(n, dim, metric, k):
    """G1 ground truth: f32 MFMA GEMM + top-k == the oracle's exact kNN in sequential-fma order,
    bit for bit (ids and distances), and == the reference's sequential order up to near ties"""
    rows = oracle.synth_rows(0, n, dim)
    store = ph.VectorStore(rows[:, :dim], metric=metric)
    q = oracle.synth_rows(2 ** 32, 300, dim)[:, :dim]
    gi, gd = store.bruteforce_topk(q, k)
    oix = oracle.Index(rows, dim=dim, metric=metric)
    ci, cd = oix.bruteforce(q, k, sum_mode=oracle.SUM_SEQFMA)
    np.testing.assert_array_equal(gi, ci)
    np.testing.assert_array_equal(gd.view(np.uint32), cd.view(np.uint32))
    si, sd = oix.bruteforce(q, k, sum_mode=oracle.SUM_SEQ)
    assert (gi == si).mean() > 0.99
    np.testing.assert_allclose(gd, sd, rtol=1e-5, atol=1e-6)
    with pytest.raises(ph.PhnswError):
        ph.VectorStore(rows[:, :dim], metric=2).bruteforce_topk(q, k)  # L2 is not a GEMM here


def test_index_counters_sum_the_per_query_stats():
    """phnsw_index_counters = the reference's per-query SearchStats (search.rs:93-99) summed over
    every launch on the index"""
    rows, ix = build_oracle_index(3000, 32, seed=4)
    store, g = to_gpu(rows, 32, ix, oracle.METRIC_COSINE_HALF)
    assert g.counters() == (0, 0)
    q = oracle.synth_rows(2 ** 32, 500, 32)[:, :32]
    _, _, _, st = g.search_batch(queries=q, sp=ph.SearchParameters(24, 24, 2), stats=True)
    assert g.counters() == (int(st[:, 0].sum()), int(st[:, 1].sum()))
    _, _, _, st2 = g.search_batch(qids=np.arange(100), sp=ph.SearchParameters(8, 8, 2), stats=True)
    assert g.counters() == (int(st[:, 0].sum() + st2[:, 0].sum()), int(st[:, 1].sum() + st2[:, 1].sum()))


def test_two_batches_in_flight_on_two_streams():
    """a throughput caller's arrangement: two query batches alternate over two streams (the second one made by
    phnsw_stream_create_beside, which tests that it runs beside the first); every launch gives the rows the oracle
    gives, whichever lane it ran on and whatever ran beside it"""
    import torch
    rows, ix = build_oracle_index(6000, 64, seed=9)
    store, g = to_gpu(rows, 64, ix, oracle.METRIC_COSINE_HALF)
    dev = torch.device("cuda", 0)
    sp, spt = ph.SearchParameters(64, 64, 3), (64, 64, 3)
    s1 = ph.stream_create_beside(0, 0)
    assert s1 != 0
    lanes = []
    for k, st in enumerate((0, s1)):
        q = np.ascontiguousarray(oracle.synth_rows(2 ** 32 + 7919 * k, 3000, 64)[:, :64])
        qd = torch.from_numpy(q).to(dev)
        lanes.append(dict(q=q, qd=qd, ld=64, stream=st,
                          ids=torch.empty((3000, 64), dtype=torch.int32, device=dev), d=torch.empty((3000, 64), dtype=torch.float32, device=dev),
                          ln=torch.empty(3000, dtype=torch.int32, device=dev), status=torch.empty(3000, dtype=torch.int32, device=dev)))
    torch.cuda.synchronize()
    for i in range(8):
        a = lanes[i & 1]
        g.search_batch_device(3000, sp, a["ids"].data_ptr(), a["d"].data_ptr(), a["ln"].data_ptr(), a["status"].data_ptr(),
                              queries=a["qd"].data_ptr(), ldq=a["ld"], stream=a["stream"])
    torch.cuda.synchronize()
    for a in lanes:
        ci, cd, cl = ix.search(queries=a["q"], sp=spt)
        assert int(a["status"].abs().sum()) == 0 and int(cl.min()) == 64
        np.testing.assert_array_equal(a["ln"].cpu().numpy().astype(np.uint64), cl.astype(np.uint64))
        np.testing.assert_array_equal(a["ids"].cpu().numpy().view(np.uint32).astype(np.uint64), ci.astype(np.uint64))
        np.testing.assert_array_equal(a["d"].cpu().numpy().view(np.uint32), cd.view(np.uint32))


def test_concurrent_host_threads_share_an_index():
    """search(&self) is re-entrant in the reference (Rayon calls it from many threads, lib.rs:1107-1117);
    here concurrent callers share the index's two workspaces behind a mutex"""
    import threading
    rows, ix = build_oracle_index(4000, 48, seed=9)
    store, g = to_gpu(rows, 48, ix, oracle.METRIC_COSINE_HALF)
    qs = [oracle.synth_rows(2 ** 32 + 1000 * t, 300, 48)[:, :48] for t in range(4)]
    sp = ph.SearchParameters(32, 32, 2)
    want = [g.search_batch(queries=q, sp=sp, stats=True) for q in qs]
    got = [[None] * 3 for _ in range(4)]

    def work(t):
        for r in range(3):
            got[t][r] = g.search_batch(queries=qs[t], sp=sp, stats=True)

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for t in range(4):
        for r in range(3):
            for a, b in zip(got[t][r], want[t]):
                np.testing.assert_array_equal(a, b)


def test_hnsw_accessors_mirror_the_crate():
    """lib.rs:592-650, 895-901, 968-984"""
    rows, ix = build_oracle_index(800, 16, seed=2, bp_kw=dict(order=6, neighborhood_size=6, zero_layer_neighborhood_size=12))
    store, g = to_gpu(rows, 16, ix, oracle.METRIC_COSINE_HALF)
    L = g.layer_count()
    assert len(g) == g.vector_count() == 800 and not g.is_empty()
    assert g.comparator() is store
    assert g.get_layer(0).node_count() == 800 and g.get_layer_from_top(L - 1).node_count() == 800
    assert g.get_layer_from_top(L) is None and g.get_layer_above(0) is None
    assert g.get_layer_above(L - 1).node_count() == g.get_layer(1).node_count()
    assert list(g.all_vectors()) == list(range(800))
    np.testing.assert_array_equal(g.supers_for_layer(0), g.get_layer(1).nodes)
    assert list(g.supers_for_layer(L - 1)) == [g.entry_vector()]


@pytest.mark.parametrize("n,dim,ef,pd,metric", [(3000, 64, 64, 2, 0), (20000, 96, 300, 3, 1), (6000, 768, 600, 2, 0),
                                                 (4000, 32, 40, 4, 2)])
def test_search_instrumented_equals_the_oracle(n, dim, ef, pd, metric):
    """Hnsw::search_instrumented (lib.rs:667-673): results AND the index_distance of search_layers_instrumented
    (search.rs:93-140 / lib.rs:211-231: index sums carried by every visit_queue entry) against the oracle, for raw and
    stored queries; the results equal the plain search's"""
    rows = oracle.synth_rows(0, n, dim, normalize=metric != 2)
    oix = oracle.Index.generate(rows, np.arange(n), oracle.default_build_params(seed=3), dim=dim, metric=metric,
                                sum_mode=oracle.SUM_BLOCKED64)
    store = ph.VectorStore(rows[:, :dim], metric=metric)
    gix = ph.Hnsw.from_layers(store, [oix.layer(l) for l in range(oix.layer_count)])
    q = oracle.synth_rows(2 ** 32, 200, dim, normalize=metric != 2)[:, :dim]
    sp, spo = ph.SearchParameters(ef, ef, pd), (ef, ef, pd)
    gi, gd, gl, gx = gix.search_instrumented_batch(queries=q, sp=sp)
    ci, cd, cl, cx = oix.search_instrumented(queries=q, sp=spo)
    np.testing.assert_array_equal(gi, ci)
    np.testing.assert_array_equal(gd.view(np.uint32), cd.view(np.uint32))
    np.testing.assert_array_equal(gl, cl)
    np.testing.assert_array_equal(gx, cx)
    assert (gx > 0).any()
    first_raw = int(cx[0])
    pi, pdist, pl = gix.search_batch(queries=q, sp=sp)
    np.testing.assert_array_equal(gi, pi)
    qid = np.arange(0, n, 37, dtype=np.uint64)
    gi, gd, gl, gx = gix.search_instrumented_batch(qids=qid, sp=sp)
    ci, cd, cl, cx = oix.search_instrumented(qids=qid, sp=spo)
    np.testing.assert_array_equal(gi, ci)
    np.testing.assert_array_equal(gx, cx)
    res, idx = gix.search_instrumented(ph.Unstored(q[0]), sp)
    assert idx == first_raw and len(res) == int(pl[0]) and res[0][0] == int(pi[0, 0])


def test_search_instrumented_spill_path(monkeypatch):
    """the index sums of entries that fall out of the queue travel through the spill list (small queue, deep probe,
    a 64-entry list forces the overflow -> re-run path as well)"""
    n, dim = 8000, 32
    rows = oracle.synth_rows(0, n, dim)
    oix = oracle.Index.generate(rows, np.arange(n), oracle.default_build_params(seed=5), dim=dim, sum_mode=oracle.SUM_BLOCKED64)
    store = ph.VectorStore(rows[:, :dim])
    q = oracle.synth_rows(2 ** 32, 64, dim)[:, :dim]
    for cap in (None, "64"):
        if cap:
            monkeypatch.setenv("PHNSW_OVF_CAP", cap)
        gix = ph.Hnsw.from_layers(store, [oix.layer(l) for l in range(oix.layer_count)])
        gi, gd, gl, gx = gix.search_instrumented_batch(queries=q, sp=ph.SearchParameters(8, 8, 40))
        ci, cd, cl, cx = oix.search_instrumented(queries=q, sp=(8, 8, 40))
        np.testing.assert_array_equal(gi, ci)
        np.testing.assert_array_equal(gx, cx)


def test_host_path_topk_and_pipelined_chunks(monkeypatch):
    """the host-pointer entry points (csrc/hostpath.hip): top-k transfer == the leading columns of the whole queue;
    a list cut into many pipelined chunks (two staging slots, two streams) == the same list in one piece, for raw
    and stored queries with exclude and counters; repeated calls reuse the staging"""
    n, dim = 20000, 96
    store = ph.VectorStore.synthetic(n, dim, seed=42)
    h = ph.Hnsw.generate(store, np.arange(n), ph.BuildParameters(seed=2))
    q = ph.VectorStore.synthetic(1500, dim, seed=42, first=2 ** 32).read()
    sp = ph.SearchParameters(100, 100, 3)
    full = h.search_batch(queries=q, sp=sp, stats=True)
    for k in (1, 10, 100):
        ti, td, tl = h.search_batch(queries=q, sp=sp, k=k)
        np.testing.assert_array_equal(ti, full[0][:, :k])
        np.testing.assert_array_equal(td.view(np.uint32), full[1][:, :k].view(np.uint32))
        np.testing.assert_array_equal(tl, np.minimum(full[2], k))
    qid = np.arange(0, n, 13, dtype=np.uint64)
    sfull = h.search_batch(qids=qid, sp=sp, exclude=qid, stats=True)
    monkeypatch.setenv("PHNSW_HOST_CHUNKS", "64,16,100")   # pipeline from 64 queries on: first chunk 16, pieces of <= 100
    for _ in range(2):
        got = h.search_batch(queries=q, sp=sp, stats=True)
        for a, b in zip(got, full):
            np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                                          b.view(np.uint32) if b.dtype == np.float32 else b)
        got = h.search_batch(qids=qid, sp=sp, exclude=qid, stats=True)
        for a, b in zip(got, sfull):
            np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                                          b.view(np.uint32) if b.dtype == np.float32 else b)
        ti, td, tl = h.search_batch(queries=q, sp=sp, k=7)
        np.testing.assert_array_equal(ti, full[0][:, :7])
    assert not (sfull[0] == qid[:, None]).any()


def test_host_path_concurrent_callers():
    """search entry points are thread safe (Hnsw::search takes &self and is called from the Rayon pool,
    lib.rs:1107-1117): four threads share one index, each call gets a staging set of its own"""
    import threading
    n, dim = 20000, 64
    store = ph.VectorStore.synthetic(n, dim, seed=42)
    h = ph.Hnsw.generate(store, np.arange(n), ph.BuildParameters(seed=2))
    sp = ph.SearchParameters(64, 64, 2)
    qs = [ph.VectorStore.synthetic(300 + 50 * t, dim, seed=42, first=2 ** 32 + 1000 * t).read() for t in range(4)]
    want = [h.search_batch(queries=x, sp=sp) for x in qs]
    got, errs = [None] * 4, []

    def work(t):
        try:
            for _ in range(5):
                got[t] = h.search_batch(queries=qs[t], sp=sp)
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    for t in range(4):
        np.testing.assert_array_equal(got[t][0], want[t][0])
        np.testing.assert_array_equal(got[t][1].view(np.uint32), want[t][1].view(np.uint32))
