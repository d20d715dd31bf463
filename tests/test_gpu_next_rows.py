"""SURVEY section 8 'next' rows on the GPU path: f1 on-disk format (serialize.rs), f3 bulk
knn / threshold_nn (lib.rs:905-962)."""
import json
import os

import numpy as np
import pytest

import oracle
import parallel_hnsw_amd as ph
from helpers import EMPTY, load, toy_vectors

pytestmark = pytest.mark.gpu
TOY = load("toy_index.json")


def fixture_graph(entry=0):
    g = TOY["test_generation"]
    data = toy_vectors()
    store = ph.VectorStore(data, metric=ph.METRIC_ONE_MINUS_DOT)
    top = (np.array([entry], dtype=np.uint64), np.full((1, 3), EMPTY, dtype=np.uint64))
    bottom = (np.arange(9, dtype=np.uint64), np.array(g["neighbors"], dtype=np.uint64))
    return store, ph.Hnsw.from_layers(store, [top, bottom])


@pytest.mark.parametrize("entry", [0, 3])
def test_threshold_nn_golden(entry):
    """reference test_threshold_nn (lib.rs:2379-2420) on the reference's own test graph"""
    store, h = fixture_graph(entry)
    t = TOY["test_threshold_nn"]
    res = h.threshold_nn(t["threshold"], t["probe_depth"], t["initial_search_depth"])
    for (v, got), exp in zip(res, t["expect"]):
        assert [g[0] for g in got] == [e[0] for e in exp], v
        np.testing.assert_allclose([g[1] for g in got], [e[1] for e in exp], rtol=1e-5, atol=1e-7)


def test_knn_and_threshold_nn_parity_large():
    n, dim = 4000, 32
    rows = oracle.synth_rows(0, n, dim)
    oix = oracle.Index.generate(rows, np.arange(n), oracle.default_build_params(seed=2), dim=dim,
                                sum_mode=oracle.SUM_BLOCKED64)
    store = ph.VectorStore(rows[:, :dim])
    g = ph.Hnsw.from_layers(store, [oix.layer(l) for l in range(oix.layer_count)])
    ki, kd, kl = oix.knn(5, 2)
    res = g.knn(5, 2)
    for i, (v, got) in enumerate(res):
        assert [x[0] for x in got] == [int(x) for x in ki[i, :int(kl[i])]]
        assert [np.float32(x[1]).view(np.uint32) for x in got] == [x.view(np.uint32) for x in kd[i, :int(kl[i])]]
    # radius query with queue doubling (initial depth 4 forces several resize_capacity steps)
    thr = np.float32(0.33)
    ti, td, tl = oix.threshold_nn(thr, 2, 4, max_out=256)
    res = g.threshold_nn(float(thr), 2, 4, max_out=256)
    assert int(tl.max()) > 8  # the queue really grew
    for i, (v, got) in enumerate(res):
        assert [x[0] for x in got] == [int(x) for x in ti[i, :int(tl[i])]], i
        assert [np.float32(x[1]).view(np.uint32) for x in got] == [x.view(np.uint32) for x in td[i, :int(tl[i])]]


def _same_threshold_rows(res, ti, td, tl):
    for i, (v, got) in enumerate(res):
        assert [x[0] for x in got] == [int(x) for x in ti[i, :int(tl[i])]], i
        assert [np.float32(x[1]).view(np.uint32) for x in got] == [x.view(np.uint32) for x in td[i, :int(tl[i])]], i


def test_threshold_nn_queue_grows_past_lds(monkeypatch):
    """resize_capacity without a bound (lib.rs:949-951): a radius that a third of the layer falls in makes the queue
    double to 2048 and 4096 entries -- past the 1024 the LDS queues hold, through the global-memory queues of
    ph_search_kernel_big -- and every row still equals the oracle's, whose queue is realloc'd without limit"""
    n, dim = 12000, 16
    rows = oracle.synth_rows(0, n, dim)
    oix = oracle.Index.generate(rows, np.arange(n), oracle.default_build_params(seed=5), dim=dim,
                                sum_mode=oracle.SUM_BLOCKED64)
    store = ph.VectorStore(rows[:, :dim])
    g = ph.Hnsw.from_layers(store, [oix.layer(l) for l in range(oix.layer_count)])
    d0 = (1.0 - rows[:, :dim] @ rows[0, :dim]) / 2
    thr = np.float32(np.quantile(d0, 0.3))
    ti, td, tl = oix.threshold_nn(thr, 2, 16, max_out=n)
    assert int(tl.max()) > 2048 and int(tl.min()) < 1024       # both paths are needed, and two doublings past LDS
    res = g.threshold_nn(float(thr), 2, 16, max_out=n)
    _same_threshold_rows(res, ti, td, tl)
    # an initial depth the LDS queues cannot hold at all
    ti, td, tl = oix.threshold_nn(thr, 2, 1500, max_out=n)
    _same_threshold_rows(g.threshold_nn(float(thr), 2, 1500, max_out=n), ti, td, tl)
    # the global-memory queues alone, also where the LDS queues would have done
    monkeypatch.setenv("PHNSW_THRESHOLD_ALL_BIG", "1")
    thr2 = np.float32(np.quantile(d0, 0.01))
    ti, td, tl = oix.threshold_nn(thr2, 2, 4, max_out=512)
    assert 8 < int(tl.max()) < 512
    _same_threshold_rows(g.threshold_nn(float(thr2), 2, 4, max_out=512), ti, td, tl)
    # few resident waves (the memory budget of the big queues): every wave serves many nodes in turn
    monkeypatch.setenv("PHNSW_THRESHOLD_BIG_BYTES", "400000")
    monkeypatch.setenv("PHNSW_THRESHOLD_BIG_PIECE", "5000")   # and the node list in three pieces
    _same_threshold_rows(g.threshold_nn(float(thr2), 2, 4, max_out=512), ti, td, tl)


@pytest.mark.parametrize("seed", range(6))
def test_threshold_nn_random_shapes(seed, monkeypatch):
    """random layers, radii, initial depths and probe depths: the LDS queues, the global-memory queues alone
    (PHNSW_THRESHOLD_ALL_BIG) and the oracle agree row by row"""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(200, 2500))
    dim = int(rng.choice([4, 8, 16, 32]))
    isd = int(rng.choice([1, 2, 3, 7, 16, 64, 100]))
    pd = int(rng.integers(1, 5))
    rows = oracle.synth_rows(seed * 10007, n, dim)
    bp = oracle.default_build_params(seed=seed + 1)
    oix = oracle.Index.generate(rows, np.arange(n), bp, dim=dim, sum_mode=oracle.SUM_BLOCKED64)
    store = ph.VectorStore(rows[:, :dim])
    g = ph.Hnsw.from_layers(store, [oix.layer(l) for l in range(oix.layer_count)])
    d0 = (1.0 - rows[:, :dim] @ rows[int(rng.integers(0, n)), :dim]) / 2
    thr = np.float32(np.quantile(d0, float(rng.choice([0.01, 0.05, 0.2, 0.6, 1.0]))))
    ti, td, tl = oix.threshold_nn(thr, pd, isd, max_out=n)
    _same_threshold_rows(g.threshold_nn(float(thr), pd, isd, max_out=n), ti, td, tl)
    monkeypatch.setenv("PHNSW_THRESHOLD_ALL_BIG", "1")
    monkeypatch.setenv("PHNSW_THRESHOLD_BIG_PIECE", str(int(rng.integers(50, 3000))))
    _same_threshold_rows(g.threshold_nn(float(thr), pd, isd, max_out=n), ti, td, tl)


def test_serialize_layout_and_roundtrip(tmp_path):
    """serialize_hnsw / deserialize_hnsw  serialize.rs:33-209"""
    n, dim = 1500, 16
    rows = oracle.synth_rows(0, n, dim)
    store = ph.VectorStore(rows[:, :dim])
    bp = ph.BuildParameters(order=6, neighborhood_size=8, zero_layer_neighborhood_size=16, seed=3)
    h = ph.Hnsw.generate(store, np.arange(n), bp)
    p = tmp_path / "index"
    h.serialize(p)
    L = h.layer_count()
    meta = json.load(open(p / "meta"))
    assert meta["layer_count"] == L
    b = meta["build_parameters"]
    assert list(b) == ["order", "zero_layer_neighborhood_size", "neighborhood_size", "optimization",
                       "initial_partition_search"]  # serde field order (parameters.rs:42-48)
    assert (b["order"], b["zero_layer_neighborhood_size"], b["neighborhood_size"]) == (6, 16, 8)
    assert b["optimization"]["search"] == {"number_of_candidates": 300, "upper_layer_candidate_count": 300,
                                           "probe_depth": 2}
    raw = open(p / "meta").read()
    assert '"promotion_threshold":0.01,' in raw and '"promotion_proportion":1.0,' in raw  # serde_json f32 text
    assert (p / "comparator").is_dir()
    for lft in range(L):
        number = L - lft - 1  # layers are numbered from the bottom (serialize.rs:67)
        lay = h._layer(lft)
        lm = json.load(open(p / ("layer.meta.%d" % number)))
        assert lm == {"node_count": lay.node_count(), "neighborhood_size": lay.neighborhood_size}
        nodes = np.fromfile(p / ("layer.nodes.%d" % number), dtype="<u8")
        nb = np.fromfile(p / ("layer.neighbors.%d" % number), dtype="<u8")
        np.testing.assert_array_equal(nodes, lay.nodes)
        np.testing.assert_array_equal(nb.reshape(lay.neighbors.shape), lay.neighbors)
        assert (nb == EMPTY).sum() == (lay.neighbors == EMPTY).sum()
    h2 = ph.Hnsw.deserialize(p, store)
    assert h2.layer_count() == L
    assert h2.build_parameters.order == 6 and h2.build_parameters.neighborhood_size == 8
    # create_dir_all (serialize.rs:41-47): parents are made on the way
    deep = tmp_path / "a" / "b" / "index"
    h.serialize(deep)
    assert ph.Hnsw.deserialize(deep, store).layer_count() == L
    # a corrupt / crafted layer.meta must be refused before it sizes anything (no exception may cross the ABI)
    good = open(deep / "layer.meta.0").read()
    for bad in ('{"node_count":18446744073709551615,"neighborhood_size":16}', '{"node_count":1e300,"neighborhood_size":16}',
                '{"node_count":1500,"neighborhood_size":0}', '{"node_count":1500,"neighborhood_size":2305843009213693952}',
                '{"node_count":-5,"neighborhood_size":16}', '{"node_count":2.5,"neighborhood_size":16}'):
        open(deep / "layer.meta.0", "w").write(bad)
        with pytest.raises(ph.PhnswError) as e:
            ph.Hnsw.deserialize(deep, store)
        assert e.value.code in (-1, -7), bad
    open(deep / "layer.meta.0", "w").write(good)
    assert ph.Hnsw.deserialize(deep, store).layer_count() == L
    q = oracle.synth_rows(2 ** 32, 50, dim)[:, :dim]
    a = h.search_batch(queries=q, sp=ph.SearchParameters(64, 64, 2))
    c = h2.search_batch(queries=q, sp=ph.SearchParameters(64, 64, 2))
    np.testing.assert_array_equal(a[0], c[0])
    np.testing.assert_array_equal(a[1].view(np.uint32), c[1].view(np.uint32))
    # a layer stack written by "the crate" (here: by hand) is adopted as well
    os.remove(p / "comparator" / "vectors.f32")
    h3 = ph.Hnsw.deserialize(p, store)
    assert h3.layer_count() == L
    # no comparator entry => Index not found (serialize.rs:144-146)
    import shutil
    shutil.rmtree(p / "comparator")
    with pytest.raises(ph.PhnswError) as e:
        ph.Hnsw.deserialize(p, store)
    assert "not found" in str(e.value).lower()
    with pytest.raises(ph.PhnswError):
        ph.Hnsw.deserialize(tmp_path / "nothing-here", store)


def test_store_append_keeps_ids_and_matches_a_whole_store():
    """phnsw_store_append: the grown store behaves exactly like one created with all rows"""
    rows = oracle.synth_rows(0, 800, 40)[:, :40]
    whole = ph.VectorStore(rows)
    grown = ph.VectorStore(rows[:500])
    assert grown.append(rows[500:]) == 500
    assert grown.n == 800
    np.testing.assert_array_equal(grown.read().view(np.uint32), whole.read().view(np.uint32))
    ids = np.arange(800, dtype=np.uint64)
    np.testing.assert_array_equal(grown.compare_vec(ph.Stored(3), ids).view(np.uint32),
                                  whole.compare_vec(ph.Stored(3), ids).view(np.uint32))
    bp = ph.BuildParameters(seed=3)
    a = ph.Hnsw.generate(grown, ids, bp)
    b = ph.Hnsw.generate(whole, ids, bp)
    for x, y in zip(a.layers, b.layers):
        np.testing.assert_array_equal(x.nodes, y.nodes)
        np.testing.assert_array_equal(x.neighbors, y.neighbors)
    bad = rows[:2].copy()
    bad[1, 5] = np.nan
    with pytest.raises(ph.PhnswError):
        grown.append(bad)
    assert grown.n == 800


def test_extend_layer_matches_the_oracle():
    """Hnsw::extend_layer lib.rs:1039-1068 through the ABI: same renumbering as the oracle's, search
    over the extended stack still works, inserting an existing vector is refused (lib.rs:1797)"""
    rows = oracle.synth_rows(0, 50, 8)
    nodes = np.array([3, 10, 20, 30], dtype=np.uint64)
    nb = np.array([[1, 2, EMPTY], [0, 3, EMPTY], [0, EMPTY, EMPTY], [1, EMPTY, EMPTY]], dtype=np.uint64)
    oix = oracle.Index(rows, dim=8)
    oix.push_layer(nodes, nb, 3)
    assert oix.extend_layer(0, [15, 1]) == 0
    store = ph.VectorStore(rows[:, :8])
    g = ph.Hnsw.from_layers(store, [(nodes, nb)])
    g.extend_layer(0, [15, 1])
    on, onb = oix.layer(0)
    np.testing.assert_array_equal(g._layer(0).nodes, on)
    np.testing.assert_array_equal(g._layer(0).neighbors, onb)
    gi, gd, gl = g.search_batch(qids=[3, 15], sp=ph.SearchParameters(4, 4, 2))
    ci, cd, cl = oix.search(qids=[3, 15], sp=(4, 4, 2))
    np.testing.assert_array_equal(gi, ci)
    np.testing.assert_array_equal(gl, cl)
    with pytest.raises(ph.PhnswError):
        g.extend_layer(0, [10])
    with pytest.raises(ph.PhnswError):
        g.extend_layer(0, [40, 40])
