"""The dense distance table kept across the rounds of a build (csrc/tiny.hip, PhBuildTable): a link round, a recall
estimate and discover_unreachable all search with Stored(node of layer X) queries, whose distances to the table layer's
nodes do not change between rounds, so the rows are computed once per layer and reused until a node list changes.
With the threshold lowered so that EVERY table is kept (PHNSW_BUILD_TABLE_MIN=1) the graphs must equal the oracle's and
the ones built with the cache off -- plain builds, builds with promotion (node lists change under the cache: extend_layer,
re-topping), improve_index on a finished index, sharded builds (a rank keeps only its node range)."""
import numpy as np
import pytest

import oracle
import parallel_hnsw_amd as ph

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def keep_every_table(monkeypatch):
    monkeypatch.setenv("PHNSW_BUILD_TABLE_MIN", "1")


def layers_equal(g, o):
    assert g.layer_count() == o.layer_count
    for l in range(o.layer_count):
        nodes, nb = o.layer(l)
        gl = g._layer(l)
        np.testing.assert_array_equal(gl.nodes, nodes)
        np.testing.assert_array_equal(gl.neighbors, nb, err_msg="layer %d" % l)


@pytest.mark.parametrize("n,dim,metric,kw", [
    (6000, 64, 0, dict(seed=3)),
    (9000, 768, 0, dict(seed=1, max_link_rounds=2)),
    (3000, 100, 1, dict(seed=5, order=6, neighborhood_size=8, zero_layer_neighborhood_size=16)),
    (5000, 32, 2, dict(seed=2)),
])
def test_builds_with_kept_tables_equal_the_oracle(n, dim, metric, kw, monkeypatch):
    rows = oracle.synth_rows(0, n, dim, normalize=metric != 2)
    oix = oracle.Index.generate(rows, np.arange(n), oracle.default_build_params(**kw), dim=dim, metric=metric,
                                sum_mode=oracle.SUM_BLOCKED64)
    store = ph.VectorStore(rows[:, :dim], metric=metric)
    g = ph.Hnsw.generate(store, np.arange(n), ph.BuildParameters(**kw))
    layers_equal(g, oix)
    monkeypatch.setenv("PHNSW_NO_BUILD_TABLE", "1")
    g2 = ph.Hnsw.generate(store, np.arange(n), ph.BuildParameters(**kw))
    layers_equal(g2, oix)
    monkeypatch.delenv("PHNSW_NO_BUILD_TABLE")
    # improve_index / improve_neighbors on the finished index (a fresh scope, a fresh table)
    bp, obp = ph.BuildParameters(**kw), oracle.default_build_params(**kw)
    assert g.improve_index(bp, None) == pytest.approx(oix.improve_index(obp), abs=0)
    layers_equal(g, oix)


def test_promotion_changes_node_lists_under_the_kept_table():
    """duplicate-heavy data: rows cannot hold every copy, nodes stay unreachable, promote_at_layer extends and re-tops
    the upper layers while link rounds of the same build keep their tables: the epoch must void them"""
    base = oracle.synth_rows(0, 40, 16)
    rows = np.repeat(base, 60, axis=0).copy()
    n = rows.shape[0]

    def weak(mod):
        bp = mod(promote=1, seed=1, order=6, neighborhood_size=4, zero_layer_neighborhood_size=8)
        bp.optimization.recall_proportion = 1.0
        s = bp.optimization.search
        s.number_of_candidates, s.upper_layer_candidate_count = 16, 16
        return bp

    oix = oracle.Index.generate(rows, np.arange(n), weak(oracle.default_build_params), dim=16, sum_mode=oracle.SUM_BLOCKED64)
    store = ph.VectorStore(rows[:, :16])
    g = ph.Hnsw.generate(store, np.arange(n), weak(ph.BuildParameters))
    assert sum(g._layer(l).node_count() for l in range(g.layer_count() - 1)) > sum(oracle.calculate_partitions(n, 6)[:-1])
    layers_equal(g, oix)


@pytest.mark.parametrize("world,rank", [(2, 1), (4, 0)])
def test_sharded_builds_keep_only_their_range(world, rank):
    from parallel_hnsw_amd.sharded import EmulatedComm, build_sharded, sharded_tuning
    n, dim = 6000, 64
    store = ph.VectorStore.synthetic(n, dim, seed=42)
    ref = ph.Hnsw.generate(store, np.arange(n), ph.BuildParameters(seed=4))
    try:
        sharded_tuning(64, 2, 64)
        h, st = build_sharded(store, np.arange(n), ph.BuildParameters(seed=4), EmulatedComm(world, rank))
    finally:
        sharded_tuning(4096, 4, 65536)
    for l in range(ref.layer_count()):
        np.testing.assert_array_equal(h._layer(l).neighbors, ref._layer(l).neighbors, err_msg="layer %d" % l)
