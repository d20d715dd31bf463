"""The CPU oracle against every golden vector the reference's own unit tests hold for the
hot path.  CPU only."""
import numpy as np
import pytest

import oracle
from helpers import EMPTY, load, sub, toy_vectors

PQ = load("priority_queue.json")["cases"]
TOY = load("toy_index.json")


@pytest.mark.parametrize("case", PQ, ids=[c["name"] for c in PQ])
def test_priority_queue_golden(case):
    q = oracle.PriorityQueue(sub(case["data"]), sub(case["priorities"]))
    ret = None
    if case["op"] == "insert":
        q.insert(case["elt"], case["priority"])
    elif case["op"] == "merge":
        ret = q.merge(case["ids"], case["prios"])
    elif case["op"] == "last":
        last = q.last()
        assert last[0] == case["expect_last"][0]
        assert last[1] == np.float32(case["expect_last"][1])
    if "expect_return" in case:
        assert ret == case["expect_return"]
    if "expect_data" in case:
        assert [int(x) for x in q.data] == sub(case["expect_data"])
    if "expect_priorities" in case:
        assert list(q.priorities) == [np.float32(x) for x in sub(case["expect_priorities"])]


@pytest.mark.parametrize("case", TOY["test_final_idx"]["cases"])
def test_final_idx(case):
    assert oracle.final_neighbor_idx(case["neighborhood_size"], sub(case["neighbors"]), case["n"]) == case["expect"]


@pytest.mark.parametrize("case", TOY["partitions"]["cases"], ids=lambda c: c["ref"])
def test_partitions(case):
    if case["fn"] == "calculate_partitions":
        p = oracle.calculate_partitions(case["total"], case["order"])
        if "expect_len" in case:
            assert len(p) == case["expect_len"]
        else:
            assert p == case["expect"]
    else:
        sf = case["sizes_from"]
        sizes = oracle.calculate_partitions(sf["total"], sf["order"])
        sizes.reverse()
        assert oracle.calculate_partitions_for_additions(sizes[sf["skip_first"]:], case["new_vecs"],
                                                         case["order"]) == case["expect"]


def make_simple_hnsw(sum_mode=oracle.SUM_SEQ, broken=False, seed=0):
    """make_simple_hnsw lib.rs:1994-2015 through the oracle's deterministic generate"""
    b = TOY["vectors"]["build"]
    bp = oracle.default_build_params(order=b["order"], neighborhood_size=b["neighborhood_size"],
                                     zero_layer_neighborhood_size=b["zero_layer_neighborhood_size"], seed=seed)
    data = toy_vectors(broken)
    ix = oracle.Index.generate(data, list(range(9)), bp, metric=oracle.METRIC_ONE_MINUS_DOT, sum_mode=sum_mode,
                               threads=1)
    return ix, bp


def fixture_hnsw(entry, sum_mode=oracle.SUM_SEQ):
    """the reference's own test graph (test_generation literal) under a 1-node top layer"""
    g = TOY["test_generation"]
    ix = oracle.Index(toy_vectors(), metric=oracle.METRIC_ONE_MINUS_DOT, sum_mode=sum_mode)
    ix.set_sum_mode(sum_mode)
    ix.push_layer([entry], np.full((1, 3), EMPTY, dtype=np.uint64), 3)
    ix.push_layer(list(range(9)), np.array(g["neighbors"], dtype=np.uint64), g["neighborhood_size"])
    return ix


@pytest.mark.parametrize("entry", TOY["test_generation"]["nearness_entries"])
@pytest.mark.parametrize("sum_mode", [oracle.SUM_SEQ, oracle.SUM_BLOCKED64])
def test_nearness_search(entry, sum_mode):
    ix = fixture_hnsw(entry, sum_mode)
    t = TOY["test_nearness_search"]
    ids, d, ln = ix.search(queries=[sub(t["query"])], sp=tuple(t["search"]), threads=1)
    n = int(ln[0])
    got = [(int(ids[0, i]), d[0, i]) for i in range(n)]
    exp = [(e[0], np.float32(e[1])) for e in t["expect"]]
    if sum_mode == oracle.SUM_SEQ:
        assert got == exp  # bit-exact f32 distances, (d, id) tie order
    else:
        assert [g[0] for g in got] == [e[0] for e in exp]
        np.testing.assert_allclose([g[1] for g in got], [e[1] for e in exp], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("seed", [0, 1, 7])
def test_search_finds_self(seed):
    """test_search lib.rs:2154-2164 on the oracle's own build"""
    ix, bp = make_simple_hnsw(seed=seed)
    data = toy_vectors()
    ids, d, ln = ix.search(queries=data, sp=(300, 300, 2), threads=1)
    for i in range(9):
        # match_within_epsilon  search.rs:173-187.  Vector 4 = [0.5773]*3 is not unit length
        # (self distance 1.7e-4 > epsilon), so the reference's own assertion cannot hold for
        # it; there the top-1 hit is checked instead.
        if i == 4:
            assert int(ids[i, 0]) == 4
            continue
        lead = []
        for j in range(int(ln[i])):
            if abs(d[i, j]) < 1e-5:
                lead.append(int(ids[i, j]))
            else:
                break
        assert i in lead


def test_generation_rows_recorded_by_the_reference():
    """test_generation (lib.rs:2090-2151): seven of the nine bottom-layer rows the reference recorded are the exact
    six nearest neighbours in (distance, id) order -- they pin the metric (1 - dot, sequential f32), the ordering and
    the tie-break of a built row (four of them hold equal distances: ids decide).  The oracle's exact k-NN must give
    those rows literally, and its own generate must reproduce all seven for a shuffle that lets every node discover
    its neighbourhood (which nodes meet is shuffle-dependent: thread_rng in the reference, lib.rs:832)."""
    g = TOY["test_generation"]
    lit, rows = np.array(g["neighbors"]), g["brute_force_consistent_rows"]
    data = toy_vectors()
    ix = oracle.Index(data, metric=oracle.METRIC_ONE_MINUS_DOT, sum_mode=oracle.SUM_SEQ)
    ids, d = ix.bruteforce(data, 7, threads=1)  # self first (distance <= everything), then the six neighbours
    for i in rows:
        got = [int(x) for x in ids[i] if int(x) != i][:6]
        assert got == lit[i].tolist(), i
    assert any((d[i, 1:-1] == d[i, 2:]).any() for i in rows)  # ties are really exercised
    reproduced = []
    for seed in range(16):
        built, _ = make_simple_hnsw(seed=seed)
        nb = built.layer(built.layer_count - 1)[1].astype(np.int64)
        reproduced.append(all((nb[i] == lit[i]).all() for i in rows))
    assert any(reproduced), "no shuffle reproduces the seven recorded rows"


@pytest.mark.parametrize("entry", range(9))
def test_knn(entry):
    ix = fixture_hnsw(entry)
    t = TOY["test_knn"]
    ids, d, ln = ix.knn(t["k"], t["probe_depth"], threads=1)
    for i, exp in enumerate(t["expect"]):
        got = [(int(ids[i, j]), d[i, j]) for j in range(int(ln[i]))]
        assert got == [(e[0], np.float32(e[1])) for e in exp]


@pytest.mark.parametrize("entry", [0, 4])
def test_threshold_nn(entry):
    ix = fixture_hnsw(entry)
    t = TOY["test_threshold_nn"]
    ids, d, ln = ix.threshold_nn(t["threshold"], t["probe_depth"], t["initial_search_depth"], threads=1)
    for i, exp in enumerate(t["expect"]):
        got = [(int(ids[i, j]), d[i, j]) for j in range(int(ln[i]))]
        assert got == [(e[0], np.float32(e[1])) for e in exp], i


def test_small_index_improvement():
    ix, bp = make_simple_hnsw()
    ix.improve_index(bp, threads=1)
    ids, d, ln = ix.search(queries=toy_vectors(), sp=(300, 300, 2), threads=1)
    assert [int(ids[i, 0]) for i in range(9)] == list(range(9))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_tiny_index_improvement(seed):
    """make_broken_hnsw lib.rs:2017-2044 + test_tiny_index_improvement lib.rs:2286-2298: a 10th node
    with an empty row pushed onto the bottom layer is repaired by improve_index.  The
    reference samples max(1, 10 * 0.1) = 1 vector for its recall estimate (lib.rs:1473-1483), so
    whether its test proceeds past "recall == 1.0" depends on rand's shuffle (unpinned); here
    the estimate samples every vector, which makes the repair deterministic."""
    b = TOY["vectors"]["build"]
    bp = oracle.default_build_params(order=b["order"], neighborhood_size=3, zero_layer_neighborhood_size=6, seed=seed)
    bp.optimization.recall_proportion = 1.0
    data = toy_vectors(broken=True)
    ix = oracle.Index.generate(data, list(range(9)), bp, metric=oracle.METRIC_ONE_MINUS_DOT, threads=1)
    uppers = [ix.layer(l) for l in range(ix.layer_count - 1)]
    nodes, nb = ix.layer(ix.layer_count - 1)
    nodes = np.concatenate([nodes, [9]]).astype(np.uint64)
    nb = np.concatenate([nb, np.full((1, 6), EMPTY, dtype=np.uint64)])
    ix2 = oracle.Index(data, metric=oracle.METRIC_ONE_MINUS_DOT)
    for u in uppers:
        ix2.push_layer(u[0], u[1], 3)
    ix2.push_layer(nodes, nb, 6)
    assert 9 in ix2.discover_unreachable(ix2.layer_count - 1, (300, 300, 2), threads=1)
    ix2.improve_index(bp, threads=1)
    ids, d, ln = ix2.search(queries=data, sp=(300, 300, 2), threads=1)
    assert [int(ids[i, 0]) for i in range(10)] == list(range(10))


def dup_rows(points=40, copies=60, dim=16):
    base = oracle.synth_rows(0, points, dim)
    return np.repeat(base, copies, axis=0).copy()


def test_promotion_extends_upper_layers():
    """promote_at_layer (lib.rs:1273-1427): with more exact duplicates than a row can hold, nodes
    stay unreachable; promotion thins them spatially, extends the layers above (or re-tops the
    stack) and keeps the layer invariants (search.rs:142-171)."""
    rows = dup_rows()
    n = rows.shape[0]
    res = {}
    for pr in (0, 1):
        bp = oracle.default_build_params(promote=pr, seed=1, order=6, neighborhood_size=4,
                                         zero_layer_neighborhood_size=8)
        bp.optimization.recall_proportion = 1.0
        s = bp.optimization.search
        s.number_of_candidates, s.upper_layer_candidate_count = 16, 16
        ix = oracle.Index.generate(rows, np.arange(n), bp, dim=16, threads=4)
        assert ix.check_layer_invariants() == 0
        res[pr] = ([ix.layer(l)[0].shape[0] for l in range(ix.layer_count)],
                   ix.stochastic_recall_at(ix.layer_count - 1, bp.optimization))
    assert res[0][0] == oracle.calculate_partitions(n, 6)
    assert sum(res[1][0][:-1]) > sum(res[0][0][:-1])   # upper layers grew
    assert res[1][1] > res[0][1]                       # and self-recall improved


def test_extend_layer_renumbers_rows():
    """extend_layer lib.rs:1039-1068: new nodes get empty rows, old rows keep their neighbours
    under the new numbering; inserting an existing vector is refused (lib.rs:1797)"""
    rows = oracle.synth_rows(0, 50, 8)
    ix = oracle.Index(rows, dim=8)
    nodes = np.array([3, 10, 20, 30], dtype=np.uint64)
    nb = np.array([[1, 2, EMPTY], [0, 3, EMPTY], [0, EMPTY, EMPTY], [1, EMPTY, EMPTY]], dtype=np.uint64)
    ix.push_layer(nodes, nb, 3)
    assert ix.extend_layer(0, [15, 1]) == 0
    n2, nb2 = ix.layer(0)
    assert list(n2) == [1, 3, 10, 15, 20, 30]
    assert [int(x) for x in nb2[1]] == [2, 4, EMPTY]     # node 3: neighbours 10, 20 under new ids
    assert [int(x) for x in nb2[5]] == [2, EMPTY, EMPTY]  # node 30 -> 10
    assert (nb2[0] == EMPTY).all() and (nb2[3] == EMPTY).all()
    assert ix.extend_layer(0, [10]) != 0


def test_layer_invariants_and_rows():
    ix, _ = make_simple_hnsw()
    assert ix.check_layer_invariants() == 0
    assert ix.layer_count == 2
    nodes, nb = ix.layer(1)
    assert list(nodes) == list(range(9))
    # rows hold no self loops, no duplicates, trailing-only sentinels
    for i in range(9):
        row = [int(x) for x in nb[i]]
        live = [x for x in row if x != EMPTY]
        assert row[:len(live)] == live
        assert i not in live and len(set(live)) == len(live)


def test_bench_shape_plumbing():
    """BASELINE configs[0] on the CPU alone (benches/bench.rs:54-63 benchmarks Hnsw::generate on
    10 000 x 100 un-normalised vectors with 1 - dot): the oracle builds it, the layer invariants
    hold and the self-recall the reference's recall tests assert (>= 0.9 after generate,
    lib.rs:2218-2224) is met"""
    n, dim = 10000, 100
    rows = np.abs(oracle.synth_rows(0, n, dim, normalize=False))
    bp = oracle.default_build_params(seed=0)
    ix = oracle.Index.generate(rows, np.arange(n), bp, dim=dim, metric=oracle.METRIC_ONE_MINUS_DOT, threads=8)
    assert [ix.layer(l)[0].shape[0] for l in range(ix.layer_count)][-1] == n
    assert ix.layer_count >= 4 and ix.check_layer_invariants() == 0
    nodes, nb = ix.layer(ix.layer_count - 1)
    assert nb.shape == (n, 48)
    # un-normalised 1 - dot is not a metric (a longer vector can beat the stored one itself), so
    # self-recall is measured the reference's way on normalised data of the same shape
    rows2 = oracle.synth_rows(0, n, dim)
    ix2 = oracle.Index.generate(rows2, np.arange(n), bp, dim=dim, threads=8)
    ids, d, ln = ix2.search(qids=np.arange(0, n, 10), sp=(300, 300, 2), threads=8)
    assert np.mean(ids[:, 0] == np.arange(0, n, 10)) >= 0.9
