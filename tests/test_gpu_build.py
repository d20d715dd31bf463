"""GPU parity for index construction: Hnsw::generate / generate_layer / link rounds /
stochastic recall through the C ABI against the oracle's deterministic build on the same
seeded inputs.  Graphs (integer work) must be identical, recall values equal."""
import numpy as np
import pytest

import oracle
import parallel_hnsw_amd as ph
from helpers import EMPTY, load, toy_vectors

pytestmark = pytest.mark.gpu

TOY = load("toy_index.json")


def layers_equal(gix, oix):
    assert gix.layer_count() == oix.layer_count
    for l in range(oix.layer_count):
        onodes, onb = oix.layer(l)
        gl = gix._layer(l)
        np.testing.assert_array_equal(gl.nodes, onodes)
        np.testing.assert_array_equal(gl.neighbors, onb, err_msg="layer %d" % l)


def obp(**kw):
    return oracle.default_build_params(**kw)


def gbp(**kw):
    return ph.BuildParameters(**kw)


@pytest.mark.parametrize("n,dim,metric,kw", [
    (600, 16, 0, dict(order=6, neighborhood_size=6, zero_layer_neighborhood_size=12)),
    (3000, 32, 0, dict()),
    (3000, 100, 0, dict(seed=5)),
    (2500, 32, 2, dict(seed=1)),                      # Euclidean comparator (lib.rs:2422-2441)
    (1500, 768, 0, dict(order=24)),
    (300, 8, 1, dict(order=400)),                     # single layer: all-pairs seeding only
])
def test_generate_parity(n, dim, metric, kw):
    normalize = metric != 2
    rows = oracle.synth_rows(0, n, dim, normalize=normalize)
    oix = oracle.Index.generate(rows, np.arange(n), obp(**kw), dim=dim, metric=metric,
                                sum_mode=oracle.SUM_BLOCKED64)
    store = ph.VectorStore(rows[:, :dim], metric=metric)
    gix = ph.Hnsw.generate(store, np.arange(n), gbp(**kw))
    layers_equal(gix, oix)
    assert oix.check_layer_invariants() == 0
    # same recall estimate (lib.rs:1463-1499)
    op = obp(**kw).optimization
    assert gix.stochastic_recall() == pytest.approx(oix.stochastic_recall_at(oix.layer_count - 1, op), abs=0)


def test_generate_subset_of_store_and_synthetic_store():
    """vids need not cover the store; the store can be generated on the device"""
    n, dim = 4000, 64
    rows = oracle.synth_rows(0, n, dim)
    vids = np.arange(0, n, 3)
    oix = oracle.Index.generate(rows, vids, obp(seed=3), dim=dim, sum_mode=oracle.SUM_BLOCKED64)
    store = ph.VectorStore.synthetic(n, dim, seed=42)
    gix = ph.Hnsw.generate(store, vids, gbp(seed=3))
    layers_equal(gix, oix)


def test_link_round_and_recall_parity_on_adopted_graph():
    """phnsw_index_from_layers + link rounds: rows come without stored distances"""
    n, dim = 3000, 48
    rows = oracle.synth_rows(0, n, dim)
    bp = obp(max_link_rounds=1)
    oix = oracle.Index(rows, dim=dim, sum_mode=oracle.SUM_BLOCKED64)
    oix.set_sum_mode(oracle.SUM_BLOCKED64)
    vs = oracle.shuffle(np.arange(n), 9)
    sizes = oracle.calculate_partitions(n, 12)
    for i, sz in enumerate(sizes):
        oix.generate_layer(vs[:sz], 48 if i == len(sizes) - 1 else 24, bp)
    store = ph.VectorStore(rows[:, :dim])
    gix = ph.Hnsw.from_layers(store, [oix.layer(l) for l in range(oix.layer_count)], gbp(max_link_rounds=1))
    layers_equal(gix, oix)
    sp = (300, 300, 2)
    for lft in range(oix.layer_count):
        oa = oix.link_layer(lft, sp, 24)
        ga = gix.link_layer_to_better_neighbors(lft, ph.SearchParameters(*sp))
        assert ga == oa
        layers_equal(gix, oix)
    op = bp.optimization
    for lft in range(oix.layer_count):
        assert gix.stochastic_recall_at(lft) == oix.stochastic_recall_at(lft, op)
    r_o = oix.improve_index(bp)
    r_g = gix.improve_index()
    assert r_g == r_o
    layers_equal(gix, oix)
    # improve_index(bp, Some(last_recall), ..)  lib.rs:1664-1671: the given recall replaces the first estimate
    r_o2 = oix.improve_index(bp, last_recall=r_o)
    r_g2 = gix.improve_index(last_recall=r_g)
    assert r_g2 == r_o2
    layers_equal(gix, oix)


def test_generate_layer_stepwise_parity():
    n, dim = 2000, 24
    rows = oracle.synth_rows(0, n, dim)
    bp_o, bp_g = obp(seed=11), gbp(seed=11)
    oix = oracle.Index(rows, dim=dim, sum_mode=oracle.SUM_BLOCKED64)
    oix.set_sum_mode(oracle.SUM_BLOCKED64)
    store = ph.VectorStore(rows[:, :dim])
    vs = oracle.shuffle(np.arange(n), 11)
    sizes = oracle.calculate_partitions(n, 12)
    oix.generate_layer(vs[:sizes[0]], 24, bp_o)
    gix = ph.Hnsw.from_layers(store, [oix.layer(0)], bp_g)
    for i, sz in enumerate(sizes[1:], 1):
        W = 48 if i == len(sizes) - 1 else 24
        oix.generate_layer(vs[:sz], W, bp_o)
        gix.generate_layer(vs[:sz], W)
        layers_equal(gix, oix)


def test_toy_index_properties():
    """make_simple_hnsw (lib.rs:1994-2015) built on the GPU: test_search / small improvement"""
    b = TOY["vectors"]["build"]
    data = toy_vectors()
    store = ph.VectorStore(data, metric=ph.METRIC_ONE_MINUS_DOT)
    for seed in (0, 1, 7):
        bp = gbp(order=b["order"], neighborhood_size=3, zero_layer_neighborhood_size=6, seed=seed)
        h = ph.Hnsw.generate(store, np.arange(9), bp)
        assert h.layer_count() == 2
        assert list(h.get_layer(0).nodes) == list(range(9))
        h.improve_index()
        for i in range(9):
            res = h.search(ph.Unstored(data[i]), ph.SearchParameters())
            assert res[0][0] == i  # test_small_index_improvement lib.rs:2270-2284
        obp_ = oracle.default_build_params(order=b["order"], neighborhood_size=3, zero_layer_neighborhood_size=6,
                                           seed=seed)
        oix = oracle.Index.generate(data, list(range(9)), obp_, metric=oracle.METRIC_ONE_MINUS_DOT,
                                    sum_mode=oracle.SUM_BLOCKED64, threads=1)
        layers_equal(h, oix)


def test_build_rejects_bad_input():
    data = toy_vectors()
    store = ph.VectorStore(data, metric=ph.METRIC_ONE_MINUS_DOT)
    with pytest.raises(ph.PhnswError):
        ph.Hnsw.generate(store, np.array([], dtype=np.uint64), gbp())   # assert!(total_size > 0) lib.rs:837
    with pytest.raises(ph.PhnswError):
        ph.Hnsw.generate(store, np.array([0, 1, 99], dtype=np.uint64), gbp())
    with pytest.raises(ph.PhnswError):
        ph.Hnsw.generate(store, np.array([0, 1, 1], dtype=np.uint64), gbp(order=2))
    with pytest.raises(ph.PhnswError):
        ph.Hnsw.generate(store, np.arange(9), gbp(zero_layer_neighborhood_size=65))


def test_sharded_builder_single_rank_equals_phnsw_build():
    """parallel_hnsw_amd.sharded (the multi-GPU driver) on one rank: same phases, same graph
    as phnsw_build and as the oracle; the 2-rank split itself is covered on CPU (gloo)"""
    n, dim = 3000, 64
    rows = oracle.synth_rows(0, n, dim)
    store = ph.VectorStore(rows[:, :dim])
    eng = ph.GpuEngine(store, gbp(seed=4))

    class OneRank:
        rank, world, bytes_gathered = 0, 1, 0

        def all_gather(self, t):
            return t

        def all_reduce_sum(self, v, device):
            return list(v)

    h = ph.ShardedBuilder(eng, OneRank()).generate(np.arange(n))
    ref = ph.Hnsw.generate(store, np.arange(n), gbp(seed=4))
    assert h.layer_count() == ref.layer_count()
    for l in range(ref.layer_count()):
        a, b = h._layer(l), ref._layer(l)
        np.testing.assert_array_equal(a.nodes, b.nodes)
        np.testing.assert_array_equal(a.neighbors, b.neighbors)
    oix = oracle.Index.generate(rows, np.arange(n), obp(seed=4), dim=dim, sum_mode=oracle.SUM_BLOCKED64)
    layers_equal(h, oix)


def test_phase_api_ranges_compose():
    """two half ranges through the phase API == one full range (what two GPUs would compute)"""
    import torch
    n, dim = 2000, 32
    rows = oracle.synth_rows(0, n, dim)
    store = ph.VectorStore(rows[:, :dim])
    ref = ph.Hnsw.generate(store, np.arange(n), gbp(seed=2))

    class TwoHalves:
        """runs the driver twice per phase on one GPU by faking rank 0 then rank 1"""
        bytes_gathered = 0

        def __init__(self):
            self.rank, self.world = 0, 1

        def all_gather(self, t):
            return t

        def all_reduce_sum(self, v, device):
            return list(v)

    eng = ph.GpuEngine(store, gbp(seed=2))
    b = ph.ShardedBuilder(eng, TwoHalves())
    orig_range = b._range

    # monkeypatch the phase calls to split every range in two launches
    def split(fn, first_idx, count_idx):
        def wrapped(*a):
            a = list(a)
            first, count = a[first_idx], a[count_idx]
            h1 = count // 2
            outs = [x for x in a if isinstance(x, torch.Tensor) and x.shape[0] == count and x is not None]
            a1 = list(a); a1[count_idx] = h1
            a2 = list(a); a2[first_idx] = first + h1; a2[count_idx] = count - h1
            # output tensors are the trailing tensor args: give the second half offset views
            for i, x in enumerate(a):
                if i > count_idx and isinstance(x, torch.Tensor):
                    a1[i] = x[:h1]
                    a2[i] = x[h1:]
            fn(*a1)
            fn(*a2)
        return wrapped

    eng.layer_init_search = split(eng.layer_init_search, 0, 1)
    eng.link_search = split(eng.link_search, 3, 4)
    seed0 = eng.layer_seed

    def seed_split(ids, d, ln, first, count, rows_, rows_d):
        h1 = count // 2
        seed0(ids, d, ln, first, h1, rows_[:h1], rows_d[:h1])
        seed0(ids, d, ln, first + h1, count - h1, rows_[h1:], rows_d[h1:])

    eng.layer_seed = seed_split
    h = b.generate(np.arange(n))
    for l in range(ref.layer_count()):
        np.testing.assert_array_equal(h._layer(l).neighbors, ref._layer(l).neighbors)


def _two_rank_worker(rank, world, port, out_dir):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, dim = 3000, 64
        store = ph.VectorStore.synthetic(n, dim, seed=42)
        eng = ph.GpuEngine(store, ph.BuildParameters(seed=6))
        comm = ph.TorchComm()
        h = ph.ShardedBuilder(eng, comm, shard_min=256).generate(np.arange(n))  # split all but the tiny layers
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), gathered=comm.bytes_gathered,
                 **{"nb%d" % l: h._layer(l).neighbors for l in range(h.layer_count())})
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_rehearsal(tmp_path):
    """two processes share the one GPU of the test box and exchange through gloo (RCCL needs
    one GPU per rank); every device-side piece of the sharded build runs with real ranges"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    n, dim = 3000, 64
    store = ph.VectorStore.synthetic(n, dim, seed=42)
    ref = ph.Hnsw.generate(store, np.arange(n), ph.BuildParameters(seed=6))
    for r in range(2):
        z = np.load(str(tmp_path / ("r%d.npz" % r)))
        assert int(z["gathered"]) > 0
        for l in range(ref.layer_count()):
            np.testing.assert_array_equal(z["nb%d" % l], ref._layer(l).neighbors, err_msg="rank %d layer %d" % (r, l))


def _dup_rows(points=40, copies=60, dim=16):
    base = oracle.synth_rows(0, points, dim)
    return np.repeat(base, copies, axis=0).copy()


def _weak_bp(mod, promote=1):
    bp = mod(promote=promote, seed=1, order=6, neighborhood_size=4, zero_layer_neighborhood_size=8)
    bp.optimization.recall_proportion = 1.0
    s = bp.optimization.search
    s.number_of_candidates, s.upper_layer_candidate_count = 16, 16
    return bp


def test_promotion_parity():
    """promote_at_layer / extend_layer / re-topping (lib.rs:1002-1068, 1167-1427) driven by GPU
    searches: same layers as the oracle on data whose duplicates leave nodes unreachable"""
    rows = _dup_rows()
    n = rows.shape[0]
    oix = oracle.Index.generate(rows, np.arange(n), _weak_bp(obp), dim=16, sum_mode=oracle.SUM_BLOCKED64)
    store = ph.VectorStore(rows[:, :16])
    gix = ph.Hnsw.generate(store, np.arange(n), _weak_bp(gbp))
    sizes = [gix._layer(l).node_count() for l in range(gix.layer_count())]
    assert sum(sizes[:-1]) > sum(oracle.calculate_partitions(n, 6)[:-1])  # promotion really happened
    layers_equal(gix, oix)
    assert oix.check_layer_invariants() == 0
    # the pieces one by one on a fresh (unpromoted) stack
    o0 = oracle.Index.generate(rows, np.arange(n), _weak_bp(obp, 0), dim=16, sum_mode=oracle.SUM_BLOCKED64)
    g0 = ph.Hnsw.generate(store, np.arange(n), _weak_bp(gbp, 0))
    layers_equal(g0, o0)
    sp = (16, 16, 2)
    for lft in range(o0.layer_count):
        np.testing.assert_array_equal(g0.discover_unreachable_vectors(lft, ph.SearchParameters(*sp)),
                                      o0.discover_unreachable(lft, sp))
    lft = o0.layer_count - 1
    assert g0.promote_at_layer(lft, _weak_bp(gbp)) == (o0.promote_at_layer(lft, _weak_bp(obp)) > 0)
    layers_equal(g0, o0)
    # searching the promoted stack still agrees (new upper-layer nodes have empty rows until linked)
    q = oracle.synth_rows(2 ** 32, 64, 16)[:, :16]
    gi, gd, gl = g0.search_batch(queries=q, sp=ph.SearchParameters(32, 32, 2))
    ci, cd, cl = o0.search(queries=q, sp=(32, 32, 2))
    np.testing.assert_array_equal(gi, ci)
    np.testing.assert_array_equal(gd.view(np.uint32), cd.view(np.uint32))


def test_baseline_config1_shape():
    """BASELINE configs[0] (the reference's own bench shape, benches/bench.rs:9,24-30,54-63):
    10 000 x 128 f32 (and the bench's literal 100 dims), cosine / 1-dot, default build, search
    at number_of_candidates = 64 -- GPU and oracle end to end"""
    for dim, metric, normalize in ((128, 0, True), (100, 1, False)):
        n = 10000
        rows = oracle.synth_rows(0, n, dim, normalize=normalize)
        if not normalize:
            rows = np.abs(rows)  # bench.rs draws rng.gen() in [0, 1), un-normalised
        oix = oracle.Index.generate(rows, np.arange(n), obp(seed=7), dim=dim, metric=metric,
                                    sum_mode=oracle.SUM_BLOCKED64)
        store = ph.VectorStore(rows[:, :dim], metric=metric)
        gix = ph.Hnsw.generate(store, np.arange(n), gbp(seed=7))
        layers_equal(gix, oix)
        q = np.abs(oracle.synth_rows(2 ** 32, 300, dim, normalize=normalize))[:, :dim] if not normalize else \
            oracle.synth_rows(2 ** 32, 300, dim)[:, :dim]
        gi, gd, gl, gs = gix.search_batch(queries=q, sp=ph.SearchParameters(64, 64, 2), stats=True)
        ci, cd, cl, cs = oix.search(queries=q, sp=(64, 64, 2), stats=True)
        np.testing.assert_array_equal(gi, ci)
        np.testing.assert_array_equal(gd.view(np.uint32), cd.view(np.uint32))
        np.testing.assert_array_equal(gs, cs)


def _nccl_worker(rank, world, port, out_dir):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        n, dim = 40000, 64
        store = ph.VectorStore.synthetic(n, dim, seed=42, device=rank)
        eng = ph.GpuEngine(store, ph.BuildParameters(seed=6, max_link_rounds=1), device=torch.device("cuda", rank))
        comm = ph.TorchComm()
        b = ph.ShardedBuilder(eng, comm, shard_min=256)
        b.SUB_MIN = 1024  # the bottom layer's share is cut into pieces whose all-gathers run asynchronously
        h = b.generate(np.arange(n))
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), gathered=comm.bytes_gathered, calls=comm.calls,
                 **{"nb%d" % l: h._layer(l).neighbors for l in range(h.layer_count())})
    finally:
        dist.destroy_process_group()


def test_sharded_build_over_rccl(tmp_path):
    """one process per GPU, torch.distributed 'nccl' (= RCCL over xGMI), asynchronous all-gathers of the
    sub-chunk pipeline: every rank ends with the graph one GPU builds.  Needs two GPUs; the one-GPU test
    boxes skip it (the gloo rehearsal above covers the device side there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs at least two GPUs (RCCL: one rank per GPU)")
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    mp.spawn(_nccl_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    n, dim = 40000, 64
    store = ph.VectorStore.synthetic(n, dim, seed=42)
    ref = ph.Hnsw.generate(store, np.arange(n), ph.BuildParameters(seed=6, max_link_rounds=1))
    for r in range(world):
        z = np.load(str(tmp_path / ("r%d.npz" % r)))
        assert int(z["gathered"]) > 0 and int(z["calls"]) > 4
        for l in range(ref.layer_count()):
            np.testing.assert_array_equal(z["nb%d" % l], ref._layer(l).neighbors, err_msg="rank %d layer %d" % (r, l))
