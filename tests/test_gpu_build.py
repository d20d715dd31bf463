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


def test_generation_rows_recorded_by_the_reference():
    """test_generation (lib.rs:2090-2151) on the GPU: for the shuffles with which the oracle reproduces the seven
    brute-force-consistent rows of the reference's literal, the GPU build gives them too (and the whole graph is
    the oracle's); the exact neighbours behind them come out of the brute-force kernel in the same order"""
    g = TOY["test_generation"]
    lit, rows = np.array(g["neighbors"]), g["brute_force_consistent_rows"]
    b = TOY["vectors"]["build"]
    data = toy_vectors()
    store = ph.VectorStore(data, metric=ph.METRIC_ONE_MINUS_DOT)
    hit = 0
    for seed in range(16):
        obp_ = oracle.default_build_params(order=b["order"], neighborhood_size=3, zero_layer_neighborhood_size=6, seed=seed)
        oix = oracle.Index.generate(data, list(range(9)), obp_, metric=oracle.METRIC_ONE_MINUS_DOT, threads=1)
        onb = oix.layer(oix.layer_count - 1)[1].astype(np.int64)
        if not all((onb[i] == lit[i]).all() for i in rows):
            continue
        hit += 1
        h = ph.Hnsw.generate(store, np.arange(9), gbp(order=b["order"], neighborhood_size=3, zero_layer_neighborhood_size=6,
                                                     seed=seed))
        gnb = h.get_layer(0).neighbors.astype(np.int64)
        for i in rows:
            assert gnb[i].tolist() == lit[i].tolist(), (seed, i)
    assert hit >= 1
    ids, d = store.bruteforce_topk(data, 7)
    for i in rows:
        assert [int(x) for x in ids[i] if int(x) != i][:6] == lit[i].tolist(), i


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


def test_sharded_build_emulated_worlds_equal_phnsw_build():
    """phnsw_build_sharded (csrc/sharded.hip) with one process playing every rank in turn: each rank's node range
    runs as its own launches through the phase API, the blocks are laid out and reassembled as over RCCL, and the
    graph must be phnsw_build's and the oracle's -- for worlds that divide the layers unevenly, with the share of
    a rank cut into pieces (the sub-chunk pipeline), and with short lists kept whole"""
    from parallel_hnsw_amd.sharded import EmulatedComm, build_sharded, sharded_tuning
    n, dim = 3000, 64
    rows = oracle.synth_rows(0, n, dim)
    store = ph.VectorStore(rows[:, :dim])
    ref = ph.Hnsw.generate(store, np.arange(n), gbp(seed=4))
    oix = oracle.Index.generate(rows, np.arange(n), obp(seed=4), dim=dim, sum_mode=oracle.SUM_BLOCKED64)
    layers_equal(ref, oix)
    try:
        for world, rank, shard_min, subchunks, sub_min in ((2, 0, 1, 1, 1), (3, 1, 1, 4, 64), (8, 7, 256, 4, 16),
                                                           (5, 0, 4096, 4, 8192)):
            sharded_tuning(shard_min, subchunks, sub_min)
            h, st = build_sharded(store, np.arange(n), gbp(seed=4), EmulatedComm(world, rank))
            assert h.layer_count() == ref.layer_count()
            for l in range(ref.layer_count()):
                a, b = h._layer(l), ref._layer(l)
                np.testing.assert_array_equal(a.nodes, b.nodes)
                np.testing.assert_array_equal(a.neighbors, b.neighbors, err_msg="world %d layer %d" % (world, l))
            if shard_min < 4096:
                assert st["all_gather_calls"] > 0 and st["seconds_others"] > 0
            else:
                assert st["all_gather_calls"] == 0  # nothing at this size is long enough to split
    finally:
        sharded_tuning(4096, 4, 65536)


def test_rccl_transport_selftest_single_rank():
    """the library's own RCCL transport (phnsw_comm_rccl_*; librccl loaded on first use): on the one GPU of the
    test box a world of one rank still runs ncclCommInitRank, ncclAllGather on the library's stream and
    ncclAllReduce end to end"""
    import ctypes as C
    from parallel_hnsw_amd._lib import Comm, check
    ident = (C.c_uint8 * 128)()
    check(ph.lib().phnsw_comm_rccl_unique_id(ident))
    p = C.POINTER(Comm)()
    check(ph.lib().phnsw_comm_rccl_create(ident, 0, 1, 0, C.byref(p)))
    try:
        assert p.contents.world == 1 and p.contents.host_buffers == 0
        check(ph.lib().phnsw_comm_selftest(p, 1 << 20))
        host, total = C.c_double(), C.c_double()
        check(ph.lib().phnsw_comm_benchmark(p, 1 << 24, 50, C.byref(host), C.byref(total)))
        print("\nRCCL (one rank): ncclAllGather of 16 MiB: %.1f us of host time per call to enqueue, %.1f us per call in all"
              % (host.value, total.value))
        assert 0 < host.value < 1000 and total.value > 0
    finally:
        ph.lib().phnsw_comm_destroy(p)


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
        comm = ph.TorchComm()   # gloo: host callbacks, libphnsw stages its device blocks through pinned memory
        from parallel_hnsw_amd._lib import check
        check(ph.lib().phnsw_comm_selftest(comm.c_comm(0), 4096))
        before = comm.bytes_gathered
        b = ph.ShardedBuilder(eng, comm, shard_min=256)  # split all but the tiny layers
        h = b.generate(np.arange(n))
        assert b.stats["all_gather_bytes"] == comm.bytes_gathered - before > 0
        # PQ encode over the same two ranks: each encodes half the vectors, codes all-gathered
        pq = ph.PqStore(store, 16, 64, 3, comm=comm)
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), gathered=comm.bytes_gathered, codes=pq.codes(),
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
    codes = ph.PqStore(store, 16, 64, 3).codes()
    for r in range(2):
        z = np.load(str(tmp_path / ("r%d.npz" % r)))
        assert int(z["gathered"]) > 0
        for l in range(ref.layer_count()):
            np.testing.assert_array_equal(z["nb%d" % l], ref._layer(l).neighbors, err_msg="rank %d layer %d" % (r, l))
        np.testing.assert_array_equal(z["codes"], codes)


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
        comm = ph.TorchComm()   # nccl: the library's own RCCL communicator, the id travels through the group
        from parallel_hnsw_amd._lib import check
        check(ph.lib().phnsw_comm_selftest(comm.c_comm(rank), 1 << 20))
        # the bottom layer's share is cut into pieces whose all-gathers run on the collectives' stream
        b = ph.ShardedBuilder(eng, comm, shard_min=256, sub_min=1024)
        h = b.generate(np.arange(n))
        torch.cuda.synchronize()
        comm.close()
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), gathered=b.stats["all_gather_bytes"],
                 calls=b.stats["all_gather_calls"],
                 **{"nb%d" % l: h._layer(l).neighbors for l in range(h.layer_count())})
    finally:
        dist.destroy_process_group()


def test_sharded_build_over_rccl(tmp_path):
    """one process per GPU, phnsw_build_sharded over the library's RCCL transport (ncclAllGather over xGMI on the
    collectives' stream, pieces of the sub-chunk pipeline in flight): every rank ends with the graph one GPU
    builds.  Needs two GPUs; the one-GPU test boxes skip it (there: the RCCL self-test with one rank, the gloo
    rehearsal with two, and the emulated worlds above)."""
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
