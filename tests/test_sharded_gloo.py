"""The N>1 build path on CPU: libphnsw's sharded driver (csrc/sharded.hip, the loop behind phnsw_build_sharded)
run through phnsw_build_sharded_engine with the ORACLE's phases as the engine -- under world_size-2 and -3 `gloo`
groups (host-callback communicator) and with an emulated world.  Checks the range split, the block layout,
the pieces of the sub-chunk pipeline, the reassembly and the replicated control flow: every rank must end with
exactly the graph a single process builds."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def make_oracle_engine(rows, dim, bp, metric=0, threads=2):
    """the oracle's phases as a phnsw_shard_engine (u64 ids, host buffers): the C++ driver of libphnsw calls back
    into these, so the split / block layout / reassembly / control flow under test are the product's own"""
    import oracle
    from parallel_hnsw_amd.sharded import PythonEngine

    class OracleEngine(PythonEngine):
        id_bytes = 8

        def __init__(self):
            super().__init__()
            self.oracle = oracle
            self.ix = oracle.Index(rows, dim=dim, metric=metric, sum_mode=oracle.SUM_BLOCKED64)
            self.ix.set_sum_mode(oracle.SUM_BLOCKED64)
            self.bp, self.threads = bp, threads
            self.L = oracle.lib()

        def _sp(self, sp):
            return type(self.bp.optimization.search).from_buffer_copy(bytes(sp))

        def plan(self, vids):
            vs = oracle.shuffle(np.asarray(vids, dtype=np.uint64), self.bp.seed)
            return vs, oracle.calculate_partitions(len(vs), self.bp.order)

        def layer_begin(self, vids, W):
            v = np.ascontiguousarray(vids, dtype=np.uint64)
            assert self.L.orc_layer_begin(self.ix.h, v.ctypes.data_as(C.c_void_p), len(v), W, C.byref(self.bp)) == 0
            return True, int(self.L.orc_layer_init_stride(self.ix.h))

        def layer_init_search(self, first, count, ids, d, ln):
            assert self.L.orc_layer_init_search(self.ix.h, C.byref(self.bp), first, count, ids, d, ln, self.threads) == 0

        def layer_seed(self, ids, d, ln, first, count, rows_, rows_d):
            assert self.L.orc_layer_seed(self.ix.h, C.byref(self.bp), ids, d, ln, first, count, rows_, rows_d,
                                         self.threads) == 0

        def layer_finish(self, rows_, rows_d):
            assert self.L.orc_layer_finish(self.ix.h, rows_, rows_d, self.threads) == 0

        def layer_count(self):
            return self.ix.layer_count

        def layer_nodes(self, lft):
            return int(self.L.orc_index_layer(self.ix.h, lft).contents.node_count)

        def link_search(self, lft, sp, M, first, count, ids, d, ln):
            assert self.L.orc_link_search(self.ix.h, lft, self._sp(sp), M, first, count, ids, d, ln, self.threads) == 0

        def link_apply(self, lft, M, ids, d, ln):
            return int(self.L.orc_link_apply(self.ix.h, lft, M, ids, d, ln, self.threads))

        def discover_hits(self, lft, sp, first, count, hit):
            assert self.L.orc_discover_hits(self.ix.h, lft, self._sp(sp), first, count, hit, self.threads) == 0

        def promote_from_hits(self, lft, hit):
            return self.L.orc_promote_at_layer_hits(self.ix.h, lft, C.byref(self.bp), hit, self.threads) > 0

        def recall_hits(self, at, op, first, count):
            hits, sel = C.c_uint64(), C.c_uint64()
            o = type(self.bp.optimization).from_buffer_copy(bytes(op))
            assert self.L.orc_recall_hits(self.ix.h, at, C.byref(o), first, count, C.byref(hits), C.byref(sel),
                                          self.threads) == 0
            return hits.value, sel.value

    return OracleEngine()


CASES = [
    dict(n=700, dim=16, kw=dict(order=6, neighborhood_size=6, zero_layer_neighborhood_size=12, seed=3)),
    dict(n=1501, dim=24, kw=dict(seed=1)),   # odd size: ranges of unequal length
    # short work lists stay whole on every rank (no collective), the bottom layer is split
    dict(n=1500, dim=16, kw=dict(seed=2), shard_min=400),
    dict(n=1000, dim=16, kw=dict(seed=4), world=3),   # three ranks: 334 + 334 + 332
    # duplicate-heavy data: rows cannot hold every copy, nodes stay unreachable, promotion
    # (lib.rs:1273-1427) extends and re-tops the upper layers while the build is sharded
    dict(n=1200, dim=16, dup=40, kw=dict(order=6, neighborhood_size=4, zero_layer_neighborhood_size=8, seed=1),
         search=(16, 16, 2), recall_proportion=1.0),
]


def _case_rows(case):
    import oracle
    if case.get("dup"):
        base = oracle.synth_rows(0, case["n"] // case["dup"], case["dim"])
        return np.repeat(base, case["dup"], axis=0).copy()
    return oracle.synth_rows(0, case["n"], case["dim"])


def _case_bp(case):
    import oracle
    bp = oracle.default_build_params(**case["kw"])
    if "search" in case:
        s = bp.optimization.search
        s.number_of_candidates, s.upper_layer_candidate_count, s.probe_depth = case["search"]
    if "recall_proportion" in case:
        bp.optimization.recall_proportion = case["recall_proportion"]
    return bp


def _worker(rank, world, port, case, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import oracle
    from parallel_hnsw_amd.sharded import ShardedBuilder, TorchComm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = _case_rows(case)
        bp = _case_bp(case)
        eng = make_oracle_engine(rows, case["dim"], bp)
        comm = TorchComm()
        b = ShardedBuilder(eng, comm, shard_min=case.get("shard_min", 0))
        b.generate(np.arange(case["n"], dtype=np.uint64))
        assert b.stats["all_gather_calls"] > 0 and b.stats["all_gather_bytes"] == comm.bytes_gathered
        layers = [eng.ix.layer(l) for l in range(eng.ix.layer_count)]
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), count=len(layers), gathered=comm.bytes_gathered,
                 **{"nodes%d" % i: l[0] for i, l in enumerate(layers)},
                 **{"nb%d" % i: l[1] for i, l in enumerate(layers)})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d%s%s" % (c["n"], "-min%d" % c["shard_min"] if "shard_min" in c else "",
                                                                   "-w%d" % c["world"] if "world" in c else ""))
def test_sharded_build_equals_single_process(case, tmp_path):
    import torch.multiprocessing as mp
    import oracle
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = case.get("world", 2)
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    rows = _case_rows(case)
    ref = oracle.Index.generate(rows, np.arange(case["n"]), _case_bp(case),
                                dim=case["dim"], sum_mode=oracle.SUM_BLOCKED64, threads=4)
    if case.get("dup"):
        assert ref.layer_count > len(oracle.calculate_partitions(case["n"], case["kw"]["order"])) or \
            ref.layer(ref.layer_count - 2)[0].shape[0] > case["n"] // case["kw"]["order"]  # promotion happened
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert int(z["count"]) == ref.layer_count
        assert int(z["gathered"]) > 0
        for l in range(ref.layer_count):
            nodes, nb = ref.layer(l)
            np.testing.assert_array_equal(z["nodes%d" % l], nodes)
            np.testing.assert_array_equal(z["nb%d" % l], nb, err_msg="rank %d layer %d" % (rank, l))


def _emulated_build(case, world, rank=0, subchunks=None, sub_min=None):
    """the C++ driver with an emulated world (one process plays every rank in turn) over the oracle's phases"""
    from parallel_hnsw_amd.sharded import EmulatedComm, ShardedBuilder, sharded_tuning
    eng = make_oracle_engine(_case_rows(case), case["dim"], _case_bp(case), threads=4)
    b = ShardedBuilder(eng, EmulatedComm(world, rank), shard_min=case.get("shard_min", 0), subchunks=subchunks,
                       sub_min=sub_min)
    try:
        b.generate(np.arange(case["n"], dtype=np.uint64))
    finally:
        sharded_tuning(4096, 4, 65536)
    return eng, b.stats


@pytest.mark.parametrize("world,rank,subchunks", [(2, 0, 1), (3, 2, 4), (8, 5, 4), (5, 0, 3)])
def test_emulated_world_equals_single_process(world, rank, subchunks):
    """every range split (unequal tails, empty tail ranks), the block layout with several pieces per rank and
    the reassembly copies: an emulated world of w ranks must end with the single-process graph"""
    import oracle
    case = dict(n=1501, dim=24, kw=dict(seed=1))
    eng, st = _emulated_build(case, world, rank, subchunks=subchunks, sub_min=16)
    ref = oracle.Index.generate(_case_rows(case), np.arange(case["n"]), _case_bp(case), dim=case["dim"],
                                sum_mode=oracle.SUM_BLOCKED64, threads=4)
    assert eng.ix.layer_count == ref.layer_count
    for l in range(ref.layer_count):
        nodes, nb = ref.layer(l)
        gn, gnb = eng.ix.layer(l)
        np.testing.assert_array_equal(gn, nodes)
        np.testing.assert_array_equal(gnb, nb, err_msg="layer %d" % l)
    assert st["phases"] > st["phases_whole"] >= 0
    assert st["all_gather_calls"] + st["all_reduce_calls"] >= st["phases"] - st["phases_whole"]
    assert st["seconds_others"] > 0 and st["seconds_sharded"] > 0


def test_short_lists_run_whole_without_collectives():
    """work lists below shard_min are not split: no all-gather, no all-reduce for them"""
    case = dict(n=700, dim=16, kw=dict(order=6, neighborhood_size=6, zero_layer_neighborhood_size=12, seed=3),
                shard_min=10 ** 6)
    eng, st = _emulated_build(case, 4)
    assert st["all_gather_calls"] == 0 and st["all_reduce_calls"] == 0 and st["phases"] == st["phases_whole"] > 0


def test_driver_refuses_bad_worlds():
    import parallel_hnsw_amd as ph
    from parallel_hnsw_amd._lib import Comm
    from parallel_hnsw_amd.sharded import ShardedBuilder

    class Bad:
        rank, world = 3, 2

        def c_comm(self, device=0):
            return Comm(rank=3, world=2, emulate=1)

    case = dict(n=300, dim=8, kw=dict(seed=1))
    eng = make_oracle_engine(_case_rows(case), case["dim"], _case_bp(case))
    with pytest.raises(ph.PhnswError):
        ShardedBuilder(eng, Bad()).generate(np.arange(300, dtype=np.uint64))

    class NoTransport(Bad):
        def c_comm(self, device=0):
            return Comm(rank=0, world=2, emulate=0)

    with pytest.raises(ph.PhnswError):
        ShardedBuilder(eng, NoTransport()).generate(np.arange(300, dtype=np.uint64))
