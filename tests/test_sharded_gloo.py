"""The N>1 build path on CPU: parallel_hnsw_amd.sharded.ShardedBuilder under a world_size-2
`gloo` group with the oracle as the engine.  Checks the range split, padding, all-gather
assembly and the replicated control flow: both ranks must end with exactly the graph a
single process builds."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleEngine:
    """engine interface of sharded.py backed by the CPU oracle (u64 ids as int64 tensors)"""

    def __init__(self, rows, dim, bp, metric=0, threads=2):
        import torch
        import oracle
        self.torch, self.oracle = torch, oracle
        self.ix = oracle.Index(rows, dim=dim, metric=metric, sum_mode=oracle.SUM_BLOCKED64)
        self.ix.set_sum_mode(oracle.SUM_BLOCKED64)
        self.bp, self.threads, self.device = bp, threads, "cpu"
        self.L = oracle.lib()

    def empty(self, shape, kind):
        return self.torch.empty(shape, dtype=self.torch.float32 if kind == "f32" else self.torch.int64)

    @staticmethod
    def _p(t):
        assert t.is_contiguous()
        return C.c_void_p(t.data_ptr())

    def plan(self, vids):
        vs = self.oracle.shuffle(np.asarray(vids, dtype=np.uint64), self.bp.seed)
        return vs, self.oracle.calculate_partitions(len(vs), self.bp.order)

    def layer_begin(self, vids, W):
        v = np.ascontiguousarray(vids, dtype=np.uint64)
        assert self.L.orc_layer_begin(self.ix.h, v.ctypes.data_as(C.c_void_p), len(v), W, C.byref(self.bp)) == 0
        return True, int(self.L.orc_layer_init_stride(self.ix.h))

    def layer_init_search(self, first, count, ids, d, ln):
        assert self.L.orc_layer_init_search(self.ix.h, C.byref(self.bp), first, count, self._p(ids), self._p(d),
                                            self._p(ln), self.threads) == 0

    def layer_seed(self, ids, d, ln, first, count, rows, rows_d):
        assert self.L.orc_layer_seed(self.ix.h, C.byref(self.bp), self._p(ids), self._p(d), self._p(ln), first, count,
                                     self._p(rows), self._p(rows_d), self.threads) == 0

    def layer_finish(self, rows, rows_d):
        assert self.L.orc_layer_finish(self.ix.h, self._p(rows), self._p(rows_d), self.threads) == 0

    def layer_count(self):
        return self.ix.layer_count

    def layer_nodes(self, lft):
        return int(self.L.orc_index_layer(self.ix.h, lft).contents.node_count)

    def link_search(self, lft, sp, M, first, count, ids, d, ln):
        assert self.L.orc_link_search(self.ix.h, lft, sp, M, first, count, self._p(ids), self._p(d), self._p(ln),
                                      self.threads) == 0

    def link_apply(self, lft, M, ids, d, ln):
        return int(self.L.orc_link_apply(self.ix.h, lft, M, self._p(ids), self._p(d), self._p(ln), self.threads))

    def promote_at_layer(self, lft):
        return self.ix.promote_at_layer(lft, self.bp, threads=self.threads) > 0

    def discover_hits(self, lft, sp, first, count, hit):
        assert self.L.orc_discover_hits(self.ix.h, lft, sp, first, count, self._p(hit), self.threads) == 0

    def promote_from_hits(self, lft, hit):
        return self.L.orc_promote_at_layer_hits(self.ix.h, lft, C.byref(self.bp), self._p(hit), self.threads) > 0

    def recall_hits(self, at, op, first, count):
        hits, sel = C.c_uint64(), C.c_uint64()
        assert self.L.orc_recall_hits(self.ix.h, at, C.byref(op), first, count, C.byref(hits), C.byref(sel),
                                      self.threads) == 0
        return hits.value, sel.value


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
        eng = OracleEngine(rows, case["dim"], bp)
        comm = TorchComm()
        b = ShardedBuilder(eng, comm, shard_min=case.get("shard_min", 0))
        b.generate(np.arange(case["n"], dtype=np.uint64))
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


def test_range_split_covers_everything():
    from parallel_hnsw_amd.sharded import ShardedBuilder

    class FakeComm:
        def __init__(self, r, w):
            self.rank, self.world = r, w

    class FakeEngine:
        bp = None

    for n in (1, 2, 7, 8, 9, 1000, 1001):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                b = ShardedBuilder(FakeEngine(), FakeComm(r, w), shard_min=0)
                chunk, first, count = b._range(n)
                assert count <= chunk and first + count <= n
                assert first == min(n, r * chunk)
                seen += list(range(first, first + count))
            assert seen == list(range(n))


def test_phase_packs_pipelines_and_reassembles_exactly():
    """ShardedBuilder._phase: each piece of a rank's range travels as one byte block (ids, distances and lengths
    side by side), pieces are gathered asynchronously and land at their global rows -- with and without the
    sub-chunk pipeline, for 4- and 8-byte ids"""
    import torch
    from parallel_hnsw_amd.sharded import ShardedBuilder

    for id_dtype in (torch.int32, torch.int64):
        for world, n, subs in ((2, 37, 1), (3, 100, 4), (2, 64, 4), (4, 1001, 4)):
            def item(i, M):  # what work item i produces
                return (torch.arange(M, dtype=id_dtype) + 1000 * i, torch.full((M,), float(i)) + torch.arange(M) / 16.0, i + 7)

            class Comm:
                """rank `me` of `world`: the other ranks' blocks are computed on the spot"""
                def __init__(self, me, builder_of):
                    self.rank, self.world, self.builder_of, self.asyncs = me, world, builder_of, 0

                def all_gather(self, t):
                    return torch.cat([self.builder_of(r).block for r in range(world)], 0)

                def all_gather_async(self, t):
                    self.asyncs += 1
                    blocks = [self.builder_of(r).block for r in range(world)]

                    class H:
                        def wait(self_inner):
                            return torch.cat(blocks, 0)
                    return H()

            class Engine:
                bp = None

                def empty(self, shape, kind):
                    return torch.zeros(shape, dtype=torch.float32 if kind == "f32" else id_dtype)

            M = 3
            # every rank's packed piece depends only on (rank, piece): precompute them by running the phase's packing
            # for each rank in turn, piece by piece, with a recording comm
            results = {}
            for me in range(world):
                pieces = []

                class Rec:
                    rank, world_ = me, world

                    def __init__(self):
                        self.rank, self.world = me, world

                    def all_gather(self, t):
                        pieces.append(t.clone())
                        return torch.cat([t] * world, 0)

                    def all_gather_async(self, t):
                        pieces.append(t.clone())

                        class H:
                            def wait(self_inner):
                                return torch.cat([t] * world, 0)
                        return H()

                b = ShardedBuilder(Engine(), Rec(), shard_min=0)
                b.SUBCHUNKS, b.SUB_MIN = subs, 1

                def run(first, count, outs):
                    for r in range(count):
                        i, dd, l = item(first + r, M)
                        outs[0][r], outs[1][r], outs[2][r] = i, dd, l
                b._phase(n, [(M, "id"), (M, "f32"), (None, "id")], run)
                results[me] = pieces
            # now the real thing on rank 0 with a comm that hands out the recorded pieces in order
            turn = [0]

            class Replay:
                def __init__(self):
                    self.rank, self.world = 0, world

                def _next(self):
                    k = turn[0]
                    turn[0] += 1
                    return torch.cat([results[r][k] for r in range(world)], 0)

                def all_gather(self, t):
                    return self._next()

                def all_gather_async(self, t):
                    g = self._next()

                    class H:
                        def wait(self_inner):
                            return g
                    return H()

            b = ShardedBuilder(Engine(), Replay(), shard_min=0)
            b.SUBCHUNKS, b.SUB_MIN = subs, 1
            gi, gd, gl = b._phase(n, [(M, "id"), (M, "f32"), (None, "id")], run)
            assert gi.shape == (n, M) and gd.shape == (n, M) and gl.shape == (n,)
            assert gi.dtype == id_dtype and gd.dtype == torch.float32 and gl.dtype == id_dtype
            for i in range(n):
                ei, ed, el = item(i, M)
                assert torch.equal(gi[i], ei) and torch.equal(gd[i], ed) and int(gl[i]) == el, (world, n, subs, i)
