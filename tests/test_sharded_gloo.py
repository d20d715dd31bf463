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


def test_packed_gather_splits_back_exactly():
    """_gather_many: int32 ids, float32 distances and int32 lengths travel as one int32 block"""
    import torch
    from parallel_hnsw_amd.sharded import ShardedBuilder

    class TwoRankComm:  # what rank 1 contributes = rank 0's block with every word's bits flipped
        rank, world = 0, 2

        def all_gather(self, t):
            assert t.dtype == torch.int32 and t.dim() == 2
            return torch.cat([t, ~t], 0)

    class FakeEngine:
        bp = None

    b = ShardedBuilder(FakeEngine(), TwoRankComm(), shard_min=0)
    chunk, M = 5, 3
    ids = torch.arange(chunk * M, dtype=torch.int32).reshape(chunk, M)
    d = torch.linspace(-1, 1, chunk * M, dtype=torch.float32).reshape(chunk, M)
    ln = torch.arange(chunk, dtype=torch.int32) + 7
    n = 8  # the second rank's last two rows fall off
    gi, gd, gl = b._gather_many([ids, d, ln], n)
    assert gi.shape == (n, M) and gd.shape == (n, M) and gl.shape == (n,)
    assert gi.dtype == torch.int32 and gd.dtype == torch.float32 and gl.dtype == torch.int32
    assert torch.equal(gi[:chunk], ids) and torch.equal(gd[:chunk], d) and torch.equal(gl[:chunk], ln)
    assert torch.equal(gi[chunk:], (~ids)[: n - chunk])
    assert torch.equal(gd[chunk:].view(torch.int32), (~d.view(torch.int32))[: n - chunk])
    assert torch.equal(gl[chunk:], (~ln)[: n - chunk])
