"""Index construction sharded over the GPUs of one node (BASELINE config 4) -- the Python
face of libphnsw's sharded driver (csrc/sharded.hip, include/phnsw.h "sharded build").

The driver itself (range split, block layout, sub-chunk pipeline, reassembly, the control flow of
Hnsw::generate / improve_index, reference src/lib.rs:825-893, 1515-1686) is C++ inside libphnsw.so, reachable from
any host language through `phnsw_build_sharded`.  This module only supplies what a Python host has to:

  * `TorchComm`     -- a `phnsw_comm` over a torch.distributed group: under `nccl` the library's own RCCL transport
                       (`phnsw_comm_rccl_create`; the 128-byte id travels through the group), under `gloo` two host
                       callbacks (the CPU tests, and two ranks sharing the one GPU of a test box);
  * `EmulatedComm`  -- one process plays all ranks in turn (one-GPU tests, bench.py's scaling model);
  * `PythonEngine`  -- a `phnsw_shard_engine` whose phases are Python methods: how tests/test_sharded_gloo.py runs
                       the same C++ driver over the oracle's phases on CPU;
  * `ShardedBuilder`/`build_sharded` -- the call itself.
"""
import ctypes as C

import numpy as np

from ._lib import (AllGatherFn, AllReduceFn, BuildParams, Comm, OptimizationParams, SearchParams, ShardEngine,
                   ShardedStats, SHARD_PHASE_NAMES, check, lib)
from .hnsw import BuildParameters, Hnsw


def _stats_dict(st):
    d = {k: getattr(st, k) for k, _ in ShardedStats._fields_ if k != "seconds_by_phase"}
    d["seconds_by_phase"] = {name: st.seconds_by_phase[i] for i, name in enumerate(SHARD_PHASE_NAMES)}
    return d


class _CommBase:
    """keeps the ctypes struct and its callbacks alive; `.c` is what the ABI takes"""
    rank, world = 0, 1
    bytes_gathered = 0
    calls = 0
    seconds = 0.0

    def c_comm(self, device=0):
        raise NotImplementedError

    def close(self):
        pass


class EmulatedComm(_CommBase):
    """rank `rank` of an emulated world: the calling process computes every rank's share itself, in rank order,
    through the driver's real split / block layout / reassembly (phnsw_comm.emulate)"""

    def __init__(self, world, rank=0):
        self.rank, self.world = rank, world
        self._c = Comm(rank=rank, world=world, host_buffers=0, emulate=1, ctx=None,
                       all_gather=C.cast(None, AllGatherFn), all_reduce_sum=C.cast(None, AllReduceFn))

    def c_comm(self, device=0):
        return self._c


class TorchComm(_CommBase):
    """a torch.distributed group as a phnsw_comm (nccl = RCCL on ROCm, gloo on CPU)"""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._c = None
        self._native = None
        self._keep = []

    def _host_all_gather(self, ctx, send, recv, nbytes, stream):
        try:
            import time
            import torch
            t0 = time.perf_counter()
            src = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(send)))
            out = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * (nbytes * self.world)).from_address(recv)))
            self.dist.all_gather_into_tensor(out, src, group=self.group)
            self.bytes_gathered += nbytes * self.world
            self.calls += 1
            self.seconds += time.perf_counter() - t0
            return 0
        except Exception as exc:  # nothing may unwind through the C frames
            print("TorchComm.all_gather failed: %r" % (exc,), flush=True)
            return 1

    def _host_all_reduce(self, ctx, values, count):
        try:
            import torch
            t = torch.from_numpy(np.ctypeslib.as_array(values, shape=(count,)).view(np.int64))
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            return 0
        except Exception as exc:
            print("TorchComm.all_reduce_sum failed: %r" % (exc,), flush=True)
            return 1

    def c_comm(self, device=0):
        if self._c is not None:
            return self._c
        if self.world == 1:
            self._c = Comm(rank=0, world=1)
            return self._c
        if self.dist.get_backend(self.group) == "nccl":
            # the library's own transport: ncclAllGather on its stream, nothing of torch on the data path
            ident = (C.c_uint8 * 128)()
            if self.rank == 0:
                check(lib().phnsw_comm_rccl_unique_id(ident))
            box = [bytes(ident)]
            self.dist.broadcast_object_list(box, src=self.dist.get_global_rank(self.group, 0) if self.group else 0,
                                            group=self.group)
            ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
            p = C.POINTER(Comm)()
            check(lib().phnsw_comm_rccl_create(ident, self.rank, self.world, device, C.byref(p)))
            self._native = p
            self._c = p.contents
            return self._c
        ag, ar = AllGatherFn(self._host_all_gather), AllReduceFn(self._host_all_reduce)
        self._keep = [ag, ar]
        self._c = Comm(rank=self.rank, world=self.world, host_buffers=1, emulate=0, ctx=None, all_gather=ag,
                       all_reduce_sum=ar)
        return self._c

    def close(self):
        if self._native is not None:
            lib().phnsw_comm_destroy(self._native)
            self._native = None
        self._c = None


class GpuEngine:
    """libphnsw's own phases: the engine of `phnsw_build_sharded` (nothing to supply -- store and parameters)"""

    def __init__(self, store, bp=None, device=None):
        self.store = store
        self.bp = bp or BuildParameters()
        self.device = store.device if device is None else getattr(device, "index", device)
        self.hnsw = None


class PythonEngine:
    """A phnsw_shard_engine whose phases are Python methods working on raw host pointers (ints).  Buffers are
    numpy arrays owned here.  Subclasses implement: plan, layer_begin -> (needs, K), layer_init_search, layer_seed,
    layer_finish, layer_count, layer_nodes, link_search, link_apply -> added, recall_hits -> (hits, selection),
    discover_hits, promote_from_hits -> bool."""
    id_bytes = 8

    def __init__(self):
        self._bufs = {}
        self._cbs = []
        self._c = None

    # -- buffers
    def _alloc(self, ctx, nbytes):
        a = np.empty(max(int(nbytes), 16) + 256, dtype=np.uint8)
        p = (a.ctypes.data + 255) & ~255
        self._bufs[p] = a
        return p

    def _release(self, ctx, p):
        self._bufs.pop(p, None)

    @staticmethod
    def _copy2d(ctx, dst, dpitch, src, spitch, width, height):
        for r in range(height):
            C.memmove(dst + r * dpitch, src + r * spitch, width)
        return 0

    def c_engine(self):
        if self._c is not None:
            return self._c
        F = dict(ShardEngine._fields_)

        def guard(fn, fail=1):
            def wrapped(*a):
                try:
                    r = fn(*a)
                    return 0 if r is None else r
                except Exception as exc:  # nothing may unwind through the C frames
                    import traceback
                    traceback.print_exc()
                    print("PythonEngine.%s failed: %r" % (getattr(fn, "__name__", "?"), exc), flush=True)
                    return fail
            return wrapped

        def plan(ctx, vids, n, shuffled, sizes, max_layers, count):
            vs, parts = self.plan(np.ctypeslib.as_array(C.cast(vids, C.POINTER(C.c_uint64)), shape=(n,)).copy())
            np.ctypeslib.as_array(C.cast(shuffled, C.POINTER(C.c_uint64)), shape=(n,))[:] = vs
            if len(parts) > max_layers:
                return 1
            np.ctypeslib.as_array(C.cast(sizes, C.POINTER(C.c_uint64)), shape=(len(parts),))[:] = parts
            count[0] = len(parts)
            return 0

        def layer_begin(ctx, vids, n, W, needs, K):
            nd, k = self.layer_begin(np.ctypeslib.as_array(C.cast(vids, C.POINTER(C.c_uint64)), shape=(n,)).copy(), W)
            needs[0], K[0] = int(bool(nd)), int(k)
            return 0

        def link_apply(ctx, lft, M, ids, d, ln, added):
            added[0] = int(self.link_apply(lft, M, ids, d, ln))
            return 0

        def recall_hits(ctx, at, op, first, count, hits, sel):
            h, s = self.recall_hits(at, op.contents, first, count)
            hits[0], sel[0] = int(h), int(s)
            return 0

        def promote_from_hits(ctx, lft, hit, promoted):
            promoted[0] = int(bool(self.promote_from_hits(lft, hit)))
            return 0

        table = {
            "alloc": (self._alloc, 0), "release": (self._release, None), "copy2d": (self._copy2d, 1),
            "plan": (plan, 1), "layer_begin": (layer_begin, 1),
            "layer_init_search": (lambda ctx, f, c, ids, d, ln: self.layer_init_search(f, c, ids, d, ln), 1),
            "layer_seed": (lambda ctx, ii, idd, il, f, c, rows, rows_d: self.layer_seed(ii, idd, il, f, c, rows, rows_d), 1),
            "layer_finish": (lambda ctx, rows, rows_d: self.layer_finish(rows, rows_d), 1),
            "layer_count": (lambda ctx: int(self.layer_count()), 0),
            "layer_nodes": (lambda ctx, lft: int(self.layer_nodes(lft)), 0),
            "link_search": (lambda ctx, lft, sp, M, f, c, ids, d, ln: self.link_search(lft, sp.contents, M, f, c, ids, d, ln), 1),
            "link_apply": (link_apply, 1), "recall_hits": (recall_hits, 1),
            "discover_hits": (lambda ctx, lft, sp, f, c, hit: self.discover_hits(lft, sp.contents, f, c, hit), 1),
            "promote_from_hits": (promote_from_hits, 1),
        }
        kw = {}
        for name, (fn, fail) in table.items():
            cb = F[name](guard(fn, fail) if fail is not None else fn)
            self._cbs.append(cb)
            kw[name] = cb
        self._c = ShardEngine(ctx=None, id_bytes=self.id_bytes, host_buffers=1, **kw)
        return self._c


def sharded_tuning(shard_min=0, subchunks=0, sub_min=0):
    """process-wide knobs of the driver (0 keeps a value): lists shorter than shard_min run whole on every rank; a
    rank's share is cut into `subchunks` pieces of at least sub_min items"""
    check(lib().phnsw_sharded_tuning(shard_min, subchunks, sub_min))


def build_sharded(store, vids, bp=None, comm=None, progress=None):
    """Hnsw::generate over the ranks of `comm` (phnsw_build_sharded) -> (Hnsw, stats dict)"""
    from ._lib import PROGRESS_CB
    bp = bp or BuildParameters()
    comm = comm or TorchComm()
    vids = np.ascontiguousarray(vids, dtype=np.uint64)
    h = C.c_void_p()
    st = ShardedStats()
    cb = PROGRESS_CB(progress) if progress else C.cast(None, PROGRESS_CB)
    cc = comm.c_comm(store.device)
    check(lib().phnsw_build_sharded(store._h, vids.ctypes.data_as(C.c_void_p), len(vids), C.byref(bp), C.byref(cc), cb,
                                    None, C.byref(h), C.byref(st)))
    return Hnsw(store, h, bp), _stats_dict(st)


def improve_index_sharded(hnsw, bp=None, comm=None, last_recall=None):
    """Hnsw::improve_index over the ranks of `comm` -> (recall, stats dict)"""
    bp = bp or hnsw.build_parameters
    comm = comm or TorchComm()
    st = ShardedStats()
    out = C.c_float()
    cc = comm.c_comm(hnsw.store.device)
    check(lib().phnsw_improve_index_sharded(hnsw._h, C.byref(bp), float("nan") if last_recall is None else last_recall,
                                            C.byref(cc), C.byref(out), C.byref(st)))
    return out.value, _stats_dict(st)


class ShardedBuilder:
    """`ShardedBuilder(engine, comm).generate(vids)`: the sharded Hnsw::generate.  With a `GpuEngine` this is
    `phnsw_build_sharded` (returns the Hnsw); with a `PythonEngine` the same driver runs over its phases
    (`phnsw_build_sharded_engine`; the engine holds the result)."""

    def __init__(self, engine, comm=None, shard_min=None, subchunks=None, sub_min=None):
        self.e = engine
        self.comm = comm or TorchComm()
        self.bp = engine.bp
        self.stats = None
        if shard_min is not None or subchunks is not None or sub_min is not None:
            # 0 means "keep" at the ABI; a caller asking for "split everything" passes 1
            sharded_tuning(max(1, shard_min) if shard_min is not None else 0, subchunks or 0,
                           max(1, sub_min) if sub_min is not None else 0)

    def generate(self, vids):
        vids = np.ascontiguousarray(vids, dtype=np.uint64)
        if isinstance(self.e, GpuEngine):
            h, self.stats = build_sharded(self.e.store, vids, self.bp, self.comm)
            self.e.hnsw = h
            return h
        st = ShardedStats()
        eng = self.e.c_engine()
        cc = self.comm.c_comm(0)
        assert C.sizeof(self.bp) == C.sizeof(BuildParams)  # the engine may carry its own ctypes mirror of the struct
        bp = BuildParams.from_buffer_copy(bytes(self.bp))
        check(lib().phnsw_build_sharded_engine(C.byref(eng), vids.ctypes.data_as(C.c_void_p), len(vids),
                                               C.byref(bp), C.byref(cc), C.byref(st)))
        self.stats = _stats_dict(st)
        return None
