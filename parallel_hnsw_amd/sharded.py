"""Index construction sharded over the GPUs of one node (BASELINE
config 4): one process per GPU, `torch.distributed` over RCCL/xGMI.

Every rank holds a replica of the vector store and of the graph.  Each build round is
"every node of a layer runs a search against a snapshot, then proposes edges"
(reference src/lib.rs:1097-1153; likewise the seeding steps of generate_layer
lib.rs:700-787), so nodes are independent within a round:

    rank r searches the node range [r*chunk, (r+1)*chunk)        (K2 / K3 kernels)
    all-gather of the per-node results  (ids u32 + distances f32, n x M x 8 bytes)
    every rank applies ALL results to its replica                 (K5, deterministic)

so the replicas stay bit-identical and the only data-path collective is one all-gather per
phase (plus an all-reduce of two integers for the recall estimate).  The control flow is the
reference's (generate lib.rs:825-893, improve_index lib.rs:1546-1686, promotion excluded),
the same as libphnsw's single-GPU `phnsw_build`.

The driver is written against a small engine interface so that the CPU tests can run it
under `gloo` with the oracle as the engine; `GpuEngine` is the product engine.
"""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .hnsw import BuildParameters, Hnsw


class _Done:
    def __init__(self, value):
        self.value = value

    def wait(self):
        return self.value


class _Pending:
    def __init__(self, work, out, src, comm):
        self.work, self.out, self.src, self.comm = work, out, src, comm

    def wait(self):
        import time
        t0 = time.perf_counter()
        self.work.wait()  # stream-level: the current stream (libphnsw's null stream) waits for the transfer
        self.comm.seconds += time.perf_counter() - t0
        return self.out


class TorchComm:
    """all-gather / all-reduce over a torch.distributed group (nccl = RCCL on ROCm, gloo on CPU)"""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bytes_gathered = 0
        self.calls = 0
        self.seconds = 0.0  # host wall time inside the collectives (enqueue + any wait that blocks the host)

    def all_gather(self, t):
        if self.world == 1:
            return t
        import time
        import torch
        t0 = time.perf_counter()
        staged = t.is_cuda and self.dist.get_backend(self.group) == "gloo"  # 1-GPU rehearsal only
        src = t.cpu() if staged else t.contiguous()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=src.device)
        self.dist.all_gather_into_tensor(out, src, group=self.group)
        self.bytes_gathered += out.numel() * out.element_size()
        self.calls += 1
        if t.is_cuda and not staged:
            torch.cuda.synchronize(t.device)  # libphnsw reads the result outside torch's stream bookkeeping
        out = out.to(t.device) if staged else out
        self.seconds += time.perf_counter() - t0
        return out

    def all_gather_async(self, t):
        """start the collective and return a handle; handle.wait() orders the CURRENT stream behind it and returns
        the gathered tensor.  Over RCCL the transfer runs on the communicator's own stream, so kernels enqueued
        after this call (the next sub-chunk's searches) overlap it; gloo and the 1-GPU rehearsal complete here."""
        import torch
        if self.world == 1 or not t.is_cuda or self.dist.get_backend(self.group) == "gloo":
            return _Done(self.all_gather(t))
        import time
        t0 = time.perf_counter()
        src = t.contiguous()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=src.device)
        work = self.dist.all_gather_into_tensor(out, src, group=self.group, async_op=True)
        self.bytes_gathered += out.numel() * out.element_size()
        self.calls += 1
        self.seconds += time.perf_counter() - t0
        return _Pending(work, out, src, self)

    def all_reduce_sum(self, values, device):
        if self.world == 1:
            return list(values)
        import torch
        if self.dist.get_backend(self.group) == "gloo":
            device = "cpu"
        t = torch.tensor(list(values), dtype=torch.int64, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return [int(x) for x in t.tolist()]


class GpuEngine:
    """libphnsw's phase API over torch device tensors (u32 ids viewed as int32)"""

    def __init__(self, store, bp=None, device=None):
        import torch
        self.torch = torch
        self.store = store
        self.bp = bp or BuildParameters()
        self.device = device if device is not None else torch.device("cuda", store.device)
        h = C.c_void_p()
        check(lib().phnsw_index_create(store._h, C.byref(self.bp), C.byref(h)))
        self.hnsw = Hnsw(store, h, self.bp)

    # -- buffers
    def empty(self, shape, kind):
        dt = self.torch.float32 if kind == "f32" else self.torch.int32
        return self.torch.empty(shape, dtype=dt, device=self.device)

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr())

    def sync(self):
        self.torch.cuda.synchronize(self.device)

    # -- plan
    def plan(self, vids):
        vids = np.ascontiguousarray(vids, dtype=np.uint64)
        sh = np.empty_like(vids)
        sizes = np.zeros(64, dtype=np.uint64)
        cnt = C.c_uint32()
        check(lib().phnsw_build_plan(vids.ctypes.data_as(C.c_void_p), len(vids), C.byref(self.bp),
                                     sh.ctypes.data_as(C.c_void_p), sizes.ctypes.data_as(C.c_void_p), 64,
                                     C.byref(cnt)))
        return sh, [int(x) for x in sizes[:cnt.value]]

    # -- generate_layer phases
    def layer_begin(self, vids, W):
        vids = np.ascontiguousarray(vids, dtype=np.uint64)
        needs = C.c_int()
        check(lib().phnsw_layer_begin(self.hnsw._h, vids.ctypes.data_as(C.c_void_p), len(vids), W, C.byref(self.bp),
                                      C.byref(needs)))
        return bool(needs.value), int(self.bp.initial_partition_search.number_of_candidates)

    def layer_init_search(self, first, count, ids, d, ln):
        check(lib().phnsw_layer_init_search_device(self.hnsw._h, C.byref(self.bp), first, count, self._ptr(ids),
                                                   self._ptr(d), self._ptr(ln)))

    def layer_seed(self, ids, d, ln, first, count, rows, rows_d):
        check(lib().phnsw_layer_seed_device(self.hnsw._h, C.byref(self.bp), self._ptr(ids), self._ptr(d),
                                            self._ptr(ln), first, count, self._ptr(rows), self._ptr(rows_d)))

    def layer_finish(self, rows, rows_d):
        check(lib().phnsw_layer_finish_device(self.hnsw._h, self._ptr(rows), self._ptr(rows_d)))

    # -- link / recall phases
    def layer_count(self):
        return self.hnsw.layer_count()

    def layer_nodes(self, lft):
        n = C.c_uint64()
        check(lib().phnsw_index_layer_info(self.hnsw._h, lft, C.byref(n), None))
        return n.value

    def link_search(self, lft, sp, M, first, count, ids, d, ln):
        check(lib().phnsw_link_search_device(self.hnsw._h, lft, C.byref(sp), M, first, count, self._ptr(ids),
                                             self._ptr(d), self._ptr(ln)))

    def link_apply(self, lft, M, ids, d, ln):
        added = C.c_uint64()
        check(lib().phnsw_link_apply_device(self.hnsw._h, lft, M, self._ptr(ids), self._ptr(d), self._ptr(ln),
                                            C.byref(added)))
        return added.value

    def recall_hits(self, at, op, first, count):
        hits, sel = C.c_uint64(), C.c_uint64()
        check(lib().phnsw_recall_hits(self.hnsw._h, at, C.byref(op), first, count, C.byref(hits), C.byref(sel)))
        return hits.value, sel.value

    def promote_at_layer(self, lft):
        return self.hnsw.promote_at_layer(lft, self.bp)

    def discover_hits(self, lft, sp, first, count, hit):
        check(lib().phnsw_discover_hits_device(self.hnsw._h, lft, C.byref(sp), first, count, self._ptr(hit)))

    def promote_from_hits(self, lft, hit):
        out = C.c_int()
        check(lib().phnsw_promote_at_layer_hits_device(self.hnsw._h, lft, C.byref(self.bp), self._ptr(hit),
                                                       C.byref(out)))
        return bool(out.value)


class ShardedBuilder:
    """Hnsw::generate with every per-node phase split over the ranks of `comm`"""

    def __init__(self, engine, comm=None, shard_min=None):
        self.e = engine
        self.comm = comm or TorchComm()
        if shard_min is not None:
            self.SHARD_MIN = shard_min
        self.rank, self.world = self.comm.rank, self.comm.world
        self.bp = engine.bp

    # Work lists shorter than this are not split: a launch over a few thousand queries takes one
    # query-latency however few of them a rank keeps, so every rank runs the whole (identical)
    # list and the phase needs no collective at all.
    SHARD_MIN = 4096

    def _range(self, n):
        if n < self.SHARD_MIN:
            return n, 0, n
        chunk = -(-n // self.world)
        first = min(n, self.rank * chunk)
        count = min(n, first + chunk) - first
        return chunk, first, count

    # A rank's share of a phase is cut into SUBCHUNKS pieces when it is long enough: the all-gather of piece k
    # is started asynchronously and travels over xGMI while piece k + 1 is being searched.
    SUBCHUNKS = 4
    SUB_MIN = 8192

    def _phase(self, n, specs, run):
        """One sharded phase over a list of n work items.  specs: [(width or None, kind)] of the per-item
        outputs; run(first, count, outs) fills rows [0, count) of the given buffers with the results of items
        [first, first + count).  Returns the full [n, ...] arrays, identical on every rank."""
        chunk, first, count = self._range(n)
        outs = [self.e.empty((chunk,) if w is None else (chunk, w), kind) for w, kind in specs]
        if n < self.SHARD_MIN:
            run(first, count, outs)
            return [o[:n] for o in outs]
        import torch
        nsub = self.SUBCHUNKS if (hasattr(self.comm, "all_gather_async") and chunk >= self.SUBCHUNKS * self.SUB_MIN) else 1
        bounds = [chunk * k // nsub for k in range(nsub + 1)]
        pending = []
        for k in range(nsub):
            lo, hi = bounds[k], bounds[k + 1]
            cnt = max(0, min(count, hi) - lo)
            if cnt:
                run(first + lo, cnt, [o[lo:lo + cnt] for o in outs])
            # the piece's outputs side by side as raw bytes: one collective per piece
            cols = [o[lo:hi].reshape(hi - lo, -1).contiguous().view(torch.uint8) for o in outs]
            packed = cols[0] if len(cols) == 1 else torch.cat(cols, dim=1)
            h = self.comm.all_gather_async(packed) if nsub > 1 else _Done(self.comm.all_gather(packed))
            pending.append((lo, hi, h, [c.shape[1] for c in cols]))
        full = [self.e.empty((self.world * chunk,) + tuple(o.shape[1:]), kind) for o, (_, kind) in zip(outs, specs)]
        for lo, hi, h, widths in pending:
            g = h.wait().view(self.world, hi - lo, -1)
            at = 0
            for f, o, wb in zip(full, outs, widths):
                piece = g[:, :, at:at + wb].contiguous().view(o.dtype)
                f.view((self.world, chunk) + tuple(o.shape[1:]))[:, lo:hi] = piece.reshape((self.world, hi - lo) + tuple(o.shape[1:]))
                at += wb
        return [f[:n] for f in full]

    # generate_layer  lib.rs:675-823
    def generate_layer(self, vids, W):
        needs, K = self.e.layer_begin(vids, W)
        if not needs:
            return
        n = len(vids)
        ids_f, d_f, ln_f = self._phase(n, [(K, "id"), (K, "f32"), (None, "id")],
                                       lambda f, c, o: self.e.layer_init_search(f, c, o[0], o[1], o[2]))
        ids_f, d_f, ln_f = ids_f.contiguous(), d_f.contiguous(), ln_f.contiguous()
        rows_f, rows_d_f = self._phase(n, [(W, "id"), (W, "f32")],
                                       lambda f, c, o: self.e.layer_seed(ids_f, d_f, ln_f, f, c, o[0], o[1]))
        self.e.layer_finish(rows_f.contiguous(), rows_d_f.contiguous())

    # link_layer_to_better_neighbors  lib.rs:1070-1154
    def link_layer(self, lft, sp, M):
        n = self.e.layer_nodes(lft)
        ids_f, d_f, ln_f = self._phase(n, [(M, "id"), (M, "f32"), (None, "id")],
                                       lambda f, c, o: self.e.link_search(lft, sp, M, f, c, o[0], o[1], o[2]))
        return self.e.link_apply(lft, M, ids_f.contiguous(), d_f.contiguous(), ln_f.contiguous())

    # stochastic_recall_at  lib.rs:1463-1499
    def stochastic_recall_at(self, at):
        op = self.bp.optimization
        total = self.e.layer_nodes(at)
        selection = min(total, max(1, int(np.float32(total) * np.float32(op.recall_proportion))))
        chunk, first, count = self._range(selection)
        hits, sel = self.e.recall_hits(at, op, first, count)
        assert sel == selection, (sel, selection)
        if selection >= self.SHARD_MIN:
            (hits,) = self.comm.all_reduce_sum([hits], getattr(self.e, "device", "cpu"))
        return float(np.float32(hits) / np.float32(selection))

    # improve_neighbors_upto  lib.rs:1515-1544
    def improve_neighbors_upto(self, upto, last_recall=None):
        op = self.bp.optimization
        last = np.float32(0.0 if last_recall is None else last_recall)
        improvement = np.float32(1.0)
        rounds = 0
        while improvement >= np.float32(op.neighborhood_threshold) and last < np.float32(1.0):
            for lft in range(upto):
                self.link_layer(lft, op.search, self.bp.neighborhood_size)
            recall = np.float32(self.stochastic_recall_at(upto - 1))
            improvement = recall - last
            last = recall
            rounds += 1
            if self.bp.max_link_rounds and rounds >= self.bp.max_link_rounds:
                break
        return float(last)

    # promote_at_layer  lib.rs:1273-1427: its n searches (discover_unreachable_vectors) are
    # sharded like a link round, the (integer, sequential) promotion itself runs replicated
    def promote_at_layer(self, lft):
        if not hasattr(self.e, "discover_hits"):
            return self.e.promote_at_layer(lft)
        n = self.e.layer_nodes(lft)
        (hit_f,) = self._phase(n, [(None, "id")],
                               lambda f, c, o: self.e.discover_hits(lft, self.bp.optimization.search, f, c, o[0]))
        return self.e.promote_from_hits(lft, hit_f.contiguous())

    # improve_index_at  lib.rs:1546-1603
    def improve_index_at(self, lft):
        op = self.bp.optimization
        recall = np.float32(self.stochastic_recall_at(lft))
        improvement, bailout = np.float32(1.0), 1
        while improvement >= np.float32(op.promotion_threshold) and recall < np.float32(1.0) and bailout != 0:
            last, cur = recall, 0
            while cur <= lft and bailout != 0:
                layer_count = self.e.layer_count()
                recall = np.float32(self.improve_neighbors_upto(cur + 1))
                if recall == np.float32(1.0):
                    cur += 1
                    continue
                if self.bp.promote and self.promote_at_layer(cur):
                    delta = self.e.layer_count() - layer_count
                    cur += delta
                    lft += delta
                    recall = np.float32(self.improve_neighbors_upto(cur + 1, float(recall)))
                cur += 1
            bailout -= 1
            improvement = recall - last
        return float(recall), lft

    # improve_index  lib.rs:1664-1686
    def improve_index(self):
        recall = self.stochastic_recall_at(self.e.layer_count() - 1)
        lft = 0
        while lft < self.e.layer_count():
            recall, lft = self.improve_index_at(lft)
            lft += 1
        return recall

    # Hnsw::generate  lib.rs:825-893
    def generate(self, vids):
        vs, sizes = self.e.plan(vids)
        n = len(vs)
        i = 0
        while i != len(sizes):
            length = min(sizes[i], n)
            level = len(sizes) - i - 1
            W = self.bp.zero_layer_neighborhood_size if level == 0 else self.bp.neighborhood_size
            self.generate_layer(vs[:length], W)
            old = self.e.layer_count()
            self.improve_index()
            delta = self.e.layer_count() - old
            if delta > 0:  # promotion added layers: fix the partitions  lib.rs:880-887
                sizes = [self.e.layer_nodes(l) for l in range(self.e.layer_count())] + sizes[i + 1:]
                i += delta
            i += 1
        return getattr(self.e, "hnsw", None)
