"""ctypes front end of the CPU ORACLE (test infrastructure, see oracle/orc.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It restates the reference crate's search/build algorithm in plain C
(oracle/orc_*.c, each function citing the reference file:line) and is pinned by the
reference's own golden vectors under tests/golden/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)
FMAX = np.float32(3.4028234663852886e38)
METRIC_COSINE_HALF, METRIC_ONE_MINUS_DOT, METRIC_L2 = 0, 1, 2
SUM_SEQ, SUM_BLOCKED64, SUM_SEQFMA = 0, 1, 2


def build_lib(force=False):
    """compile oracle/liboracle.so with gcc (oracle/Makefile)"""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class SearchParams(C.Structure):
    _fields_ = [("number_of_candidates", C.c_uint64),
                ("upper_layer_candidate_count", C.c_uint64),
                ("probe_depth", C.c_uint64)]


class OptParams(C.Structure):
    _fields_ = [("promotion_threshold", C.c_float), ("neighborhood_threshold", C.c_float),
                ("recall_proportion", C.c_float), ("promotion_proportion", C.c_float),
                ("search", SearchParams)]


class BuildParams(C.Structure):
    _fields_ = [("order", C.c_uint64), ("zero_layer_neighborhood_size", C.c_uint64),
                ("neighborhood_size", C.c_uint64), ("optimization", OptParams),
                ("initial_partition_search", SearchParams), ("seed", C.c_uint64),
                ("max_link_rounds", C.c_uint64), ("promote", C.c_uint64)]


class Store(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("n", C.c_uint64), ("dim", C.c_uint32), ("ld", C.c_uint32),
                ("metric", C.c_int), ("sum_mode", C.c_int), ("codes", C.c_void_p), ("codebook", C.c_void_p),
                ("pq_m", C.c_uint32), ("pq_ksub", C.c_uint32), ("pq_dsub", C.c_uint32),
                ("pq_table_f16", C.c_uint32)]


class LayerS(C.Structure):
    _fields_ = [("node_count", C.c_uint64), ("neighborhood_size", C.c_uint64),
                ("nodes", C.POINTER(C.c_uint64)), ("neighbors", C.POINTER(C.c_uint64))]


class Pq(C.Structure):
    _fields_ = [("data", C.c_void_p), ("prio", C.c_void_p), ("cap", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("n_dist", C.c_uint64), ("n_hops", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_lib()
        L = C.CDLL(_LIB_PATH)
        vp, u64, u32, i32, f32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_float
        L.orc_distance.restype = f32
        L.orc_distance.argtypes = [C.POINTER(Store), vp, vp]
        L.orc_pq_len.restype = u64
        L.orc_pq_len.argtypes = [C.POINTER(Pq)]
        L.orc_pq_insert.restype = u64
        L.orc_pq_insert.argtypes = [C.POINTER(Pq), u64, f32]
        L.orc_pq_merge.restype = i32
        L.orc_pq_merge.argtypes = [C.POINTER(Pq), vp, vp, u64]
        L.orc_final_neighbor_idx.restype = u64
        L.orc_final_neighbor_idx.argtypes = [u64, vp, u64]
        L.orc_default_build_params.argtypes = [C.POINTER(BuildParams)]
        L.orc_index_new.restype = vp
        L.orc_index_new.argtypes = [vp, u64, u32, u32, i32, i32]
        L.orc_index_free.argtypes = [vp]
        L.orc_index_set_sum_mode.argtypes = [vp, i32]
        L.orc_index_push_layer.argtypes = [vp, vp, vp, u64, u64]
        L.orc_index_layer_count.restype = u32
        L.orc_index_layer_count.argtypes = [vp]
        L.orc_index_layer.restype = C.POINTER(LayerS)
        L.orc_index_layer.argtypes = [vp, u32]
        L.orc_search_batch.restype = i32
        L.orc_search_batch.argtypes = [vp, vp, u32, vp, u64, SearchParams, vp, vp, vp, vp, vp, i32]
        L.orc_knn.restype = i32
        L.orc_knn.argtypes = [vp, u64, u64, vp, vp, vp, i32]
        L.orc_threshold_nn.restype = i32
        L.orc_threshold_nn.argtypes = [vp, f32, u64, u64, u64, vp, vp, vp, i32]
        L.orc_bruteforce.restype = i32
        L.orc_bruteforce.argtypes = [C.POINTER(Store), vp, u32, u64, u64, vp, vp, i32]
        L.orc_calculate_partitions.restype = u32
        L.orc_calculate_partitions.argtypes = [u64, u64, vp, u32]
        L.orc_calculate_partitions_for_additions.restype = u32
        L.orc_calculate_partitions_for_additions.argtypes = [vp, u32, u64, u64, vp, u32]
        L.orc_generate.restype = vp
        L.orc_generate.argtypes = [vp, u64, u32, u32, i32, i32, vp, u64, C.POINTER(BuildParams), i32]
        L.orc_generate_layer.restype = i32
        L.orc_generate_layer.argtypes = [vp, vp, u64, u64, C.POINTER(BuildParams), i32]
        L.orc_layer_begin.restype = i32
        L.orc_layer_begin.argtypes = [vp, vp, u64, u64, C.POINTER(BuildParams)]
        L.orc_layer_init_stride.restype = u64
        L.orc_layer_init_stride.argtypes = [vp]
        L.orc_layer_init_search.restype = i32
        L.orc_layer_init_search.argtypes = [vp, C.POINTER(BuildParams), u64, u64, vp, vp, vp, i32]
        L.orc_layer_seed.restype = i32
        L.orc_layer_seed.argtypes = [vp, C.POINTER(BuildParams), vp, vp, vp, u64, u64, vp, vp, i32]
        L.orc_layer_finish.restype = i32
        L.orc_layer_finish.argtypes = [vp, vp, vp, i32]
        L.orc_link_search.restype = i32
        L.orc_link_search.argtypes = [vp, u32, SearchParams, u64, u64, u64, vp, vp, vp, i32]
        L.orc_link_apply.restype = u64
        L.orc_link_apply.argtypes = [vp, u32, u64, vp, vp, vp, i32]
        L.orc_recall_hits.restype = i32
        L.orc_recall_hits.argtypes = [vp, u32, C.POINTER(OptParams), u64, u64, C.POINTER(u64), C.POINTER(u64), i32]
        L.orc_link_layer.restype = u64
        L.orc_link_layer.argtypes = [vp, u32, SearchParams, u64, i32]
        L.orc_stochastic_recall_at.restype = f32
        L.orc_stochastic_recall_at.argtypes = [vp, u32, C.POINTER(OptParams), i32]
        L.orc_improve_neighbors_upto.restype = f32
        L.orc_improve_neighbors_upto.argtypes = [vp, u32, C.POINTER(BuildParams), f32, i32]
        L.orc_improve_index.restype = f32
        L.orc_improve_index.argtypes = [vp, C.POINTER(BuildParams), i32]
        L.orc_improve_index_from.restype = f32
        L.orc_improve_index_from.argtypes = [vp, C.POINTER(BuildParams), f32, i32]
        L.orc_promote_at_layer.restype = i32
        L.orc_promote_at_layer.argtypes = [vp, u32, C.POINTER(BuildParams), i32]
        L.orc_discover_hits.restype = i32
        L.orc_discover_hits.argtypes = [vp, u32, SearchParams, u64, u64, vp, i32]
        L.orc_promote_at_layer_hits.restype = i32
        L.orc_promote_at_layer_hits.argtypes = [vp, u32, C.POINTER(BuildParams), vp, i32]
        L.orc_extend_layer.restype = i32
        L.orc_extend_layer.argtypes = [vp, u32, vp, u64]
        L.orc_discover_unreachable.restype = u64
        L.orc_discover_unreachable.argtypes = [vp, u32, SearchParams, C.POINTER(C.POINTER(u64)), i32]
        L.orc_check_layer_invariants.restype = i32
        L.orc_check_layer_invariants.argtypes = [vp]
        L.orc_mix64.restype = u64
        L.orc_mix64.argtypes = [u64]
        L.orc_shuffle_u64.argtypes = [vp, u64, u64]
        L.orc_synth_rows.argtypes = [vp, u64, u64, u32, u32, u64, i32, i32]
        L.orc_synth_clustered_rows.argtypes = [vp, u64, u64, u32, u32, u64, u32, f32, i32]
        L.orc_first_touch.argtypes = [vp, u64, i32]
        L.orc_first_touch.restype = None
        L.orc_pq_create.restype = i32
        L.orc_pq_create.argtypes = [vp, u64, u32, u32, u32, u32, u64, vp, vp, i32]
        L.orc_pq_create_kmeans.restype = i32
        L.orc_pq_create_kmeans.argtypes = [vp, u64, u32, u32, u32, u32, u64, u32, u64, vp, vp, i32]
        L.orc_index_set_pq.argtypes = [vp, vp, vp, u32, u32, u32]
        L.orc_index_set_pq_table_f16.argtypes = [vp, i32]
        L.orc_pq_search_batch.restype = i32
        L.orc_pq_search_batch.argtypes = [vp, C.POINTER(Store), vp, u32, u64, SearchParams, i32, vp, vp, vp, vp, i32]
        L.orc_feistel_perm.restype = u64
        L.orc_feistel_perm.argtypes = [u64, u64, u64]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def default_build_params(**kw):
    bp = BuildParams()
    lib().orc_default_build_params(C.byref(bp))
    for k, v in kw.items():
        setattr(bp, k, v)
    return bp


def pad_rows(x):
    """rows padded with zeros to a multiple of 4 floats (the store layout)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, dim = x.shape
    ld = (dim + 3) // 4 * 4
    if ld == dim:
        return x, dim, ld
    out = np.zeros((n, ld), dtype=np.float32)
    out[:, :dim] = x
    return out, dim, ld


class PriorityQueue:
    """src/priority_queue.rs PriorityQueue over caller arrays (from_slices)"""

    def __init__(self, data, priorities):
        self.data = np.array(data, dtype=np.uint64)
        self.priorities = np.array(priorities, dtype=np.float32)
        assert len(self.data) == len(self.priorities)
        self._pq = Pq(_p(self.data), _p(self.priorities), len(self.data))

    @classmethod
    def new(cls, size):
        return cls([EMPTY] * size, [FMAX] * size)

    def insert(self, elt, priority):
        return lib().orc_pq_insert(C.byref(self._pq), int(elt), float(np.float32(priority)))

    def merge(self, ids, prios):
        ids = np.array(ids, dtype=np.uint64)
        prios = np.array(prios, dtype=np.float32)
        return bool(lib().orc_pq_merge(C.byref(self._pq), _p(ids), _p(prios), len(ids)))

    def merge_pairs(self, pairs):
        return self.merge([p[0] for p in pairs], [p[1] for p in pairs])

    def __len__(self):
        return lib().orc_pq_len(C.byref(self._pq))

    def last(self):
        n = len(self)
        return None if n == 0 else (int(self.data[n - 1]), self.priorities[n - 1])


def final_neighbor_idx(neighborhood_size, neighbors, n):
    nb = np.array(neighbors, dtype=np.uint64)
    return lib().orc_final_neighbor_idx(neighborhood_size, _p(nb), n)


def calculate_partitions(total, order):
    out = np.zeros(128, dtype=np.uint64)
    n = lib().orc_calculate_partitions(total, order, _p(out), 128)
    return [int(x) for x in out[:n]]


def calculate_partitions_for_additions(sizes_from_bottom, new_vecs, order):
    s = np.array(sizes_from_bottom, dtype=np.uint64)
    out = np.zeros(128, dtype=np.uint64)
    n = lib().orc_calculate_partitions_for_additions(_p(s), len(s), new_vecs, order, _p(out), 128)
    return [int(x) for x in out[:n]]


def synth_rows(first, count, dim, seed=42, normalize=True, threads=8):
    ld = (dim + 3) // 4 * 4
    rows = np.zeros((count, ld), dtype=np.float32)
    lib().orc_synth_rows(_p(rows), first, count, dim, ld, seed, int(normalize), threads)
    return rows


def synth_clustered_rows(first, count, dim, seed=42, n_clusters=1000, noise=1.0, threads=8):
    ld = (dim + 3) // 4 * 4
    rows = np.zeros((count, ld), dtype=np.float32)
    lib().orc_synth_clustered_rows(_p(rows), first, count, dim, ld, seed, n_clusters, noise, threads)
    return rows


def empty_rows_first_touched(n, dim, threads=8):
    """[n, dim] f32 whose pages were first written by `threads` OpenMP threads (static schedule)"""
    rows = np.empty((n, dim), dtype=np.float32)
    lib().orc_first_touch(_p(rows), rows.size, threads)
    return rows


def tie_swap_report(a_ids, a_d, b_ids, b_d, k=10, rel=1e-5):
    """Two result lists of the same queries computed with different f32 summation orders (e.g. the GPU's
    blocked sum and the reference's sequential sum, bigvec.rs:48-51) may only differ by near-tie swaps
    (SURVEY 7.3): for every slot s < k whose ids differ, each of the two ids must also be in the OTHER
    list (anywhere in its full length), at a position whose distance -- in that list's own arithmetic --
    is within `rel` (relative) of that list's distance at slot s: the two ids sit in one tie group and
    only their order changed.  Returns counts; `unexplained` must be 0."""
    a_ids, b_ids = np.asarray(a_ids), np.asarray(b_ids)
    a_d, b_d = np.asarray(a_d, dtype=np.float64), np.asarray(b_d, dtype=np.float64)
    nq = a_ids.shape[0]
    differing = explained = 0
    worst = 0.0
    for q in np.nonzero((a_ids[:, :k] != b_ids[:, :k]).any(1))[0]:
        for s in np.nonzero(a_ids[q, :k] != b_ids[q, :k])[0]:
            differing += 1
            ok = True
            for x, ids_o, d_o in ((a_ids[q, s], b_ids[q], b_d[q]), (b_ids[q, s], a_ids[q], a_d[q])):
                pos = np.nonzero(ids_o == x)[0]
                if not len(pos):
                    ok = False
                    break
                gap = abs(d_o[pos[0]] - d_o[s]) / max(abs(d_o[s]), 1e-30)
                worst = max(worst, gap) if gap <= rel else worst
                if gap > rel:
                    ok = False
                    break
            explained += int(ok)
    return {"queries": int(nq), "k": int(k), "slots_equal": round(float((a_ids[:, :k] == b_ids[:, :k]).mean()), 6),
            "differing_slots": int(differing), "explained_as_near_tie_swaps": int(explained),
            "unexplained": int(differing - explained), "rel_tolerance": rel, "largest_tie_gap_rel": float(worst)}


def pq_create(rows, dim, m, ksub, seed=0, threads=8, kmeans_iters=0, sample=0):
    """codes [n, m] u8 and codebook [m, ksub, dim/m] (orc_pq_create_kmeans; kmeans_iters=0: random centroids)"""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    n, ld = rows.shape
    codes = np.zeros((n, m), dtype=np.uint8)
    codebook = np.zeros((m, ksub, dim // m), dtype=np.float32)
    rc = lib().orc_pq_create_kmeans(_p(rows), n, dim, ld, m, ksub, seed, kmeans_iters, sample, _p(codes), _p(codebook),
                                    threads)
    if rc:
        raise RuntimeError("orc_pq_create rc=%d" % rc)
    return codes, codebook


def pq_encode(rows, dim, codebook, threads=8):
    """Quantizer::quantize for every row (orc_pq_encode): codes [n, m] u8"""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    cb = np.ascontiguousarray(codebook, dtype=np.float32)
    m, ksub, dsub = cb.shape
    assert m * dsub == dim
    codes = np.zeros((rows.shape[0], m), dtype=np.uint8)
    f = lib().orc_pq_encode
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    f(_p(rows), rows.shape[0], rows.shape[1], m, ksub, dsub, _p(cb), _p(codes), threads)
    return codes


def shuffle(ids, seed):
    v = np.array(ids, dtype=np.uint64)
    lib().orc_shuffle_u64(_p(v), len(v), seed)
    return v


class Index:
    """Hnsw<C> restated: layers (top first) over a flat f32 store."""

    def __init__(self, rows, dim=None, metric=METRIC_COSINE_HALF, sum_mode=SUM_SEQ, _handle=None):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if dim is None:
            rows, dim, ld = pad_rows(rows)
        else:
            ld = rows.shape[1]
        self.rows, self.dim, self.ld, self.metric = rows, dim, ld, metric
        self.h = _handle or lib().orc_index_new(_p(rows), rows.shape[0], dim, ld, metric, sum_mode)
        if not self.h:
            raise ValueError("orc_index_new failed")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_index_free(self.h)
            self.h = None

    def store(self, sum_mode=None):
        return Store(_p(self.rows), self.rows.shape[0], self.dim, self.ld, self.metric,
                     self._sum_mode if sum_mode is None else sum_mode, None, None, 0, 0, 0, 0)

    _sum_mode = SUM_SEQ

    def set_sum_mode(self, m):
        self._sum_mode = m
        lib().orc_index_set_sum_mode(self.h, m)

    @classmethod
    def generate(cls, rows, vids, bp, dim=None, metric=METRIC_COSINE_HALF, sum_mode=SUM_SEQ, threads=8):
        self = cls.__new__(cls)
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if dim is None:
            rows, dim, ld = pad_rows(rows)
        else:
            ld = rows.shape[1]
        vids = np.array(vids, dtype=np.uint64)
        h = lib().orc_generate(_p(rows), rows.shape[0], dim, ld, metric, sum_mode, _p(vids), len(vids),
                               C.byref(bp), threads)
        if not h:
            raise ValueError("orc_generate failed")
        self.rows, self.dim, self.ld, self.metric, self.h = rows, dim, ld, metric, h
        self._sum_mode = sum_mode
        return self

    def push_layer(self, nodes, neighbors, neighborhood_size):
        nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
        neighbors = np.ascontiguousarray(neighbors, dtype=np.uint64).reshape(-1)
        assert len(neighbors) == len(nodes) * neighborhood_size
        lib().orc_index_push_layer(self.h, _p(nodes), _p(neighbors), len(nodes), neighborhood_size)

    def generate_layer(self, vs, neighborhood_size, bp, threads=8):
        vs = np.array(vs, dtype=np.uint64)
        rc = lib().orc_generate_layer(self.h, _p(vs), len(vs), neighborhood_size, C.byref(bp), threads)
        if rc:
            raise RuntimeError("orc_generate_layer rc=%d" % rc)

    @property
    def layer_count(self):
        return lib().orc_index_layer_count(self.h)

    def layer(self, layer_from_top):
        """-> (nodes[u64], neighbors[node_count, W] u64) copies"""
        L = lib().orc_index_layer(self.h, layer_from_top).contents
        n, w = L.node_count, L.neighborhood_size
        nodes = np.ctypeslib.as_array(L.nodes, shape=(n,)).copy()
        nb = np.ctypeslib.as_array(L.neighbors, shape=(n * w,)).copy().reshape(n, w)
        return nodes, nb

    def search(self, queries=None, qids=None, sp=(300, 300, 2), exclude=None, threads=8, stats=False):
        """batched Hnsw::search; queries [nq, dim] (Unstored) or qids (Stored)"""
        sp = SearchParams(*sp)
        if queries is not None:
            q, _, ldq = pad_rows(np.atleast_2d(queries))
            nq = q.shape[0]
            qi = None
        else:
            q, ldq = None, 0
            qi = np.array(qids, dtype=np.uint64)
            nq = len(qi)
        ex = None if exclude is None else np.array(exclude, dtype=np.uint64)
        cap = sp.number_of_candidates
        ids = np.empty((nq, cap), dtype=np.uint64)
        d = np.empty((nq, cap), dtype=np.float32)
        ln = np.zeros(nq, dtype=np.uint64)
        st = np.zeros((nq, 2), dtype=np.uint64)
        rc = lib().orc_search_batch(self.h, _p(q), ldq, _p(qi), nq, sp, _p(ex), _p(ids), _p(d), _p(ln),
                                    _p(st), threads)
        if rc:
            raise RuntimeError("orc_search_batch rc=%d" % rc)
        return (ids, d, ln, st) if stats else (ids, d, ln)

    def search_instrumented(self, queries=None, qids=None, sp=(300, 300, 2), threads=8):
        """batched Hnsw::search_instrumented (lib.rs:667-673) -> ids, d, len, index_distance"""
        sp = SearchParams(*sp)
        if queries is not None:
            q, _, ldq = pad_rows(np.atleast_2d(queries))
            nq, qi = q.shape[0], None
        else:
            q, ldq = None, 0
            qi = np.array(qids, dtype=np.uint64)
            nq = len(qi)
        cap = sp.number_of_candidates
        ids = np.empty((nq, cap), dtype=np.uint64)
        d = np.empty((nq, cap), dtype=np.float32)
        ln = np.zeros(nq, dtype=np.uint64)
        idx = np.zeros(nq, dtype=np.uint64)
        f = lib().orc_search_batch_instrumented
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, SearchParams, C.c_void_p, C.c_void_p,
                      C.c_void_p, C.c_void_p, C.c_int]
        rc = f(self.h, _p(q), ldq, _p(qi), nq, sp, _p(ids), _p(d), _p(ln), _p(idx), threads)
        if rc:
            raise RuntimeError("orc_search_batch_instrumented rc=%d" % rc)
        return ids, d, ln, idx

    def knn(self, k, probe_depth, threads=8):
        n = self.layer(self.layer_count - 1)[0].shape[0]
        ids = np.empty((n, k), dtype=np.uint64)
        d = np.empty((n, k), dtype=np.float32)
        ln = np.zeros(n, dtype=np.uint64)
        rc = lib().orc_knn(self.h, k, probe_depth, _p(ids), _p(d), _p(ln), threads)
        if rc:
            raise RuntimeError("orc_knn rc=%d" % rc)
        return ids, d, ln

    def threshold_nn(self, threshold, probe_depth, initial_search_depth, max_out=64, threads=8):
        n = self.layer(self.layer_count - 1)[0].shape[0]
        ids = np.full((n, max_out), EMPTY, dtype=np.uint64)
        d = np.full((n, max_out), FMAX, dtype=np.float32)
        ln = np.zeros(n, dtype=np.uint64)
        rc = lib().orc_threshold_nn(self.h, threshold, probe_depth, initial_search_depth, max_out, _p(ids),
                                    _p(d), _p(ln), threads)
        if rc:
            raise RuntimeError("orc_threshold_nn rc=%d" % rc)
        return ids, d, ln

    def bruteforce(self, queries, k, threads=8, sum_mode=None):
        q, _, ldq = pad_rows(np.atleast_2d(queries))
        ids = np.empty((q.shape[0], k), dtype=np.uint64)
        d = np.empty((q.shape[0], k), dtype=np.float32)
        st = self.store(sum_mode)
        rc = lib().orc_bruteforce(C.byref(st), _p(q), ldq, q.shape[0], k, _p(ids), _p(d), threads)
        if rc:
            raise RuntimeError("orc_bruteforce rc=%d" % rc)
        return ids, d

    def distance(self, a, b, sum_mode=None):
        a, _, _ = pad_rows(np.atleast_2d(a))
        b, _, _ = pad_rows(np.atleast_2d(b))
        st = self.store(sum_mode)
        return lib().orc_distance(C.byref(st), _p(a), _p(b))

    def link_layer(self, layer_from_top, sp, link_count, threads=8):
        return lib().orc_link_layer(self.h, layer_from_top, SearchParams(*sp), link_count, threads)

    def stochastic_recall_at(self, at, op, threads=8):
        return lib().orc_stochastic_recall_at(self.h, at, C.byref(op), threads)

    def improve_neighbors_upto(self, upto, bp, last_recall=float("nan"), threads=8):
        return lib().orc_improve_neighbors_upto(self.h, upto, C.byref(bp), last_recall, threads)

    def improve_index(self, bp, threads=8, last_recall=None):
        return lib().orc_improve_index_from(self.h, C.byref(bp), float("nan") if last_recall is None else last_recall,
                                            threads)

    # -- product quantisation (pq.rs) --
    def set_pq(self, codes, codebook, table_f16=False):
        """turn this index's store into a PQ store over the given codes / codebooks"""
        self.pq_codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self.pq_codebook = np.ascontiguousarray(codebook, dtype=np.float32)
        m, ksub, dsub = self.pq_codebook.shape
        assert self.pq_codes.shape == (self.rows.shape[0], m)
        lib().orc_index_set_pq(self.h, _p(self.pq_codes), _p(self.pq_codebook), m, ksub, dsub)
        lib().orc_index_set_pq_table_f16(self.h, int(table_f16))

    def pq_search(self, full, queries, sp, quantize_query=False, threads=8, stats=False):
        """QuantizedHnsw::search: `full` is an Index over the f32 rows (its sum mode is used)"""
        sp = SearchParams(*sp)
        q, _, ldq = pad_rows(np.atleast_2d(queries))
        nq, cap = q.shape[0], sp.number_of_candidates
        ids = np.empty((nq, cap), dtype=np.uint64)
        d = np.empty((nq, cap), dtype=np.float32)
        ln = np.zeros(nq, dtype=np.uint64)
        st = np.zeros((nq, 2), dtype=np.uint64)
        fs = full.store()
        rc = lib().orc_pq_search_batch(self.h, C.byref(fs), _p(q), ldq, nq, sp, int(quantize_query), _p(ids), _p(d),
                                       _p(ln), _p(st), threads)
        if rc:
            raise RuntimeError("orc_pq_search_batch rc=%d" % rc)
        return (ids, d, ln, st) if stats else (ids, d, ln)

    def promote_at_layer(self, layer_from_top, bp, threads=8):
        return lib().orc_promote_at_layer(self.h, layer_from_top, C.byref(bp), threads)

    def extend_layer(self, layer_from_top, vecs):
        v = np.ascontiguousarray(vecs, dtype=np.uint64)
        return lib().orc_extend_layer(self.h, layer_from_top, _p(v), len(v))

    def discover_unreachable(self, layer_from_top, sp, threads=8):
        out = C.POINTER(C.c_uint64)()
        n = lib().orc_discover_unreachable(self.h, layer_from_top, SearchParams(*sp), C.byref(out), threads)
        res = np.ctypeslib.as_array(out, shape=(max(n, 1),))[:n].copy()
        C.CDLL(None).free(out)
        return res

    def check_layer_invariants(self):
        return lib().orc_check_layer_invariants(self.h)
