"""Host-side mirror of the reference crate's public surface for the hot path
(terminusdb-labs/parallel-hnsw src/lib.rs:585-1686, src/parameters.rs, src/types.rs),
implemented over the C ABI of libphnsw.  Names, argument meaning and result ordering
follow the reference so tests read like its own tests:

    Hnsw.generate(c, vs, bp)            lib.rs:825-830
    hnsw.search(AbstractVector, sp)     lib.rs:663-665   -> [(VectorId, f32)] sorted (d, id)
    hnsw.improve_index(bp)              lib.rs:1664-1669
    hnsw.knn(k, probe_depth)            lib.rs:905-928
    hnsw.layers[i].nodes / .neighbors   lib.rs:85-91  (top first)

`VectorStore` plays the role of the Comparator (store + metric, bigvec.rs:38-57); the
per-pair compare_raw seam is replaced by batches (distance_batch / search_batch).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BuildParams, OptimizationParams, PhnswError, SearchParams, check, lib

EMPTY = 0xFFFFFFFFFFFFFFFF  # VectorId::MAX / NodeId::MAX (types.rs:8-13)
METRIC_COSINE_HALF, METRIC_ONE_MINUS_DOT, METRIC_L2 = 0, 1, 2


def stream_create_beside(device=0, other_stream=0):
    """phnsw_stream_create_beside: a non-blocking hipStream_t (as an integer) that was seen to run beside `other_stream`
    (0 = the default stream) -- the second lane of a caller that keeps two batches in flight.  Streams that share a
    hardware queue do not overlap; hipStreamDestroy is the caller's."""
    import ctypes
    out = ctypes.c_void_p()
    check(lib().phnsw_stream_create_beside(int(device), ctypes.c_void_p(int(other_stream) or None), ctypes.byref(out)))
    return int(out.value or 0)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def SearchParameters(number_of_candidates=300, upper_layer_candidate_count=300, probe_depth=2):
    return SearchParams(number_of_candidates, upper_layer_candidate_count, probe_depth)


def BuildParameters(**kw):
    bp = BuildParams()
    lib().phnsw_default_build_params(C.byref(bp))
    for k, v in kw.items():
        if not hasattr(bp, k):
            raise TypeError("unknown build parameter %r" % k)
        setattr(bp, k, v)
    return bp


class Stored:
    """AbstractVector::Stored(VectorId)  types.rs:40-43"""

    def __init__(self, vector_id):
        self.id = int(vector_id)


class Unstored:
    """AbstractVector::Unstored(&T)"""

    def __init__(self, vec):
        self.vec = np.ascontiguousarray(vec, dtype=np.float32)


class VectorStore:
    """flat HBM vector store + metric: the Comparator of the GPU path"""

    def __init__(self, rows=None, metric=METRIC_COSINE_HALF, device=0, _handle=None):
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            rows = np.ascontiguousarray(rows, dtype=np.float32)
            assert rows.ndim == 2
            check(lib().phnsw_store_create(_p(rows), rows.shape[0], rows.shape[1], metric, device, C.byref(self._h)))
        self.device = device
        self._refresh()

    def _refresh(self):
        n, dim, ld, m = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_int()
        ptr = C.c_void_p()
        check(lib().phnsw_store_info(self._h, C.byref(n), C.byref(dim), C.byref(ld), C.byref(m), C.byref(ptr)))
        self.n, self.dim, self.ld, self.metric, self.rows_dev = n.value, dim.value, ld.value, m.value, ptr.value

    def append(self, rows):
        """more vectors behind the same comparator; returns the first new VectorId"""
        rows = np.ascontiguousarray(np.atleast_2d(rows), dtype=np.float32)
        assert rows.shape[1] == self.dim
        first = C.c_uint64()
        check(lib().phnsw_store_append(self._h, _p(rows), rows.shape[0], C.byref(first)))
        self._refresh()
        return first.value

    @classmethod
    def synthetic(cls, n, dim, seed=42, first=0, normalize=True, metric=METRIC_COSINE_HALF, device=0):
        """make_random_hnsw's data (bigvec.rs:23-29, 59-65) generated on the GPU"""
        h = C.c_void_p()
        check(lib().phnsw_store_create_synthetic(first, n, dim, seed, int(normalize), metric, device, C.byref(h)))
        return cls(_handle=h, device=device)

    @classmethod
    def clustered(cls, n, dim, seed=42, first=0, n_clusters=1000, noise=1.0, metric=METRIC_COSINE_HALF, device=0):
        """clustered synthetic rows (benchmark dataset)"""
        h = C.c_void_p()
        check(lib().phnsw_store_create_clustered(first, n, dim, seed, n_clusters, noise, metric, device, C.byref(h)))
        return cls(_handle=h, device=device)

    @classmethod
    def from_device(cls, data_ptr, n, dim, ld, metric=METRIC_COSINE_HALF, device=0, keepalive=None):
        h = C.c_void_p()
        check(lib().phnsw_store_create_device(C.c_void_p(data_ptr), n, dim, ld, metric, device, C.byref(h)))
        s = cls(_handle=h, device=device)
        s._keepalive = keepalive
        return s

    def read(self, first=0, count=None, out=None):
        count = self.n - first if count is None else count
        if out is None:
            out = np.empty((count, self.dim), dtype=np.float32)
        assert out.shape == (count, self.dim) and out.dtype == np.float32 and out.flags.c_contiguous
        check(lib().phnsw_store_read(self._h, first, count, _p(out)))
        return out

    def bruteforce_topk(self, queries, k=10):
        """exact k nearest by (distance, id): MFMA GEMM + top-k (ground truth for recall@k)"""
        q = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
        assert q.shape[1] == self.dim
        ids = np.empty((q.shape[0], k), dtype=np.uint64)
        d = np.empty((q.shape[0], k), dtype=np.float32)
        check(lib().phnsw_bruteforce_topk(self._h, _p(q), q.shape[0], k, _p(ids), _p(d)))
        return ids, d

    def bruteforce_topk_device(self, queries_ptr, ldq, nq, k, out_ids_ptr, out_d_ptr, stream=0):
        check(lib().phnsw_bruteforce_topk_device(self._h, C.c_void_p(queries_ptr), ldq, nq, k, C.c_void_p(out_ids_ptr),
                                                 C.c_void_p(out_d_ptr), C.c_void_p(stream or None)))
        return lib().phnsw_bruteforce_last_gemm_ms()

    def compare_vec(self, v, ids):
        """Comparator::compare_vec batched (lib.rs:69-73): distances from v to Stored(ids)"""
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        out = np.empty(len(ids), dtype=np.float32)
        if isinstance(v, Stored):
            check(lib().phnsw_distance_batch(self._h, None, v.id, _p(ids), len(ids), _p(out)))
        else:
            q = v.vec if isinstance(v, Unstored) else np.ascontiguousarray(v, dtype=np.float32)
            assert q.shape[-1] == self.dim
            check(lib().phnsw_distance_batch(self._h, _p(q), 0, _p(ids), len(ids), _p(out)))
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().phnsw_store_destroy(h)
            except Exception:  # interpreter shutdown
                pass


class PqStore(VectorStore):
    """product-quantised view of a VectorStore (pq.rs): u8 codes over per-sub-space codebooks"""

    TABLE_MODES = {"f32": 0, "f16": 1, "u8": 2}

    def __init__(self, full, m, ksub=256, seed=0, table_f16=False, table_mode=None, kmeans_iters=0, kmeans_sample=0,
                 comm=None):
        """table_mode: "f32" (reference arithmetic), "f16" or "u8" (phnsw_pq_set_table_mode); kmeans_iters: Lloyd
        iterations on the codebooks (0 = the reference's random_centroids, pq.rs:261-285); comm: a communicator of
        parallel_hnsw_amd.sharded -- the encode of pq.rs:326-333 split over its ranks, codes all-gathered"""
        h = C.c_void_p()
        cc = C.byref(comm.c_comm(full.device)) if comm is not None else None
        check(lib().phnsw_store_create_pq_sharded(full._h, m, ksub, seed, kmeans_iters, kmeans_sample, cc, C.byref(h)))
        VectorStore.__init__(self, _handle=h, device=full.device)
        mode = self.TABLE_MODES[table_mode] if table_mode is not None else int(table_f16)
        if mode:
            check(lib().phnsw_pq_set_table_mode(self._h, mode))
        self.table_mode = mode
        self.table_f16 = mode == 1

        self.full = full
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().phnsw_pq_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        self.m, self.ksub, self.dsub = a.value, b.value, c.value

    def set_table_mode(self, table_mode):
        """switch the lookup-table storage; "u8" is search-only (see phnsw_pq_set_table_mode)"""
        mode = self.TABLE_MODES[table_mode] if isinstance(table_mode, str) else int(table_mode)
        check(lib().phnsw_pq_set_table_mode(self._h, mode))
        self.table_mode, self.table_f16 = mode, mode == 1

    def quantize(self, rows):
        """Quantizer::quantize  pq.rs:61-71 -> codes [n, m]"""
        rows = np.ascontiguousarray(np.atleast_2d(rows), dtype=np.float32)
        assert rows.shape[1] == self.dim
        out = np.empty((rows.shape[0], self.m), dtype=np.uint8)
        check(lib().phnsw_pq_quantize(self._h, _p(rows), rows.shape[0], _p(out)))
        return out

    def reconstruct(self, codes):
        """Quantizer::reconstruct  pq.rs:73-81 -> rows [n, dim]"""
        codes = np.ascontiguousarray(np.atleast_2d(codes), dtype=np.uint8)
        assert codes.shape[1] == self.m
        out = np.empty((codes.shape[0], self.dim), dtype=np.float32)
        check(lib().phnsw_pq_reconstruct(self._h, _p(codes), codes.shape[0], _p(out)))
        return out

    def codes(self):
        out = np.empty((self.n, self.m), dtype=np.uint8)
        check(lib().phnsw_pq_read(self._h, _p(out), None))
        return out

    def codebook(self):
        out = np.empty((self.m, self.ksub, self.dsub), dtype=np.float32)
        check(lib().phnsw_pq_read(self._h, None, _p(out)))
        return out


class SharedPqStore(VectorStore):
    """the reference's quantizer shape (pq.rs:19-27, 61-81, 261-285): ONE codebook of up to 65535 centroid
    sub-vectors shared by all sub-spaces, u16 codes, quantize = top-1 of an HNSW search over the centroids"""

    def __init__(self, full, centroid_size, number_of_centroids, seed=0, centroid_bp=None, quantized_search=None,
                 centroid_metric=METRIC_L2, comm=None):
        h = C.c_void_p()
        cbp = centroid_bp or BuildParameters()
        qs = quantized_search or SearchParameters()
        cc = C.byref(comm.c_comm(full.device)) if comm is not None else None
        check(lib().phnsw_store_create_pq_shared_sharded(full._h, centroid_size, number_of_centroids, seed, C.byref(cbp),
                                                         C.byref(qs), centroid_metric, cc, C.byref(h)))
        VectorStore.__init__(self, _handle=h, device=full.device)
        self.full = full
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().phnsw_pq_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        self.m, self.ksub, self.dsub = a.value, b.value, c.value

    def codes(self):
        out = np.empty((self.n, self.m), dtype=np.uint16)
        check(lib().phnsw_pq_shared_read(self._h, _p(out), None))
        return out

    def codebook(self):
        out = np.empty((self.ksub, self.dsub), dtype=np.float32)
        check(lib().phnsw_pq_shared_read(self._h, None, _p(out)))
        return out

    def reconstruct_store(self):
        """the reconstructions as an f32 VectorStore (same distance bits as the code rows)"""
        h = C.c_void_p()
        check(lib().phnsw_pq_shared_reconstruct_store(self._h, C.byref(h)))
        return VectorStore(_handle=h, device=self.device)


class QuantizedHnsw:
    """QuantizedHnsw (pq.rs:120-131, 287-364): quantizer + Hnsw over the codes + full comparator"""

    @classmethod
    def reference_shaped(cls, number_of_centroids, comparator, centroid_size, bp=None, centroid_bp=None,
                         quantized_search=None, seed=0, centroid_metric=METRIC_L2, improve_neighbors=False):
        """QuantizedHnsw::new(number_of_centroids, comparator, PqBuildParameters{centroids, hnsw, quantized_search})
        pq.rs:287-344 in its own shape: shared codebook, HNSW quantizer, u16 codes, Hnsw over the quantised
        vectors (built on their materialised reconstructions -- identical distance bits -- and adopted over
        the codes)"""
        self = cls.__new__(cls)
        self.full = comparator
        self.store = SharedPqStore(comparator, centroid_size, number_of_centroids, seed, centroid_bp, quantized_search,
                                   centroid_metric)
        rec = self.store.reconstruct_store()
        g = Hnsw.generate(rec, np.arange(comparator.n, dtype=np.uint64), bp or BuildParameters(promote=0))
        # QuantizedHnsw::improve_neighbors (pq.rs:372-380, what test_pq_recall asserts on): run on the graph over
        # the reconstructions before it is adopted over the codes (same distance bits; the code rows themselves
        # are searched, not built on)
        self.improve_neighbors_recall = (g.improve_neighbors_upto(g.layer_count(), g.build_parameters, None)
                                         if improve_neighbors else None)
        self.hnsw = Hnsw.from_layers(self.store, [(l.nodes, l.neighbors) for l in g.layers], g.build_parameters)
        del g, rec
        return self

    def __init__(self, number_of_centroids, comparator, bp=None, m=None, seed=0, vids=None, table_f16=False,
                 table_mode=None, kmeans_iters=0, kmeans_sample=0, graph=None):
        """QuantizedHnsw::new(number_of_centroids, comparator, bp): per-sub-space codebooks of
        `number_of_centroids` (<= 256) centroids, encode, Hnsw::generate over the codes.

        Without an explicit `bp` the graph over the codes is built with promote=0: a
        reconstruction is not unit length, so a stored code's distance to itself is not within
        the 1e-5 of match_within_epsilon (search.rs:173-187), every vector counts as
        "unreachable" and promote_at_layer's sequential thinning (lib.rs:1243-1262, quadratic in
        the candidates) never finishes at scale -- in the reference as much as here."""
        m = m or max(4, comparator.dim // 8)
        self.full = comparator
        self.store = PqStore(comparator, m, number_of_centroids, seed, table_f16, table_mode, kmeans_iters, kmeans_sample)
        vids = np.arange(comparator.n, dtype=np.uint64) if vids is None else vids
        if graph is not None:
            # `graph`: an Hnsw built over the full-precision comparator; its layers are adopted over the code rows
            # (phnsw_index_from_layers), so the traversal follows the full-precision graph with quantised distances
            self.hnsw = Hnsw.from_layers(self.store, [(l.nodes, l.neighbors) for l in graph.layers], graph.build_parameters)
        else:
            self.hnsw = Hnsw.generate(self.store, vids, bp or BuildParameters(promote=0))

    def search_batch(self, queries, sp=None, quantize_query=False, stats=False):
        sp = sp or SearchParameters()
        q = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
        assert q.shape[1] == self.full.dim
        nq, ef = q.shape[0], sp.number_of_candidates
        ids = np.empty((nq, ef), dtype=np.uint64)
        d = np.empty((nq, ef), dtype=np.float32)
        ln = np.zeros(nq, dtype=np.uint64)
        st = np.zeros((nq, 2), dtype=np.uint64) if stats else None
        check(lib().phnsw_pq_search_batch(self.hnsw._h, self.full._h, _p(q), nq, C.byref(sp), int(quantize_query),
                                          _p(ids), _p(d), _p(ln), _p(st)))
        return (ids, d, ln, st) if stats else (ids, d, ln)

    def search_batch_device(self, nq, sp, queries, ldq, out_ids, out_d, out_len, status, out_stats=0, stream=0):
        check(lib().phnsw_pq_search_batch_device(self.hnsw._h, self.full._h, C.c_void_p(queries), ldq, nq, C.byref(sp),
                                                 C.c_void_p(out_ids), C.c_void_p(out_d), C.c_void_p(out_len),
                                                 C.c_void_p(out_stats or None), C.c_void_p(status),
                                                 C.c_void_p(stream or None)))

    def search(self, v, sp=None):
        """QuantizedHnsw::search(v, sp)  pq.rs:346-364"""
        ids, d, ln = self.search_batch(v.vec if isinstance(v, Unstored) else v, sp)
        return [(int(ids[0, i]), d[0, i]) for i in range(int(ln[0]))]

    def vector_count(self):
        return self.hnsw.vector_count()

    # QuantizedHnsw forwards these to the Hnsw over the codes  pq.rs:366-411
    def improve_index(self, bp=None, last_recall=None, progress=None):
        return self.hnsw.improve_index(bp or self.build_parameters_for_improve_index(), last_recall, progress)

    def improve_neighbors_upto(self, upto, bp=None, last_recall=None):
        return self.hnsw.improve_neighbors_upto(upto, bp or self.build_parameters_for_improve_index(), last_recall)

    def promote_at_layer(self, layer_from_top, bp=None):
        return self.hnsw.promote_at_layer(layer_from_top, bp or self.hnsw.build_parameters)

    def threshold_nn(self, threshold, probe_depth, initial_search_depth, max_out=64):
        return self.hnsw.threshold_nn(threshold, probe_depth, initial_search_depth, max_out)

    def stochastic_recall(self, op=None):
        return self.hnsw.stochastic_recall(op)

    def build_parameters_for_improve_index(self):
        return self.hnsw.build_parameters


class Layer:
    """Layer { neighborhood_size, nodes, neighbors }  lib.rs:85-91 (host copies, u64)"""

    def __init__(self, nodes, neighbors, neighborhood_size):
        self.nodes = nodes
        self.neighbors = neighbors
        self.neighborhood_size = neighborhood_size

    def node_count(self):
        return len(self.nodes)

    def get_neighbors(self, n):
        row = self.neighbors[n]
        return row[row != EMPTY]  # trailing sentinels trimmed (lib.rs:114-125)


class Hnsw:
    def __init__(self, store, handle, build_parameters=None):
        self.store = store
        self._h = handle
        self.build_parameters = build_parameters or BuildParameters()

    # -- construction -------------------------------------------------------
    @classmethod
    def generate(cls, c, vs, bp=None, progress=None):
        """Hnsw::generate(c, vs, bp, progress)  lib.rs:825-893"""
        bp = bp or BuildParameters()
        vs = np.ascontiguousarray(vs, dtype=np.uint64)
        h = C.c_void_p()
        cb = _lib.PROGRESS_CB(progress) if progress else None
        check(lib().phnsw_build(c._h, _p(vs), len(vs), C.byref(bp), cb, None, C.byref(h)))
        return cls(c, h, bp)

    @classmethod
    def from_layers(cls, c, layers, bp=None):
        """adopt [(nodes, neighbors[n, W])...] top first, e.g. deserialised from the Rust crate"""
        L = len(layers)
        nodes = [np.ascontiguousarray(l[0], dtype=np.uint64) for l in layers]
        nbs = [np.ascontiguousarray(l[1], dtype=np.uint64).reshape(len(nodes[i]), -1) for i, l in enumerate(layers)]
        counts = np.array([len(x) for x in nodes], dtype=np.uint64)
        widths = np.array([nb.shape[1] for nb in nbs], dtype=np.uint64)
        pn = (C.c_void_p * L)(*[x.ctypes.data for x in nodes])
        pb = (C.c_void_p * L)(*[x.ctypes.data for x in nbs])
        h = C.c_void_p()
        check(lib().phnsw_index_from_layers(c._h, L, _p(counts), _p(widths), pn, pb, C.byref(h)))
        return cls(c, h, bp)

    def generate_layer(self, vs, neighborhood_size, bp=None):
        vs = np.ascontiguousarray(vs, dtype=np.uint64)
        check(lib().phnsw_generate_layer(self._h, _p(vs), len(vs), neighborhood_size,
                                         C.byref(bp or self.build_parameters)))

    def link_layer_to_better_neighbors(self, layer_from_top, sp):
        """lib.rs:1070-1082; returns the number of new edges"""
        added = C.c_uint64()
        check(lib().phnsw_link_layer(self._h, layer_from_top, C.byref(sp), self.build_parameters.neighborhood_size,
                                     C.byref(added)))
        return added.value

    def improve_index(self, bp=None, last_recall=None, progress=None):
        """Hnsw::improve_index(bp, last_recall: Option<f32>, progress)  lib.rs:1664-1686"""
        out = C.c_float()
        cb = _lib.PROGRESS_CB(progress) if progress else None
        check(lib().phnsw_improve_index(self._h, C.byref(bp or self.build_parameters),
                                        float("nan") if last_recall is None else last_recall, cb, None, C.byref(out)))
        return out.value

    def improve_neighbors_upto(self, upto, bp=None, last_recall=None):
        out = C.c_float()
        check(lib().phnsw_improve_neighbors_upto(self._h, upto, C.byref(bp or self.build_parameters),
                                                 float("nan") if last_recall is None else last_recall,
                                                 C.byref(out)))
        return out.value

    def extend_layer(self, layer_from_top, vecs):
        """Hnsw::extend_layer  lib.rs:1039-1068"""
        v = np.ascontiguousarray(vecs, dtype=np.uint64)
        check(lib().phnsw_extend_layer(self._h, layer_from_top, _p(v), len(v)))

    def promote_at_layer(self, layer_from_top, bp=None):
        """lib.rs:1273-1427"""
        out = C.c_int()
        check(lib().phnsw_promote_at_layer(self._h, layer_from_top, C.byref(bp or self.build_parameters), C.byref(out)))
        return bool(out.value)

    def discover_unreachable_vectors(self, layer_from_top, sp):
        """lib.rs:1002-1037"""
        n = self._layer(layer_from_top).node_count()
        out = np.empty(n, dtype=np.uint64)
        cnt = C.c_uint64()
        check(lib().phnsw_discover_unreachable(self._h, layer_from_top, C.byref(sp), _p(out), C.byref(cnt)))
        return out[:cnt.value].copy()

    def stochastic_recall_at(self, at, op=None):
        out = C.c_float()
        op = op or self.build_parameters.optimization
        check(lib().phnsw_stochastic_recall_at(self._h, at, C.byref(op), C.byref(out)))
        return out.value

    def stochastic_recall(self, op=None):
        return self.stochastic_recall_at(self.layer_count() - 1, op)

    # -- accessors ----------------------------------------------------------
    def layer_count(self):
        return lib().phnsw_index_layer_count(self._h)

    def _layer(self, lft):
        n, w = C.c_uint64(), C.c_uint64()
        check(lib().phnsw_index_layer_info(self._h, lft, C.byref(n), C.byref(w)))
        nodes = np.empty(n.value, dtype=np.uint64)
        nb = np.empty((n.value, w.value), dtype=np.uint64)
        check(lib().phnsw_index_layer_read(self._h, lft, _p(nodes), _p(nb)))
        return Layer(nodes, nb, w.value)

    @property
    def layers(self):
        """Vec<Layer>, top first (lib.rs:587)"""
        return [self._layer(i) for i in range(self.layer_count())]

    def get_layer(self, i):
        """counts from the bottom (lib.rs:604-606)"""
        return self._layer(self.layer_count() - i - 1)

    def get_layer_from_top(self, i):
        """lib.rs:617-624 (None past the stack)"""
        return self._layer(i) if 0 <= i < self.layer_count() else None

    def get_layer_above(self, i):
        """lib.rs:631-637: the layer above layer-from-top i"""
        return None if i == 0 else self.get_layer_from_top(i - 1)

    def neighborhood_size(self):
        return int(self.build_parameters.neighborhood_size)  # lib.rs:596-598

    def zero_neighborhood_size(self):
        return int(self.build_parameters.zero_layer_neighborhood_size)  # lib.rs:600-602

    def comparator(self):
        return self.store  # lib.rs:648-650

    def __len__(self):
        return self.vector_count()  # lib.rs:895-897

    def is_empty(self):
        return self.vector_count() == 0  # lib.rs:899-901

    def all_vectors(self):
        """lib.rs:968-975: the bottom layer's VectorIds"""
        return self._layer(self.layer_count() - 1).nodes

    def supers_for_layer(self, layer_id):
        """lib.rs:977-984: the VectorIds of the layer above `layer_id` (counted from the bottom); the top
        layer's only super is its entry node"""
        if self.layer_count() == layer_id + 1:
            return self.get_layer(layer_id).nodes[0:1]
        return self.get_layer(layer_id + 1).nodes

    def entry_vector(self):
        return int(self._layer(0).nodes[0])

    def vector_count(self):
        n = C.c_uint64()
        check(lib().phnsw_index_layer_info(self._h, self.layer_count() - 1, C.byref(n), None))
        return n.value

    # -- search -------------------------------------------------------------
    def search_batch(self, queries=None, qids=None, sp=None, exclude=None, upto=0, stats=False, k=None):
        """Hnsw::search for many queries; returns (ids[nq, ef] u64, d[nq, ef] f32, len[nq]); with k only the best k
        results of each query are transferred (phnsw_search_batch_topk): ids[nq, k], d[nq, k]"""
        sp = sp or SearchParameters()
        ef = sp.number_of_candidates
        q = qi = None
        if queries is not None:
            q = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
            assert q.shape[1] == self.store.dim
            nq = q.shape[0]
        else:
            qi = np.ascontiguousarray(qids, dtype=np.uint64)
            nq = len(qi)
        w = ef if k is None else int(k)
        ids = np.empty((nq, w), dtype=np.uint64)
        d = np.empty((nq, w), dtype=np.float32)
        ln = np.zeros(nq, dtype=np.uint64)
        st = np.zeros((nq, 2), dtype=np.uint64) if stats else None
        ex = None if exclude is None else np.ascontiguousarray(exclude, dtype=np.uint64)
        if k is not None:
            assert not stats
            check(lib().phnsw_search_batch_topk(self._h, _p(q), _p(qi), nq, C.byref(sp), upto, _p(ex), w, _p(ids), _p(d),
                                                _p(ln)))
        elif queries is not None:
            check(lib().phnsw_search_batch(self._h, _p(q), nq, C.byref(sp), upto, _p(ex), _p(ids), _p(d), _p(ln), _p(st)))
        else:
            check(lib().phnsw_search_batch_stored(self._h, _p(qi), nq, C.byref(sp), upto, _p(ex), _p(ids), _p(d),
                                                  _p(ln), _p(st)))
        return (ids, d, ln, st) if stats else (ids, d, ln)

    def search_instrumented_batch(self, queries=None, qids=None, sp=None):
        """Hnsw::search_instrumented (lib.rs:667-673) for many queries -> ids, d, len, index_distance[nq] u64"""
        sp = sp or SearchParameters()
        ef = sp.number_of_candidates
        q = qi = None
        if queries is not None:
            q = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
            assert q.shape[1] == self.store.dim
            nq = q.shape[0]
        else:
            qi = np.ascontiguousarray(qids, dtype=np.uint64)
            nq = len(qi)
        ids = np.empty((nq, ef), dtype=np.uint64)
        d = np.empty((nq, ef), dtype=np.float32)
        ln = np.zeros(nq, dtype=np.uint64)
        idx = np.zeros(nq, dtype=np.uint64)
        check(lib().phnsw_search_instrumented(self._h, _p(q), _p(qi), nq, C.byref(sp), _p(ids), _p(d), _p(ln), _p(idx)))
        return ids, d, ln, idx

    def search_instrumented(self, v, sp=None):
        """Hnsw::search_instrumented(v, sp) -> (Vec<(VectorId, f32)>, usize)  lib.rs:667-673"""
        if isinstance(v, Stored):
            ids, d, ln, idx = self.search_instrumented_batch(qids=[v.id], sp=sp)
        else:
            ids, d, ln, idx = self.search_instrumented_batch(queries=v.vec if isinstance(v, Unstored) else v, sp=sp)
        return [(int(ids[0, i]), d[0, i]) for i in range(int(ln[0]))], int(idx[0])

    def search_batch_device(self, nq, sp, out_ids, out_d, out_len, status, queries=0, ldq=0, qids=0, exclude=0,
                            out_stats=0, upto=0, stream=0):
        """zero-copy launch: every argument is a device pointer (int); u32 ids; returns after
        enqueueing on `stream` (a hipStream_t value, 0 = default stream)"""
        check(lib().phnsw_search_batch_device(self._h, C.c_void_p(queries or None), ldq, C.c_void_p(qids or None), nq,
                                              C.byref(sp), upto, C.c_void_p(exclude or None), C.c_void_p(out_ids),
                                              C.c_void_p(out_d), C.c_void_p(out_len), C.c_void_p(out_stats or None),
                                              C.c_void_p(status), C.c_void_p(stream or None)))

    def search(self, v, sp=None):
        """Hnsw::search(v, sp) -> Vec<(VectorId, f32)>  lib.rs:663-665"""
        if isinstance(v, Stored):
            ids, d, ln = self.search_batch(qids=[v.id], sp=sp)
        else:
            ids, d, ln = self.search_batch(queries=v.vec if isinstance(v, Unstored) else v, sp=sp)
        return [(int(ids[0, i]), d[0, i]) for i in range(int(ln[0]))]

    def search_upto(self, v, sp, upto_layer_from_top):
        """lib.rs:654-661"""
        if isinstance(v, Stored):
            ids, d, ln = self.search_batch(qids=[v.id], sp=sp, upto=upto_layer_from_top)
        else:
            ids, d, ln = self.search_batch(queries=v.vec, sp=sp, upto=upto_layer_from_top)
        return [(int(ids[0, i]), d[0, i]) for i in range(int(ln[0]))]

    def knn(self, k, probe_depth):
        """Hnsw::knn  lib.rs:905-928 -> [(VectorId, [(VectorId, f32)])]"""
        n = self.vector_count()
        ids = np.empty((n, k), dtype=np.uint64)
        d = np.empty((n, k), dtype=np.float32)
        ln = np.zeros(n, dtype=np.uint64)
        check(lib().phnsw_knn(self._h, k, probe_depth, _p(ids), _p(d), _p(ln)))
        nodes = self._layer(self.layer_count() - 1).nodes
        return [(int(nodes[i]), [(int(ids[i, j]), d[i, j]) for j in range(int(ln[i]))]) for i in range(n)]

    def threshold_nn(self, threshold, probe_depth, initial_search_depth, max_out=64):
        """Hnsw::threshold_nn  lib.rs:930-962 -> [(VectorId, [(VectorId, f32)])]"""
        n = self.vector_count()
        ids = np.empty((n, max_out), dtype=np.uint64)
        d = np.empty((n, max_out), dtype=np.float32)
        ln = np.zeros(n, dtype=np.uint64)
        check(lib().phnsw_threshold_nn(self._h, threshold, probe_depth, initial_search_depth, max_out, _p(ids), _p(d),
                                       _p(ln)))
        nodes = self._layer(self.layer_count() - 1).nodes
        return [(int(nodes[i]), [(int(ids[i, j]), d[i, j]) for j in range(int(ln[i]))]) for i in range(n)]

    # -- Serializable (lib.rs:1688-1699, serialize.rs) -------------------------------
    def serialize(self, path):
        check(lib().phnsw_index_serialize(self._h, str(path).encode()))

    @classmethod
    def deserialize(cls, path, store):
        """Hnsw::deserialize(path, params): `store` plays the role of the comparator params"""
        h = C.c_void_p()
        check(lib().phnsw_index_deserialize(store._h, str(path).encode(), C.byref(h)))
        bp = BuildParams()
        check(lib().phnsw_index_build_params(h, C.byref(bp)))
        return cls(store, h, bp)

    def counters(self):
        """(distance evaluations, hops) of every search launched on this index, build rounds included"""
        a, b = C.c_uint64(), C.c_uint64()
        check(lib().phnsw_index_counters(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kernel_ms(self):
        ms = C.c_float()
        check(lib().phnsw_last_search_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def dispatches(self):
        """the last descent dispatch by dispatch: [{"layers": (lo, hi), "ms", "n_dist", "n_hops", "n_table"}];
        entry 0 is the dense-top-layer tile pass (layers (0, 0)); n_table of n_dist evaluations were table look-ups"""
        cap = 32
        cnt = C.c_uint32()
        ms = np.zeros(cap, dtype=np.float32)
        nd, nh = np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint64)
        lo, hi = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint32)
        check(lib().phnsw_last_search_dispatches(self._h, cap, C.byref(cnt), _p(ms), _p(nd), _p(nh), _p(lo), _p(hi)))
        nt = np.zeros(cap, dtype=np.uint64)
        c2 = C.c_uint32()
        check(lib().phnsw_last_search_table_evals(self._h, cap, C.byref(c2), _p(nt)))
        return [{"layers": (int(lo[i]), int(hi[i])), "ms": float(ms[i]), "n_dist": int(nd[i]), "n_hops": int(nh[i]),
                 "n_table": int(nt[i])} for i in range(min(cap, cnt.value))]

    def dense_top_layers(self, number_of_candidates):
        """(layers walked through the dense distance table, nodes of the largest, built on the matrix cores?)"""
        t, n, m = C.c_uint32(), C.c_uint64(), C.c_uint32()
        check(lib().phnsw_dense_top_layers(self._h, int(number_of_candidates), C.byref(t), C.byref(n), C.byref(m)))
        return int(t.value), int(n.value), bool(m.value)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().phnsw_index_destroy(h)
            except Exception:  # interpreter shutdown
                pass
