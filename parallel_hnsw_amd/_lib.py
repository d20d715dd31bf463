"""ctypes binding of libphnsw.so (include/phnsw.h).  There is no CPU fallback: importing
works anywhere, but every compute call raises PhnswError without a gfx950 GPU, and loading
fails loudly when the shared library has not been built (python __graft_entry__.py build)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PHNSW_LIB_PATH") or os.path.join(_HERE, "libphnsw.so")  # the override: A/B experiments
CSRC = os.path.join(_HERE, "csrc")


class PhnswError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("phnsw error %d: %s" % (code, msg))
        self.code = code


class SearchParams(C.Structure):
    """SearchParameters  (reference src/parameters.rs:3-18)"""
    _fields_ = [("number_of_candidates", C.c_uint64),
                ("upper_layer_candidate_count", C.c_uint64),
                ("probe_depth", C.c_uint64)]


class OptimizationParams(C.Structure):
    """OptimizationParameters  (reference src/parameters.rs:20-40)"""
    _fields_ = [("promotion_threshold", C.c_float), ("neighborhood_threshold", C.c_float),
                ("recall_proportion", C.c_float), ("promotion_proportion", C.c_float),
                ("search", SearchParams)]


class BuildParams(C.Structure):
    """BuildParameters  (reference src/parameters.rs:42-64) + seed / max_link_rounds"""
    _fields_ = [("order", C.c_uint64), ("zero_layer_neighborhood_size", C.c_uint64),
                ("neighborhood_size", C.c_uint64), ("optimization", OptimizationParams),
                ("initial_partition_search", SearchParams), ("seed", C.c_uint64),
                ("max_link_rounds", C.c_uint64), ("promote", C.c_uint64)]


PROGRESS_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64)

# ---- sharded build (include/phnsw.h: phnsw_comm, phnsw_sharded_stats, phnsw_shard_engine)
AllGatherFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
AllReduceFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32)


class Comm(C.Structure):
    """phnsw_comm"""
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("host_buffers", C.c_uint32), ("emulate", C.c_uint32),
                ("ctx", C.c_void_p), ("all_gather", AllGatherFn), ("all_reduce_sum", AllReduceFn)]


class ShardedStats(C.Structure):
    """phnsw_sharded_stats"""
    _fields_ = [("seconds_total", C.c_double), ("seconds_sharded", C.c_double), ("seconds_replicated", C.c_double),
                ("seconds_comm", C.c_double), ("seconds_others", C.c_double), ("all_gather_bytes", C.c_uint64),
                ("all_gather_calls", C.c_uint64), ("all_reduce_calls", C.c_uint64), ("phases", C.c_uint64),
                ("phases_whole", C.c_uint64), ("seconds_by_phase", C.c_double * 10)]


SHARD_PHASE_NAMES = ("layer_init_search", "layer_seed", "link_search", "recall_hits", "discover_hits", "plan", "layer_begin",
                     "layer_finish", "link_apply", "promote_from_hits")


_vpx, _u64x, _u32x = C.c_void_p, C.c_uint64, C.c_uint32


class ShardEngine(C.Structure):
    """phnsw_shard_engine: the phases behind the sharded driver as callbacks (raw pointers as ints)"""
    _fields_ = [
        ("ctx", C.c_void_p), ("id_bytes", C.c_uint32), ("host_buffers", C.c_uint32),
        ("alloc", C.CFUNCTYPE(C.c_void_p, _vpx, _u64x)),
        ("release", C.CFUNCTYPE(None, _vpx, _vpx)),
        ("copy2d", C.CFUNCTYPE(C.c_int, _vpx, _vpx, _u64x, _vpx, _u64x, _u64x, _u64x)),
        ("plan", C.CFUNCTYPE(C.c_int, _vpx, _vpx, _u64x, _vpx, _vpx, _u32x, C.POINTER(C.c_uint32))),
        ("layer_begin", C.CFUNCTYPE(C.c_int, _vpx, _vpx, _u64x, _u64x, C.POINTER(C.c_int), C.POINTER(C.c_uint32))),
        ("layer_init_search", C.CFUNCTYPE(C.c_int, _vpx, _u64x, _u64x, _vpx, _vpx, _vpx)),
        ("layer_seed", C.CFUNCTYPE(C.c_int, _vpx, _vpx, _vpx, _vpx, _u64x, _u64x, _vpx, _vpx)),
        ("layer_finish", C.CFUNCTYPE(C.c_int, _vpx, _vpx, _vpx)),
        ("layer_count", C.CFUNCTYPE(C.c_uint32, _vpx)),
        ("layer_nodes", C.CFUNCTYPE(C.c_uint64, _vpx, _u32x)),
        ("link_search", C.CFUNCTYPE(C.c_int, _vpx, _u32x, C.POINTER(SearchParams), _u64x, _u64x, _u64x, _vpx, _vpx, _vpx)),
        ("link_apply", C.CFUNCTYPE(C.c_int, _vpx, _u32x, _u64x, _vpx, _vpx, _vpx, C.POINTER(C.c_uint64))),
        ("recall_hits", C.CFUNCTYPE(C.c_int, _vpx, _u32x, C.POINTER(OptimizationParams), _u64x, _u64x,
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))),
        ("discover_hits", C.CFUNCTYPE(C.c_int, _vpx, _u32x, C.POINTER(SearchParams), _u64x, _u64x, _vpx)),
        ("promote_from_hits", C.CFUNCTYPE(C.c_int, _vpx, _u32x, _vpx, C.POINTER(C.c_int))),
        ("layer_cells", C.CFUNCTYPE(C.c_int, _vpx, _u64x, _u64x, _vpx)),
        ("layer_set_cells", C.CFUNCTYPE(C.c_int, _vpx, _vpx)),
    ]

# name -> (restype, argtypes): every symbol include/phnsw.h declares
_vp, _u64, _u32, _i32, _f32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_float
_pp = C.POINTER(C.c_void_p)
SYMBOLS = {
    "phnsw_default_search_params": (None, [C.POINTER(SearchParams)]),
    "phnsw_default_build_params": (None, [C.POINTER(BuildParams)]),
    "phnsw_last_error": (C.c_char_p, []),
    "phnsw_device_count": (_i32, []),
    "phnsw_store_create": (_i32, [_vp, _u64, _u32, _i32, _i32, _pp]),
    "phnsw_store_append": (_i32, [_vp, _vp, _u64, _vp]),
    "phnsw_store_create_device": (_i32, [_vp, _u64, _u32, _u32, _i32, _i32, _pp]),
    "phnsw_store_create_synthetic": (_i32, [_u64, _u64, _u32, _u64, _i32, _i32, _i32, _pp]),
    "phnsw_store_create_clustered": (_i32, [_u64, _u64, _u32, _u64, _u32, _f32, _i32, _i32, _pp]),
    "phnsw_store_info": (_i32, [_vp, C.POINTER(_u64), C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_i32), _pp]),
    "phnsw_store_read": (_i32, [_vp, _u64, _u64, _vp]),
    "phnsw_store_destroy": (None, [_vp]),
    "phnsw_distance_batch": (_i32, [_vp, _vp, _u64, _vp, _u64, _vp]),
    "phnsw_index_from_layers": (_i32, [_vp, _u32, _vp, _vp, _vp, _vp, _pp]),
    "phnsw_build": (_i32, [_vp, _vp, _u64, C.POINTER(BuildParams), _vp, _vp, _pp]),
    "phnsw_generate_layer": (_i32, [_vp, _vp, _u64, _u64, C.POINTER(BuildParams)]),
    "phnsw_link_layer": (_i32, [_vp, _u32, C.POINTER(SearchParams), _u64, C.POINTER(_u64)]),
    "phnsw_improve_index": (_i32, [_vp, C.POINTER(BuildParams), _f32, _vp, _vp, C.POINTER(_f32)]),
    "phnsw_improve_neighbors_upto": (_i32, [_vp, _u32, C.POINTER(BuildParams), _f32, C.POINTER(_f32)]),
    "phnsw_extend_layer": (_i32, [_vp, _u32, _vp, _u64]),
    "phnsw_promote_at_layer": (_i32, [_vp, _u32, C.POINTER(BuildParams), C.POINTER(_i32)]),
    "phnsw_discover_unreachable": (_i32, [_vp, _u32, C.POINTER(SearchParams), _vp, C.POINTER(_u64)]),
    "phnsw_stochastic_recall_at": (_i32, [_vp, _u32, C.POINTER(OptimizationParams), C.POINTER(_f32)]),
    "phnsw_index_destroy": (None, [_vp]),
    "phnsw_index_layer_count": (_u32, [_vp]),
    "phnsw_index_layer_info": (_i32, [_vp, _u32, C.POINTER(_u64), C.POINTER(_u64)]),
    "phnsw_index_layer_read": (_i32, [_vp, _u32, _vp, _vp]),
    "phnsw_search_batch": (_i32, [_vp, _vp, _u64, C.POINTER(SearchParams), _u32, _vp, _vp, _vp, _vp, _vp]),
    "phnsw_search_batch_stored": (_i32, [_vp, _vp, _u64, C.POINTER(SearchParams), _u32, _vp, _vp, _vp, _vp, _vp]),
    "phnsw_search_batch_topk": (_i32, [_vp, _vp, _vp, _u64, C.POINTER(SearchParams), _u32, _vp, _u64, _vp, _vp, _vp]),
    "phnsw_stream_create_beside": (_i32, [_i32, _vp, C.POINTER(_vp)]),
    "phnsw_search_instrumented": (_i32, [_vp, _vp, _vp, _u64, C.POINTER(SearchParams), _vp, _vp, _vp, _vp]),
    "phnsw_search_batch_device": (_i32, [_vp, _vp, _u32, _vp, _u64, C.POINTER(SearchParams), _u32, _vp, _vp, _vp,
                                         _vp, _vp, _vp, _vp]),
    "phnsw_index_counters": (_i32, [_vp, _vp, _vp]),
    "phnsw_last_search_kernel_ms": (_i32, [_vp, C.POINTER(_f32)]),
    "phnsw_last_search_dispatches": (_i32, [_vp, _u32, C.POINTER(_u32), _vp, _vp, _vp, _vp, _vp]),
    "phnsw_last_search_table_evals": (_i32, [_vp, _u32, C.POINTER(_u32), _vp]),
    "phnsw_dense_top_layers": (_i32, [_vp, _u64, C.POINTER(_u32), C.POINTER(_u64), C.POINTER(_u32)]),
    "phnsw_index_create": (_i32, [_vp, C.POINTER(BuildParams), _pp]),
    "phnsw_build_plan": (_i32, [_vp, _u64, C.POINTER(BuildParams), _vp, _vp, _u32, C.POINTER(_u32)]),
    "phnsw_layer_begin": (_i32, [_vp, _vp, _u64, _u64, C.POINTER(BuildParams), C.POINTER(_i32)]),
    "phnsw_layer_begin_sharded": (_i32, [_vp, _vp, _u64, _u64, C.POINTER(BuildParams), C.POINTER(_i32), C.POINTER(_i32)]),
    "phnsw_layer_cells_device": (_i32, [_vp, _u64, _u64, _vp]),
    "phnsw_layer_set_cells_device": (_i32, [_vp, _vp]),
    "phnsw_layer_init_search_device": (_i32, [_vp, C.POINTER(BuildParams), _u64, _u64, _vp, _vp, _vp]),
    "phnsw_layer_seed_device": (_i32, [_vp, C.POINTER(BuildParams), _vp, _vp, _vp, _u64, _u64, _vp, _vp]),
    "phnsw_layer_finish_device": (_i32, [_vp, _vp, _vp]),
    "phnsw_link_search_device": (_i32, [_vp, _u32, C.POINTER(SearchParams), _u64, _u64, _u64, _vp, _vp, _vp]),
    "phnsw_link_apply_device": (_i32, [_vp, _u32, _u64, _vp, _vp, _vp, C.POINTER(_u64)]),
    "phnsw_discover_hits_device": (_i32, [_vp, _u32, C.POINTER(SearchParams), _u64, _u64, _vp]),
    "phnsw_promote_at_layer_hits_device": (_i32, [_vp, _u32, C.POINTER(BuildParams), _vp, C.POINTER(_i32)]),
    "phnsw_recall_hits": (_i32, [_vp, _u32, C.POINTER(OptimizationParams), _u64, _u64, C.POINTER(_u64),
                                 C.POINTER(_u64)]),
    "phnsw_build_sharded": (_i32, [_vp, _vp, _u64, C.POINTER(BuildParams), C.POINTER(Comm), PROGRESS_CB, _vp, _pp,
                                   C.POINTER(ShardedStats)]),
    "phnsw_improve_index_sharded": (_i32, [_vp, C.POINTER(BuildParams), _f32, C.POINTER(Comm), C.POINTER(_f32),
                                           C.POINTER(ShardedStats)]),
    "phnsw_sharded_tuning": (_i32, [_u64, _u32, _u64]),
    "phnsw_comm_rccl_unique_id": (_i32, [_vp]),
    "phnsw_comm_rccl_create": (_i32, [_vp, _u32, _u32, _i32, C.POINTER(C.POINTER(Comm))]),
    "phnsw_comm_destroy": (None, [C.POINTER(Comm)]),
    "phnsw_comm_selftest": (_i32, [C.POINTER(Comm), _u64]),
    "phnsw_comm_benchmark": (_i32, [C.POINTER(Comm), _u64, _u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "phnsw_build_sharded_engine": (_i32, [C.POINTER(ShardEngine), _vp, _u64, C.POINTER(BuildParams), C.POINTER(Comm),
                                          C.POINTER(ShardedStats)]),
    "phnsw_store_create_pq": (_i32, [_vp, _u32, _u32, _u64, _pp]),
    "phnsw_store_create_pq_kmeans": (_i32, [_vp, _u32, _u32, _u64, _u32, _u64, _pp]),
    "phnsw_store_create_pq_shared": (_i32, [_vp, _u32, _u32, _u64, C.POINTER(BuildParams), C.POINTER(SearchParams), _i32, _pp]),
    "phnsw_store_create_pq_sharded": (_i32, [_vp, _u32, _u32, _u64, _u32, _u64, C.POINTER(Comm), _pp]),
    "phnsw_store_create_pq_shared_sharded": (_i32, [_vp, _u32, _u32, _u64, C.POINTER(BuildParams), C.POINTER(SearchParams),
                                                    _i32, C.POINTER(Comm), _pp]),
    "phnsw_pq_shared_read": (_i32, [_vp, _vp, _vp]),
    "phnsw_pq_shared_reconstruct_store": (_i32, [_vp, _pp]),
    "phnsw_pq_info": (_i32, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "phnsw_pq_quantize": (_i32, [_vp, _vp, _u64, _vp]),
    "phnsw_pq_reconstruct": (_i32, [_vp, _vp, _u64, _vp]),
    "phnsw_pq_set_table_mode": (_i32, [_vp, _i32]),
    "phnsw_pq_set_table_f16": (_i32, [_vp, _i32]),
    "phnsw_pq_read": (_i32, [_vp, _vp, _vp]),
    "phnsw_pq_search_batch": (_i32, [_vp, _vp, _vp, _u64, C.POINTER(SearchParams), _i32, _vp, _vp, _vp, _vp]),
    "phnsw_pq_search_batch_device": (_i32, [_vp, _vp, _vp, _u32, _u64, C.POINTER(SearchParams), _vp, _vp, _vp, _vp,
                                            _vp, _vp]),
    "phnsw_index_serialize": (_i32, [_vp, C.c_char_p]),
    "phnsw_index_deserialize": (_i32, [_vp, C.c_char_p, _pp]),
    "phnsw_index_build_params": (_i32, [_vp, C.POINTER(BuildParams)]),
    "phnsw_bruteforce_topk": (_i32, [_vp, _vp, _u64, _u32, _vp, _vp]),
    "phnsw_bruteforce_topk_device": (_i32, [_vp, _vp, _u32, _u64, _u32, _vp, _vp, _vp]),
    "phnsw_bruteforce_last_gemm_ms": (_f32, []),
    "phnsw_threshold_nn": (_i32, [_vp, _f32, _u64, _u64, _u64, _vp, _vp, _vp]),
    "phnsw_knn": (_i32, [_vp, _u64, _u64, _vp, _vp, _vp]),
}

_lib = None


def build_lib(force=False):
    """hipcc --offload-arch=gfx950 the HIP sources into parallel_hnsw_amd/libphnsw.so"""
    args = ["make", "-C", CSRC, "-s", "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libphnsw.so is not built (run `python __graft_entry__.py` or "
                              "`make -C parallel_hnsw_amd/csrc`); there is no CPU fallback")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  If it is
        # importable, load it first so that libphnsw binds to that copy (same soname) instead
        # of bringing /opt/rocm's into the process next to it -- with two runtimes the second
        # one finds no GPU.
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().phnsw_last_error()
        raise PhnswError(rc, msg.decode() if msg else "")
