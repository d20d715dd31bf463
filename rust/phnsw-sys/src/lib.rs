//! Raw bindings of `include/phnsw.h` -- the C ABI of libphnsw, the MI355X-native replacement of
//! parallel-hnsw's hot path (`Hnsw::{generate, search, improve_index, knn, ...}` and the
//! `Comparator::compare_vec` batches under them).  One declaration per prototype of the header,
//! in the header's order; `tests/test_rust_shim.py` parses this file and the header and fails when
//! a name, an argument count or an integer width drifts.
//!
//! Never compiled in the build image of this repository (no Rust toolchain there): written
//! against the header, checked by the parser test.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_float, c_int, c_void};

pub const PHNSW_EMPTY: u64 = u64::MAX; // VectorId::MAX / NodeId::MAX  (types.rs:8-13)

pub const PHNSW_OK: c_int = 0;
pub const PHNSW_E_INVALID: c_int = -1;
pub const PHNSW_E_NO_DEVICE: c_int = -2;
pub const PHNSW_E_HIP: c_int = -3;
pub const PHNSW_E_MISSING_NODE: c_int = -4;
pub const PHNSW_E_OVERFLOW: c_int = -5;
pub const PHNSW_E_NAN: c_int = -6;
pub const PHNSW_E_UNSUPPORTED: c_int = -7;
pub const PHNSW_E_NOMEM: c_int = -8;

pub const PHNSW_METRIC_COSINE_HALF: c_int = 0; // (1 - dot)/2   bigvec.rs:47-53
pub const PHNSW_METRIC_ONE_MINUS_DOT: c_int = 1; // 1 - dot       lib.rs:1985-1991
pub const PHNSW_METRIC_L2: c_int = 2; // sqrt(sum (a-b)^2)  lib.rs:2431-2437

/// SearchParameters  parameters.rs:3-18
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct phnsw_search_params {
    pub number_of_candidates: u64,
    pub upper_layer_candidate_count: u64,
    pub probe_depth: u64,
}

/// OptimizationParameters  parameters.rs:20-40
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct phnsw_optimization_params {
    pub promotion_threshold: c_float,
    pub neighborhood_threshold: c_float,
    pub recall_proportion: c_float,
    pub promotion_proportion: c_float,
    pub search: phnsw_search_params,
}

/// BuildParameters  parameters.rs:42-64 + seed / max_link_rounds / promote
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct phnsw_build_params {
    pub order: u64,
    pub zero_layer_neighborhood_size: u64,
    pub neighborhood_size: u64,
    pub optimization: phnsw_optimization_params,
    pub initial_partition_search: phnsw_search_params,
    pub seed: u64,
    pub max_link_rounds: u64,
    pub promote: u64,
}

#[repr(C)]
pub struct phnsw_store {
    _private: [u8; 0],
}
#[repr(C)]
pub struct phnsw_index {
    _private: [u8; 0],
}

/// ProgressMonitor::update / keep_alive  progress.rs:12-16
pub type phnsw_progress_cb =
    Option<unsafe extern "C" fn(user: *mut c_void, phase: *const c_char, done: u64, total: u64) -> c_int>;

// ---- sharded build (include/phnsw.h "sharded build"): the collective seam, its statistics, the phase engine

/// all_gather(ctx, send, recv, bytes, stream): device pointers + a hipStream_t to enqueue on, or host pointers
/// (host_buffers = 1, stream NULL, returns when recv is complete)
pub type phnsw_all_gather_fn =
    Option<unsafe extern "C" fn(ctx: *mut c_void, send: *const c_void, recv: *mut c_void, bytes: u64, stream: *mut c_void) -> c_int>;
/// all_reduce_sum(ctx, values, count): element-wise sum of host u64 over the ranks, in place
pub type phnsw_all_reduce_sum_fn = Option<unsafe extern "C" fn(ctx: *mut c_void, values: *mut u64, count: u32) -> c_int>;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct phnsw_comm {
    pub rank: u32,
    pub world: u32,
    pub host_buffers: u32,
    pub emulate: u32,
    pub ctx: *mut c_void,
    pub all_gather: phnsw_all_gather_fn,
    pub all_reduce_sum: phnsw_all_reduce_sum_fn,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct phnsw_sharded_stats {
    pub seconds_total: f64,
    pub seconds_sharded: f64,
    pub seconds_replicated: f64,
    pub seconds_comm: f64,
    pub seconds_others: f64,
    pub all_gather_bytes: u64,
    pub all_gather_calls: u64,
    pub all_reduce_calls: u64,
    pub phases: u64,
    pub phases_whole: u64,
    pub seconds_by_phase: [f64; 10],
}

/// the phases behind the sharded driver as callbacks (tests plug the CPU oracle in; hosts normally never touch it)
#[repr(C)]
pub struct phnsw_shard_engine {
    pub ctx: *mut c_void,
    pub id_bytes: u32,
    pub host_buffers: u32,
    pub alloc: Option<unsafe extern "C" fn(ctx: *mut c_void, bytes: u64) -> *mut c_void>,
    pub release: Option<unsafe extern "C" fn(ctx: *mut c_void, p: *mut c_void)>,
    pub copy2d: Option<unsafe extern "C" fn(ctx: *mut c_void, dst: *mut c_void, dpitch: u64, src: *const c_void, spitch: u64,
                                            width: u64, height: u64) -> c_int>,
    pub plan: Option<unsafe extern "C" fn(ctx: *mut c_void, vids: *const u64, n: u64, shuffled: *mut u64, layer_sizes: *mut u64,
                                          max_layers: u32, layer_count: *mut u32) -> c_int>,
    pub layer_begin: Option<unsafe extern "C" fn(ctx: *mut c_void, vids: *const u64, n: u64, w: u64, needs_phases: *mut c_int,
                                                 k: *mut u32) -> c_int>,
    pub layer_init_search: Option<unsafe extern "C" fn(ctx: *mut c_void, first: u64, count: u64, ids: *mut c_void, d: *mut c_float,
                                                       len: *mut c_void) -> c_int>,
    pub layer_seed: Option<unsafe extern "C" fn(ctx: *mut c_void, init_ids: *const c_void, init_d: *const c_float,
                                                init_len: *const c_void, first: u64, count: u64, rows: *mut c_void,
                                                rows_d: *mut c_float) -> c_int>,
    pub layer_finish: Option<unsafe extern "C" fn(ctx: *mut c_void, rows: *const c_void, rows_d: *const c_float) -> c_int>,
    pub layer_count: Option<unsafe extern "C" fn(ctx: *mut c_void) -> u32>,
    pub layer_nodes: Option<unsafe extern "C" fn(ctx: *mut c_void, layer_from_top: u32) -> u64>,
    pub link_search: Option<unsafe extern "C" fn(ctx: *mut c_void, layer_from_top: u32, sp: *const phnsw_search_params,
                                                 link_count: u64, first: u64, count: u64, ids: *mut c_void, d: *mut c_float,
                                                 len: *mut c_void) -> c_int>,
    pub link_apply: Option<unsafe extern "C" fn(ctx: *mut c_void, layer_from_top: u32, link_count: u64, ids: *const c_void,
                                                d: *const c_float, len: *const c_void, added: *mut u64) -> c_int>,
    pub recall_hits: Option<unsafe extern "C" fn(ctx: *mut c_void, layer_from_top: u32, op: *const phnsw_optimization_params,
                                                 first: u64, count: u64, hits: *mut u64, selection: *mut u64) -> c_int>,
    pub discover_hits: Option<unsafe extern "C" fn(ctx: *mut c_void, layer_from_top: u32, sp: *const phnsw_search_params,
                                                   first: u64, count: u64, hit: *mut c_void) -> c_int>,
    pub promote_from_hits: Option<unsafe extern "C" fn(ctx: *mut c_void, layer_from_top: u32, hit: *const c_void,
                                                       promoted: *mut c_int) -> c_int>,
    pub layer_cells: Option<unsafe extern "C" fn(ctx: *mut c_void, first: u64, count: u64, pos: *mut c_void) -> c_int>,
    pub layer_set_cells: Option<unsafe extern "C" fn(ctx: *mut c_void, pos: *const c_void) -> c_int>,
}

extern "C" {
    pub fn phnsw_default_search_params(sp: *mut phnsw_search_params);
    pub fn phnsw_default_build_params(bp: *mut phnsw_build_params);
    pub fn phnsw_last_error() -> *const c_char;
    pub fn phnsw_device_count() -> c_int;

    // ---- store
    pub fn phnsw_store_create(rows: *const c_float, n: u64, dim: u32, metric: c_int, device: c_int,
                              out: *mut *mut phnsw_store) -> c_int;
    pub fn phnsw_store_append(s: *mut phnsw_store, rows: *const c_float, count: u64, out_first_id: *mut u64) -> c_int;
    pub fn phnsw_store_create_device(rows_dev: *const c_float, n: u64, dim: u32, ld: u32, metric: c_int,
                                     device: c_int, out: *mut *mut phnsw_store) -> c_int;
    pub fn phnsw_store_create_synthetic(first: u64, n: u64, dim: u32, seed: u64, normalize: c_int, metric: c_int,
                                        device: c_int, out: *mut *mut phnsw_store) -> c_int;
    pub fn phnsw_store_create_clustered(first: u64, n: u64, dim: u32, seed: u64, n_clusters: u32, noise: c_float,
                                        metric: c_int, device: c_int, out: *mut *mut phnsw_store) -> c_int;
    pub fn phnsw_store_info(s: *const phnsw_store, n: *mut u64, dim: *mut u32, ld: *mut u32, metric: *mut c_int,
                            rows_dev: *mut *const c_float) -> c_int;
    pub fn phnsw_store_read(s: *const phnsw_store, first: u64, count: u64, out: *mut c_float) -> c_int;
    pub fn phnsw_store_destroy(s: *mut phnsw_store);
    pub fn phnsw_distance_batch(s: *const phnsw_store, query: *const c_float, query_id: u64, ids: *const u64,
                                k: u64, out: *mut c_float) -> c_int;

    // ---- index
    pub fn phnsw_index_from_layers(s: *mut phnsw_store, layer_count: u32, node_counts: *const u64,
                                   neighborhood_sizes: *const u64, nodes: *const *const u64,
                                   neighbors: *const *const u64, out: *mut *mut phnsw_index) -> c_int;
    pub fn phnsw_build(s: *mut phnsw_store, vids: *const u64, n: u64, bp: *const phnsw_build_params,
                       cb: phnsw_progress_cb, user: *mut c_void, out: *mut *mut phnsw_index) -> c_int;
    pub fn phnsw_generate_layer(ix: *mut phnsw_index, vids: *const u64, n: u64, neighborhood_size: u64,
                                bp: *const phnsw_build_params) -> c_int;
    pub fn phnsw_link_layer(ix: *mut phnsw_index, layer_from_top: u32, sp: *const phnsw_search_params,
                            link_count: u64, out_added: *mut u64) -> c_int;
    pub fn phnsw_improve_index(ix: *mut phnsw_index, bp: *const phnsw_build_params, last_recall: c_float,
                               cb: phnsw_progress_cb, user: *mut c_void, out_recall: *mut c_float) -> c_int;
    pub fn phnsw_improve_neighbors_upto(ix: *mut phnsw_index, upto: u32, bp: *const phnsw_build_params,
                                        last_recall: c_float, out_recall: *mut c_float) -> c_int;
    pub fn phnsw_extend_layer(ix: *mut phnsw_index, layer_from_top: u32, vids: *const u64, n: u64) -> c_int;
    pub fn phnsw_promote_at_layer(ix: *mut phnsw_index, layer_from_top: u32, bp: *const phnsw_build_params,
                                  out_promoted: *mut c_int) -> c_int;
    pub fn phnsw_discover_unreachable(ix: *mut phnsw_index, layer_from_top: u32, sp: *const phnsw_search_params,
                                      out_vecs: *mut u64, out_count: *mut u64) -> c_int;
    pub fn phnsw_stochastic_recall_at(ix: *mut phnsw_index, layer_from_top: u32,
                                      op: *const phnsw_optimization_params, out_recall: *mut c_float) -> c_int;
    pub fn phnsw_index_destroy(ix: *mut phnsw_index);
    pub fn phnsw_index_layer_count(ix: *const phnsw_index) -> u32;
    pub fn phnsw_index_layer_info(ix: *const phnsw_index, layer_from_top: u32, node_count: *mut u64,
                                  neighborhood_size: *mut u64) -> c_int;
    pub fn phnsw_index_layer_read(ix: *const phnsw_index, layer_from_top: u32, nodes: *mut u64,
                                  neighbors: *mut u64) -> c_int;

    // ---- search
    pub fn phnsw_search_batch(ix: *const phnsw_index, queries: *const c_float, nq: u64,
                              sp: *const phnsw_search_params, upto_layers: u32, exclude: *const u64,
                              out_ids: *mut u64, out_d: *mut c_float, out_len: *mut u64, out_stats: *mut u64) -> c_int;
    pub fn phnsw_search_batch_stored(ix: *const phnsw_index, qids: *const u64, nq: u64,
                                     sp: *const phnsw_search_params, upto_layers: u32, exclude: *const u64,
                                     out_ids: *mut u64, out_d: *mut c_float, out_len: *mut u64,
                                     out_stats: *mut u64) -> c_int;
    pub fn phnsw_search_batch_topk(ix: *const phnsw_index, queries: *const c_float, qids: *const u64, nq: u64,
                                   sp: *const phnsw_search_params, upto_layers: u32, exclude: *const u64, k: u64,
                                   out_ids: *mut u64, out_d: *mut c_float, out_len: *mut u64) -> c_int;
    pub fn phnsw_search_instrumented(ix: *const phnsw_index, queries: *const c_float, qids: *const u64, nq: u64,
                                     sp: *const phnsw_search_params, out_ids: *mut u64, out_d: *mut c_float,
                                     out_len: *mut u64, out_index_distance: *mut u64) -> c_int;
    /// a non-blocking hipStream_t seen to run beside `other_stream` (NULL = the default stream): lanes for two batches in flight
    pub fn phnsw_stream_create_beside(device: c_int, other_stream: *mut c_void, out_stream: *mut *mut c_void) -> c_int;
    pub fn phnsw_search_batch_device(ix: *const phnsw_index, queries_dev: *const c_float, ldq: u32,
                                     qids_dev: *const u32, nq: u64, sp: *const phnsw_search_params,
                                     upto_layers: u32, exclude_dev: *const u32, out_ids_dev: *mut u32,
                                     out_d_dev: *mut c_float, out_len_dev: *mut u32, out_stats_dev: *mut u32,
                                     status_dev: *mut u32, stream: *mut c_void) -> c_int;
    pub fn phnsw_index_counters(ix: *const phnsw_index, n_dist: *mut u64, n_hops: *mut u64) -> c_int;
    pub fn phnsw_last_search_kernel_ms(ix: *const phnsw_index, ms: *mut c_float) -> c_int;
    pub fn phnsw_last_search_dispatches(ix: *const phnsw_index, cap: u32, count: *mut u32, ms: *mut c_float,
                                        n_dist: *mut u64, n_hops: *mut u64, layer_lo: *mut u32,
                                        layer_hi: *mut u32) -> c_int;
    pub fn phnsw_last_search_table_evals(ix: *const phnsw_index, cap: u32, count: *mut u32, n_table: *mut u64) -> c_int;
    pub fn phnsw_dense_top_layers(ix: *const phnsw_index, number_of_candidates: u64, layers: *mut u32, nodes: *mut u64,
                                  matrix_cores: *mut u32) -> c_int;

    // ---- phase API (multi-GPU drivers)
    pub fn phnsw_index_create(s: *mut phnsw_store, bp: *const phnsw_build_params, out: *mut *mut phnsw_index) -> c_int;
    pub fn phnsw_build_plan(vids: *const u64, n: u64, bp: *const phnsw_build_params, shuffled: *mut u64,
                            layer_sizes: *mut u64, max_layers: u32, layer_count: *mut u32) -> c_int;
    pub fn phnsw_layer_begin(ix: *mut phnsw_index, vids: *const u64, n: u64, neighborhood_size: u64,
                             bp: *const phnsw_build_params, needs_phases: *mut c_int) -> c_int;
    pub fn phnsw_layer_begin_sharded(ix: *mut phnsw_index, vids: *const u64, n: u64, neighborhood_size: u64,
                                     bp: *const phnsw_build_params, needs_phases: *mut c_int,
                                     needs_cells: *mut c_int) -> c_int;
    pub fn phnsw_layer_cells_device(ix: *mut phnsw_index, first: u64, count: u64, out_pos: *mut u32) -> c_int;
    pub fn phnsw_layer_set_cells_device(ix: *mut phnsw_index, pos: *const u32) -> c_int;
    pub fn phnsw_layer_init_search_device(ix: *mut phnsw_index, bp: *const phnsw_build_params, first: u64,
                                          count: u64, out_ids: *mut u32, out_d: *mut c_float,
                                          out_len: *mut u32) -> c_int;
    pub fn phnsw_layer_seed_device(ix: *mut phnsw_index, bp: *const phnsw_build_params, init_ids: *const u32,
                                   init_d: *const c_float, init_len: *const u32, first: u64, count: u64,
                                   out_rows: *mut u32, out_rows_d: *mut c_float) -> c_int;
    pub fn phnsw_layer_finish_device(ix: *mut phnsw_index, rows: *const u32, rows_d: *const c_float) -> c_int;
    pub fn phnsw_link_search_device(ix: *mut phnsw_index, layer_from_top: u32, sp: *const phnsw_search_params,
                                    link_count: u64, first: u64, count: u64, out_ids: *mut u32,
                                    out_d: *mut c_float, out_len: *mut u32) -> c_int;
    pub fn phnsw_link_apply_device(ix: *mut phnsw_index, layer_from_top: u32, link_count: u64, ids: *const u32,
                                   d: *const c_float, len: *const u32, out_added: *mut u64) -> c_int;
    pub fn phnsw_discover_hits_device(ix: *mut phnsw_index, layer_from_top: u32, sp: *const phnsw_search_params,
                                      first: u64, count: u64, out_hit: *mut u32) -> c_int;
    pub fn phnsw_promote_at_layer_hits_device(ix: *mut phnsw_index, layer_from_top: u32,
                                              bp: *const phnsw_build_params, hit: *const u32,
                                              out_promoted: *mut c_int) -> c_int;
    pub fn phnsw_recall_hits(ix: *mut phnsw_index, layer_from_top: u32, op: *const phnsw_optimization_params,
                             first: u64, count: u64, out_hits: *mut u64, out_selection: *mut u64) -> c_int;

    // ---- sharded build (BASELINE config 4): one process per GPU, RCCL or a host-supplied transport
    pub fn phnsw_build_sharded(s: *mut phnsw_store, vids: *const u64, n: u64, bp: *const phnsw_build_params,
                               comm: *const phnsw_comm, cb: phnsw_progress_cb, user: *mut c_void,
                               out: *mut *mut phnsw_index, stats: *mut phnsw_sharded_stats) -> c_int;
    pub fn phnsw_improve_index_sharded(ix: *mut phnsw_index, bp: *const phnsw_build_params, last_recall: c_float,
                                       comm: *const phnsw_comm, out_recall: *mut c_float,
                                       stats: *mut phnsw_sharded_stats) -> c_int;
    pub fn phnsw_sharded_tuning(shard_min: u64, subchunks: u32, sub_min: u64) -> c_int;
    pub fn phnsw_comm_rccl_unique_id(out_id128: *mut u8) -> c_int;
    pub fn phnsw_comm_rccl_create(id128: *const u8, rank: u32, world: u32, device: c_int, out: *mut *mut phnsw_comm) -> c_int;
    pub fn phnsw_comm_destroy(c: *mut phnsw_comm);
    pub fn phnsw_comm_selftest(comm: *const phnsw_comm, bytes: u64) -> c_int;
    pub fn phnsw_comm_benchmark(comm: *const phnsw_comm, bytes: u64, iters: u32, host_us: *mut f64,
                                total_us: *mut f64) -> c_int;
    pub fn phnsw_build_sharded_engine(e: *const phnsw_shard_engine, vids: *const u64, n: u64,
                                      bp: *const phnsw_build_params, comm: *const phnsw_comm,
                                      stats: *mut phnsw_sharded_stats) -> c_int;

    // ---- product quantisation (pq.rs)
    pub fn phnsw_store_create_pq(full: *mut phnsw_store, m: u32, ksub: u32, seed: u64,
                                 out: *mut *mut phnsw_store) -> c_int;
    pub fn phnsw_store_create_pq_kmeans(full: *mut phnsw_store, m: u32, ksub: u32, seed: u64, kmeans_iters: u32,
                                        sample: u64, out: *mut *mut phnsw_store) -> c_int;
    pub fn phnsw_store_create_pq_shared(full: *mut phnsw_store, dsub: u32, n_centroids: u32, seed: u64,
                                        centroid_bp: *const phnsw_build_params,
                                        quantized_search: *const phnsw_search_params, centroid_metric: c_int,
                                        out: *mut *mut phnsw_store) -> c_int;
    pub fn phnsw_store_create_pq_sharded(full: *mut phnsw_store, m: u32, ksub: u32, seed: u64, kmeans_iters: u32,
                                         sample: u64, comm: *const phnsw_comm, out: *mut *mut phnsw_store) -> c_int;
    pub fn phnsw_store_create_pq_shared_sharded(full: *mut phnsw_store, dsub: u32, n_centroids: u32, seed: u64,
                                                centroid_bp: *const phnsw_build_params,
                                                quantized_search: *const phnsw_search_params, centroid_metric: c_int,
                                                comm: *const phnsw_comm, out: *mut *mut phnsw_store) -> c_int;
    pub fn phnsw_pq_shared_read(s: *const phnsw_store, codes: *mut u16, codebook: *mut c_float) -> c_int;
    pub fn phnsw_pq_shared_reconstruct_store(s: *const phnsw_store, out: *mut *mut phnsw_store) -> c_int;
    pub fn phnsw_pq_info(s: *const phnsw_store, m: *mut u32, ksub: *mut u32, dsub: *mut u32) -> c_int;
    pub fn phnsw_pq_set_table_mode(s: *mut phnsw_store, mode: c_int) -> c_int;
    pub fn phnsw_pq_set_table_f16(s: *mut phnsw_store, on: c_int) -> c_int;
    pub fn phnsw_pq_read(s: *const phnsw_store, codes: *mut u8, codebook: *mut c_float) -> c_int;
    pub fn phnsw_pq_quantize(s: *const phnsw_store, rows: *const c_float, n: u64, out_codes: *mut u8) -> c_int;
    pub fn phnsw_pq_reconstruct(s: *const phnsw_store, codes: *const u8, n: u64, out_rows: *mut c_float) -> c_int;
    pub fn phnsw_pq_search_batch(ix: *const phnsw_index, full: *const phnsw_store, queries: *const c_float,
                                 nq: u64, sp: *const phnsw_search_params, quantize_query: c_int,
                                 out_ids: *mut u64, out_d: *mut c_float, out_len: *mut u64,
                                 out_stats: *mut u64) -> c_int;
    pub fn phnsw_pq_search_batch_device(ix: *const phnsw_index, full: *const phnsw_store,
                                        queries_dev: *const c_float, ldq: u32, nq: u64,
                                        sp: *const phnsw_search_params, out_ids_dev: *mut u32,
                                        out_d_dev: *mut c_float, out_len_dev: *mut u32, out_stats_dev: *mut u32,
                                        status_dev: *mut u32, stream: *mut c_void) -> c_int;

    // ---- on-disk interchange (serialize.rs:33-209)
    pub fn phnsw_index_serialize(ix: *const phnsw_index, path: *const c_char) -> c_int;
    pub fn phnsw_index_deserialize(s: *mut phnsw_store, path: *const c_char, out: *mut *mut phnsw_index) -> c_int;
    pub fn phnsw_index_build_params(ix: *const phnsw_index, bp: *mut phnsw_build_params) -> c_int;

    // ---- bulk neighbour queries, ground truth
    pub fn phnsw_knn(ix: *const phnsw_index, k: u64, probe_depth: u64, out_ids: *mut u64, out_d: *mut c_float,
                     out_len: *mut u64) -> c_int;
    pub fn phnsw_bruteforce_topk(s: *const phnsw_store, queries: *const c_float, nq: u64, k: u32,
                                 out_ids: *mut u64, out_d: *mut c_float) -> c_int;
    pub fn phnsw_bruteforce_topk_device(s: *const phnsw_store, queries_dev: *const c_float, ldq: u32, nq: u64,
                                        k: u32, out_ids_dev: *mut u32, out_d_dev: *mut c_float,
                                        stream: *mut c_void) -> c_int;
    pub fn phnsw_bruteforce_last_gemm_ms() -> c_float;
    pub fn phnsw_threshold_nn(ix: *const phnsw_index, threshold: c_float, probe_depth: u64,
                              initial_search_depth: u64, max_out: u64, out_ids: *mut u64, out_d: *mut c_float,
                              out_len: *mut u64) -> c_int;
}
