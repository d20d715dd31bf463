/*
 * phnsw.h -- C ABI of the MI355X-native HNSW build + search engine.
 *
 * Drop-in boundary for ONE hot path of terminusdb-labs/parallel-hnsw (Rust): the per-hop
 * candidate distance batch and the greedy layer search / layer construction loops around
 * it.  The reference's seam is the generic, one-pair-per-call trait
 *     Comparator::{lookup, compare_raw, compare_vec}            src/lib.rs:53-74
 * which cannot feed a GPU, so this ABI sits one level up, under the method surface of
 *     Hnsw<C>::{generate, search, search_upto, improve_index, improve_neighbors,
 *               stochastic_recall, knn, threshold_nn}           src/lib.rs:585-1686
 * and batches beneath it.  A Rust shim implements `Comparator` for a
 * store handle and wraps `Hnsw` around an index handle.
 *
 * Conventions
 *  - every entry point returns 0 on success, a negative PHNSW_E_* otherwise, and never
 *    unwinds; phnsw_last_error() gives the message of the calling thread's last failure.
 *    (The reference panics: unwrap lib.rs:261, NaN types.rs:86, empty input lib.rs:683,837.)
 *  - ids are u64 at the boundary like the reference's usize VectorId/NodeId
 *    (src/types.rs:3-13); PHNSW_EMPTY == !0 is the empty-slot sentinel.
 *  - host pointers unless the name ends in _device; outputs are caller allocated; the
 *    library never frees caller memory and copies what it needs before returning.
 *  - search entry points are thread safe; build / improve entry points need exclusive
 *    access to their index (lib.rs: &self vs &mut self).
 *  - there is NO CPU fallback: every call fails with PHNSW_E_NO_DEVICE without a gfx950 GPU.
 */
#ifndef PHNSW_H
#define PHNSW_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHNSW_EMPTY UINT64_MAX /* VectorId::MAX / NodeId::MAX  src/types.rs:8-13 */

enum {
  PHNSW_OK = 0,
  PHNSW_E_INVALID = -1,   /* bad argument (the reference would panic/assert) */
  PHNSW_E_NO_DEVICE = -2, /* no HIP device / wrong architecture */
  PHNSW_E_HIP = -3,       /* HIP runtime error, see phnsw_last_error */
  PHNSW_E_MISSING_NODE = -4, /* candidate vector absent from a lower layer (lib.rs:261 unwrap) */
  PHNSW_E_OVERFLOW = -5,  /* frontier workspace exhausted even after growth */
  PHNSW_E_NAN = -6,       /* NaN in input vectors (types.rs:86 would panic later) */
  PHNSW_E_UNSUPPORTED = -7,
  PHNSW_E_NOMEM = -8      /* host allocation failed (a C++ exception never crosses this ABI) */
};

/* the three Comparator::compare_raw implementations in the reference tree */
enum {
  PHNSW_METRIC_COSINE_HALF = 0,   /* (1 - dot)/2        BigComparator, src/bigvec.rs:47-53 */
  PHNSW_METRIC_ONE_MINUS_DOT = 1, /* 1 - dot            src/lib.rs:1985-1991 */
  PHNSW_METRIC_L2 = 2             /* sqrt(sum (a-b)^2)  src/lib.rs:2431-2437, src/pq.rs:499-505 */
};

/* SearchParameters  src/parameters.rs:3-18 (field for field) */
typedef struct {
  uint64_t number_of_candidates;
  uint64_t upper_layer_candidate_count;
  uint64_t probe_depth;
} phnsw_search_params;

/* OptimizationParameters  src/parameters.rs:20-40 */
typedef struct {
  float promotion_threshold;
  float neighborhood_threshold;
  float recall_proportion;
  float promotion_proportion;
  phnsw_search_params search;
} phnsw_optimization_params;

/* BuildParameters  src/parameters.rs:42-64, plus the two knobs that replace the
 * reference's non-determinism / unbounded loops */
typedef struct {
  uint64_t order;
  uint64_t zero_layer_neighborhood_size;
  uint64_t neighborhood_size;
  phnsw_optimization_params optimization;
  phnsw_search_params initial_partition_search;
  uint64_t seed;            /* replaces thread_rng() of lib.rs:832 */
  uint64_t max_link_rounds; /* 0 = loop until improvement < neighborhood_threshold (lib.rs:1527) */
  uint64_t promote;         /* 1 (default) = promote_at_layer as the reference (lib.rs:1575-1589); 0 = never */
} phnsw_build_params;

void phnsw_default_search_params(phnsw_search_params *sp);
void phnsw_default_build_params(phnsw_build_params *bp);

typedef struct phnsw_store phnsw_store; /* vector store + metric == a Comparator */
typedef struct phnsw_index phnsw_index; /* Hnsw<C> */

/* ProgressMonitor::update / keep_alive  src/progress.rs:12-16: called from the calling
 * thread between build phases; return non-zero to request an interrupt. */
typedef int (*phnsw_progress_cb)(void *user, const char *phase, uint64_t done, uint64_t total);

const char *phnsw_last_error(void);
int phnsw_device_count(void);

/* ---- store: replaces BigComparator{data: Arc<Vec<Vec<f32>>>}  src/bigvec.rs:38-57 ----
 * rows are copied into one flat HBM array [n][ld], ld = dim rounded up to 4 floats. */
int phnsw_store_create(const float *rows, uint64_t n, uint32_t dim, int metric, int device,
                       phnsw_store **out);
/* append rows to a store that owns its array (ids continue at the old n; *out_first_id = first
 * new VectorId).  The reference grows the Vec behind its comparator the same way before it
 * indexes new ids (src/bigvec.rs:38-44).  Not concurrent with searches/builds on this store. */
int phnsw_store_append(phnsw_store *s, const float *rows, uint64_t count, uint64_t *out_first_id);
/* adopt an existing device array (e.g. a torch tensor); caller keeps it alive */
int phnsw_store_create_device(const float *rows_dev, uint64_t n, uint32_t dim, uint32_t ld,
                              int metric, int device, phnsw_store **out);
/* synthetic rows with the distribution of random_normed_vec (src/bigvec.rs:59-65),
 * generated on the device: component j of vector i keyed (seed + first + i, j) */
int phnsw_store_create_synthetic(uint64_t first, uint64_t n, uint32_t dim, uint64_t seed,
                                 int normalize, int metric, int device, phnsw_store **out);
/* clustered synthetic rows (n_clusters unit centres + uniform noise of norm ~noise,
 * normalised): the benchmark dataset on which recall@10 >= 0.95 is reachable */
int phnsw_store_create_clustered(uint64_t first, uint64_t n, uint32_t dim, uint64_t seed,
                                 uint32_t n_clusters, float noise, int metric, int device,
                                 phnsw_store **out);
int phnsw_store_info(const phnsw_store *s, uint64_t *n, uint32_t *dim, uint32_t *ld, int *metric,
                     const float **rows_dev);
/* copy rows [first, first+count) back to the host, dim floats each */
int phnsw_store_read(const phnsw_store *s, uint64_t first, uint64_t count, float *out);
void phnsw_store_destroy(phnsw_store *s);

/* Comparator::compare_vec batched  src/lib.rs:69-73: out[i] = d(query, Stored(ids[i])).
 * query == NULL means AbstractVector::Stored(query_id). */
int phnsw_distance_batch(const phnsw_store *s, const float *query, uint64_t query_id,
                         const uint64_t *ids, uint64_t k, float *out);

/* ---- index ---- */
/* adopt a layer stack built elsewhere (e.g. by the Rust crate); layers top first like
 * Hnsw.layers (lib.rs:587); nodes sorted ascending; neighbor rows padded with PHNSW_EMPTY */
int phnsw_index_from_layers(phnsw_store *s, uint32_t layer_count, const uint64_t *node_counts,
                            const uint64_t *neighborhood_sizes, const uint64_t *const *nodes,
                            const uint64_t *const *neighbors, phnsw_index **out);
/* Hnsw::generate  src/lib.rs:825-893 */
int phnsw_build(phnsw_store *s, const uint64_t *vids, uint64_t n, const phnsw_build_params *bp,
                phnsw_progress_cb cb, void *user, phnsw_index **out);
/* Hnsw::generate_layer appended below the current stack  src/lib.rs:675-823 */
int phnsw_generate_layer(phnsw_index *ix, const uint64_t *vids, uint64_t n,
                         uint64_t neighborhood_size, const phnsw_build_params *bp);
/* link_layer_to_better_neighbors  src/lib.rs:1070-1154 ; *out_added = new edges */
int phnsw_link_layer(phnsw_index *ix, uint32_t layer_from_top, const phnsw_search_params *sp,
                     uint64_t link_count, uint64_t *out_added);
/* Hnsw::improve_index(bp, last_recall, progress)  src/lib.rs:1664-1686, promotion included
 * (bp->promote).  last_recall: NaN = None (the recall is estimated first, lib.rs:1671); a number is
 * taken as the current recall, exactly like Some(r) -- it only saves that first estimate, every
 * improve_index_at call below is made with None (lib.rs:1679). */
int phnsw_improve_index(phnsw_index *ix, const phnsw_build_params *bp, float last_recall,
                        phnsw_progress_cb cb, void *user, float *out_recall);
/* Hnsw::improve_neighbors_upto  src/lib.rs:1515-1544 ; last_recall NaN = None */
int phnsw_improve_neighbors_upto(phnsw_index *ix, uint32_t upto, const phnsw_build_params *bp,
                                 float last_recall, float *out_recall);
/* Hnsw::extend_layer  src/lib.rs:1039-1068: vids join the layer with empty neighbourhoods, NodeIds
 * are renumbered (generate_node_maps :1767-1812); inserting a vector twice is PHNSW_E_INVALID */
int phnsw_extend_layer(phnsw_index *ix, uint32_t layer_from_top, const uint64_t *vids, uint64_t n);
/* Hnsw::promote_at_layer  src/lib.rs:1273-1427 ; *out_promoted = the bool it returns */
int phnsw_promote_at_layer(phnsw_index *ix, uint32_t layer_from_top, const phnsw_build_params *bp,
                           int *out_promoted);
/* Hnsw::discover_unreachable_vectors  src/lib.rs:1002-1037 ; out sized node_count of the layer */
int phnsw_discover_unreachable(phnsw_index *ix, uint32_t layer_from_top, const phnsw_search_params *sp,
                               uint64_t *out_vecs, uint64_t *out_count);
/* Hnsw::stochastic_recall_at  src/lib.rs:1463-1499 */
int phnsw_stochastic_recall_at(phnsw_index *ix, uint32_t layer_from_top,
                               const phnsw_optimization_params *op, float *out_recall);
void phnsw_index_destroy(phnsw_index *ix);

uint32_t phnsw_index_layer_count(const phnsw_index *ix);
/* Layer{neighborhood_size, nodes, neighbors}  src/lib.rs:85-91 */
int phnsw_index_layer_info(const phnsw_index *ix, uint32_t layer_from_top, uint64_t *node_count,
                           uint64_t *neighborhood_size);
int phnsw_index_layer_read(const phnsw_index *ix, uint32_t layer_from_top, uint64_t *nodes,
                           uint64_t *neighbors);

/* Hnsw::search for a batch  src/lib.rs:663-665 -> src/search.rs:84-140.
 * queries: nq rows of dim floats (AbstractVector::Unstored).  exclude: NULL or nq ids
 * (PHNSW_EMPTY = None).  upto_layers: 0 = all (Hnsw::search_upto lib.rs:654-661).
 * Outputs: [nq][number_of_candidates] ids / distances sorted by (distance, id), padded with
 * PHNSW_EMPTY / f32::MAX; out_len[q] = valid entries; out_stats (nullable) [nq][2] =
 * {distance evaluations, hops}. */
int phnsw_search_batch(const phnsw_index *ix, const float *queries, uint64_t nq,
                       const phnsw_search_params *sp, uint32_t upto_layers,
                       const uint64_t *exclude, uint64_t *out_ids, float *out_d,
                       uint64_t *out_len, uint64_t *out_stats);
/* same for AbstractVector::Stored(qids[q]) */
int phnsw_search_batch_stored(const phnsw_index *ix, const uint64_t *qids, uint64_t nq,
                              const phnsw_search_params *sp, uint32_t upto_layers,
                              const uint64_t *exclude, uint64_t *out_ids, float *out_d,
                              uint64_t *out_len, uint64_t *out_stats);
/* the same keeping only the best k <= number_of_candidates results per query, out_* [nq][k]: the reference
 * returns the whole queue and its callers truncate (lib.rs:1118 takes neighborhood_size, a k-NN service takes k);
 * here the truncation happens on the device, before the transfer.  Exactly one of queries / qids is non-NULL. */
int phnsw_search_batch_topk(const phnsw_index *ix, const float *queries, const uint64_t *qids, uint64_t nq,
                            const phnsw_search_params *sp, uint32_t upto_layers, const uint64_t *exclude, uint64_t k,
                            uint64_t *out_ids, float *out_d, uint64_t *out_len);
/* Hnsw::search_instrumented  src/lib.rs:667-673: the results of phnsw_search_batch plus, per query, the second
 * value of search_layers_instrumented (src/search.rs:93-140): the index_sum of the last hop of the bottom layer's
 * closest_nodes that changed the best candidate (lib.rs:211-231); UINT64_MAX = usize::MAX.  f32 stores. */
int phnsw_search_instrumented(const phnsw_index *ix, const float *queries, const uint64_t *qids, uint64_t nq,
                              const phnsw_search_params *sp, uint64_t *out_ids, float *out_d, uint64_t *out_len,
                              uint64_t *out_index_distance);
/* zero-copy form: everything already resident in HBM, u32 ids (0xFFFFFFFF = empty),
 * queries [nq][ldq] with ldq a multiple of 4 and zero padding, launched on `stream`
 * (a hipStream_t, NULL = default) without synchronising.  out_stats_dev [nq][2] u32.
 * Returns after enqueueing; per-query status lands in status_dev[nq] (0 = ok). */
int phnsw_search_batch_device(const phnsw_index *ix, const float *queries_dev, uint32_t ldq,
                              const uint32_t *qids_dev, uint64_t nq, const phnsw_search_params *sp,
                              uint32_t upto_layers, const uint32_t *exclude_dev,
                              uint32_t *out_ids_dev, float *out_d_dev, uint32_t *out_len_dev,
                              uint32_t *out_stats_dev, uint32_t *status_dev, void *stream);
/* Throughput callers keep TWO batches in flight: phnsw_search_batch_device calls issued alternately on two streams
 * overlap (an index holds two search workspaces) -- if the two streams sit on different hardware queues.  HIP maps a
 * process's streams onto a few of them (GPU_MAX_HW_QUEUES, 4 by default) and two streams that share one run in issue
 * order.  This makes a non-blocking stream that was SEEN to run beside `other_stream` (a hipStream_t; NULL = the
 * default stream): a kernel spinning for a millisecond on `other_stream`, an empty one on the candidate, up to six
 * candidates.  *out_stream is a hipStream_t the caller destroys with hipStreamDestroy. */
int phnsw_stream_create_beside(int device, void *other_stream, void **out_stream);
/* timing of the last phnsw_search_batch_device launch on this index measured with HIP
 * events on its stream: kernel milliseconds */
/* running totals of distance evaluations and hops over every search launched on the index since
 * its creation (build rounds included; the reference's SearchStats per query, search.rs:93-99,
 * summed): the build's algorithmic bytes are n_dist * row bytes + n_hops * neighbour-row bytes */
int phnsw_index_counters(const phnsw_index *ix, uint64_t *n_dist, uint64_t *n_hops);
int phnsw_last_search_kernel_ms(const phnsw_index *ix, float *ms);
/* the same descent dispatch by dispatch (measurement only): entry 0 = the dense-top-layer tile pass
 * (csrc/tiny.hip; layers 0..0), then one entry per launch of the search kernel: layers
 * [layer_lo, layer_hi), milliseconds, distance evaluations and hops.  *count = entries available;
 * at most cap are written; any output array may be NULL.  A query list longer than the dense table holds
 * (4 GiB by default: ~145 000 queries at 1M x 768) runs in chunks; times and counters then describe the LAST chunk. */
int phnsw_last_search_dispatches(const phnsw_index *ix, uint32_t cap, uint32_t *count, float *ms,
                                 uint64_t *n_dist, uint64_t *n_hops, uint32_t *layer_lo,
                                 uint32_t *layer_hi);
/* per entry of phnsw_last_search_dispatches: how many of the launch's distance evaluations were served by the
 * dense tables (measurement only); n_dist - n_table are gathered rows, the bytes an HBM roofline counts */
int phnsw_last_search_table_evals(const phnsw_index *ix, uint32_t cap, uint32_t *count, uint64_t *n_table);
/* how a search with this number_of_candidates treats the leading layers (measurement only): *layers =
 * how many of them are walked through the dense distance table (csrc/tiny.hip), *nodes = nodes of
 * the largest of them (the table's width), *matrix_cores = 1 when the table is built by the MFMA
 * kernel (dot-product metric, rows of 256 / 768 / 1536 floats), 0 for the vector-unit kernel. */
int phnsw_dense_top_layers(const phnsw_index *ix, uint64_t number_of_candidates, uint32_t *layers,
                           uint64_t *nodes, uint32_t *matrix_cores);

/* ---- phase API: the per-round pieces of phnsw_generate_layer / phnsw_link_layer /
 * phnsw_stochastic_recall_at over a NODE RANGE, device buffers, u32 ids (0xFFFFFFFF empty).
 * A multi-GPU driver (parallel_hnsw_amd/sharded.py) gives every rank a replica of store and
 * graph, lets rank r run the searches of its node range, all-gathers the per-node results
 * over RCCL and lets every rank apply them -- replicas stay bit-identical.  The
 * single-GPU entry points above are these phases over the whole range. ---- */
int phnsw_index_create(phnsw_store *s, const phnsw_build_params *bp, phnsw_index **out); /* no layers yet */
/* the id shuffle of phnsw_build (lib.rs:832-833) and its layer sizes, top first
 * (calculate_partitions lib.rs:1883-1899) */
int phnsw_build_plan(const uint64_t *vids, uint64_t n, const phnsw_build_params *bp,
                     uint64_t *shuffled, uint64_t *layer_sizes, uint32_t max_layers,
                     uint32_t *layer_count);
/* generate_layer: begin (sort, id maps; *needs_phases = 0 when the layer is already complete:
 * first layer of a stack) -> init_search(range) -> seed(range) -> finish */
int phnsw_layer_begin(phnsw_index *ix, const uint64_t *vids, uint64_t n, uint64_t neighborhood_size,
                      const phnsw_build_params *bp, int *needs_phases);
/* the same for a sharded driver: the cells of the new layer's locality schedule (a GEMM of its vectors against the
 * store's anchors -- a scheduling hint, never part of a result, but the one costly step of begin) are left out when
 * *needs_cells comes back 1: every rank computes a node range (cells_device), the ranges are all-gathered, every
 * rank installs the whole array (set_cells_device) */
int phnsw_layer_begin_sharded(phnsw_index *ix, const uint64_t *vids, uint64_t n, uint64_t neighborhood_size,
                              const phnsw_build_params *bp, int *needs_phases, int *needs_cells);
int phnsw_layer_cells_device(phnsw_index *ix, uint64_t first, uint64_t count, uint32_t *out_pos);
int phnsw_layer_set_cells_device(phnsw_index *ix, const uint32_t *pos);
/* out_* : [count][K] NodeIds of the new layer / distances, [count] lengths;
 * K = initial_partition_search.number_of_candidates  (search.rs:32-71) */
int phnsw_layer_init_search_device(phnsw_index *ix, const phnsw_build_params *bp, uint64_t first,
                                   uint64_t count, uint32_t *out_ids, float *out_d, uint32_t *out_len);
/* init_* : the FULL [n][K] lists; out_rows : [count][W]  (lib.rs:711-787) */
int phnsw_layer_seed_device(phnsw_index *ix, const phnsw_build_params *bp, const uint32_t *init_ids,
                            const float *init_d, const uint32_t *init_len, uint64_t first,
                            uint64_t count, uint32_t *out_rows, float *out_rows_d);
/* rows : the FULL [n][W] seeded rows; bidirectional pass + push  (lib.rs:789-822) */
int phnsw_layer_finish_device(phnsw_index *ix, const uint32_t *rows, const float *rows_d);
/* link round: searches of nodes [first, first+count) -> best link_count results as
 * VectorIds [count][link_count]  (lib.rs:1107-1117) */
int phnsw_link_search_device(phnsw_index *ix, uint32_t layer_from_top, const phnsw_search_params *sp,
                             uint64_t link_count, uint64_t first, uint64_t count, uint32_t *out_ids,
                             float *out_d, uint32_t *out_len);
/* ... and the row updates from the FULL [n][link_count] results  (lib.rs:1118-1147) */
int phnsw_link_apply_device(phnsw_index *ix, uint32_t layer_from_top, uint64_t link_count,
                            const uint32_t *ids, const float *d, const uint32_t *len,
                            uint64_t *out_added);
/* promote_at_layer in two phases: the self-hit flags (match_within_epsilon) of nodes
 * [first, first+count) -> all-gather -> the promotion from the FULL [node_count] flags */
int phnsw_discover_hits_device(phnsw_index *ix, uint32_t layer_from_top, const phnsw_search_params *sp,
                               uint64_t first, uint64_t count, uint32_t *out_hit);
int phnsw_promote_at_layer_hits_device(phnsw_index *ix, uint32_t layer_from_top,
                                       const phnsw_build_params *bp, const uint32_t *hit,
                                       int *out_promoted);
/* self-hits among sample[first, first+count) of stochastic_recall_at; *out_selection = sample size */
int phnsw_recall_hits(phnsw_index *ix, uint32_t layer_from_top, const phnsw_optimization_params *op,
                      uint64_t first, uint64_t count, uint64_t *out_hits, uint64_t *out_selection);

/* ---- sharded build: Hnsw::generate / improve_index with every per-node phase split over the GPUs
 * of one node (BASELINE config 4).  One process per GPU; every rank holds a replica of
 * store and graph; within a round nodes are independent (the searches of lib.rs:1107-1117 read a
 * snapshot), so rank r runs the phase for the node range [r*chunk, (r+1)*chunk), the per-node results
 * (u32 ids + f32 distances) are all-gathered, and every rank applies ALL of them (K5): replicas stay
 * bit-identical and the graph equals phnsw_build's.  Control flow = lib.rs:825-893, 1515-1686.
 *
 * phnsw_comm is the collective seam: a host supplies its own transport (MPI, a torch.distributed
 * group, ...) as two callbacks, or takes the built-in RCCL one (phnsw_comm_rccl_create: ncclAllGather
 * over xGMI on a stream of the library's, no torch).
 *   all_gather: every rank contributes `bytes` at `send`; `recv` receives world*bytes in rank order.
 *     host_buffers == 0: device pointers; the call ENQUEUES on `stream` (a hipStream_t) and may return
 *     before the transfer is done -- the library overlaps it with the next piece's searches and
 *     synchronises the stream itself.  host_buffers == 1: host pointers (the library stages through
 *     pinned memory), stream is NULL, the call returns when recv is complete.
 *   all_reduce_sum: element-wise sum of `count` host u64 over the ranks, in place.
 *   emulate != 0 with all_gather == NULL: ONE process plays all `world` ranks in turn on its GPU (every
 *     range is computed here, in rank order, with the same split, packing and reassembly): what the
 *     one-GPU tests and the scaling model of bench.py use; phnsw_sharded_stats separates rank `rank`'s
 *     time from the others'. */
typedef int (*phnsw_all_gather_fn)(void *ctx, const void *send, void *recv, uint64_t bytes, void *stream);
typedef int (*phnsw_all_reduce_sum_fn)(void *ctx, uint64_t *values, uint32_t count);
typedef struct phnsw_comm {
  uint32_t rank, world;
  uint32_t host_buffers;
  uint32_t emulate;
  void *ctx;
  phnsw_all_gather_fn all_gather;
  phnsw_all_reduce_sum_fn all_reduce_sum;
} phnsw_comm;

/* where a sharded build spent its time (seconds on the calling rank) and what it moved */
typedef struct phnsw_sharded_stats {
  double seconds_total;
  double seconds_sharded;    /* this rank's share of the per-node phases (searches, seeding) */
  double seconds_replicated; /* phases every rank repeats (row merges K5, layer bookkeeping, promotion) */
  double seconds_comm;       /* host time inside collectives, their waits and the reassembly copies */
  double seconds_others;     /* emulate: the other ranks' shares, computed here in turn */
  uint64_t all_gather_bytes; /* received, summed over calls */
  uint64_t all_gather_calls;
  uint64_t all_reduce_calls;
  uint64_t phases;           /* sharded phases run */
  uint64_t phases_whole;     /* work lists too short to split (every rank ran them whole, no collective) */
  /* this rank's seconds by phase: the sharded ones 0 layer_init_search, 1 layer_seed, 2 link_search, 3 recall_hits,
   * 4 discover_hits; the replicated ones 5 plan, 6 layer_begin, 7 layer_finish, 8 link_apply, 9 promote_from_hits */
  double seconds_by_phase[10];
} phnsw_sharded_stats;

/* Hnsw::generate (lib.rs:825-893) over the ranks of `comm`; comm == NULL or world == 1 is phnsw_build.
 * Every rank must call it with the same store contents, vids and bp.  stats nullable. */
int phnsw_build_sharded(phnsw_store *s, const uint64_t *vids, uint64_t n, const phnsw_build_params *bp,
                        const phnsw_comm *comm, phnsw_progress_cb cb, void *user, phnsw_index **out,
                        phnsw_sharded_stats *stats);
/* Hnsw::improve_index (lib.rs:1664-1686) on an existing, replicated index */
int phnsw_improve_index_sharded(phnsw_index *ix, const phnsw_build_params *bp, float last_recall,
                                const phnsw_comm *comm, float *out_recall, phnsw_sharded_stats *stats);
/* work lists shorter than shard_min run whole on every rank (default 4096); a rank's share is cut into
 * `subchunks` pieces of at least sub_min items whose all-gathers overlap the next piece (defaults 4, 65536).
 * 0 keeps a value.  Process-wide; for tests and tuning. */
int phnsw_sharded_tuning(uint64_t shard_min, uint32_t subchunks, uint64_t sub_min);

/* the built-in transport: RCCL (librccl is loaded on first use; PHNSW_RCCL_LIB overrides its path).
 * Rank 0 makes the 128-byte id, the host hands it to the other ranks by any means, every rank creates
 * its communicator on its own device.  The returned phnsw_comm has host_buffers == 0. */
int phnsw_comm_rccl_unique_id(uint8_t *out_id128);
int phnsw_comm_rccl_create(const uint8_t *id128, uint32_t rank, uint32_t world, int device, phnsw_comm **out);
void phnsw_comm_destroy(phnsw_comm *c);
/* checks a communicator end to end before a build is trusted to it: every rank contributes a known pattern of
 * `bytes` bytes, verifies all world blocks of the all-gather, then the all-reduce.  Collective: every rank calls it. */
int phnsw_comm_selftest(const phnsw_comm *comm, uint64_t bytes);
/* what a collective costs: `iters` all-gathers of `bytes` per rank back to back -- *host_us = host time per call to
 * enqueue it, *total_us = wall time per call until the last has landed.  Collective; device-buffer transports. */
int phnsw_comm_benchmark(const phnsw_comm *comm, uint64_t bytes, uint32_t iters, double *host_us, double *total_us);

/* The phase engine behind the sharded driver.  phnsw_build_sharded runs the driver over libphnsw's own
 * GPU phases (the phase API above); this entry runs the SAME driver over an engine given as callbacks,
 * which is how the CPU tests drive it under gloo with the oracle's phases (tests/test_sharded_gloo.py).
 * ids / lengths / hit flags are id_bytes wide (4 or 8), distances f32; host_buffers: alloc returns host
 * memory (then comm->host_buffers must be 1 too). */
typedef struct phnsw_shard_engine {
  void *ctx;
  uint32_t id_bytes;
  uint32_t host_buffers;
  void *(*alloc)(void *ctx, uint64_t bytes);
  void (*release)(void *ctx, void *p);
  int (*copy2d)(void *ctx, void *dst, uint64_t dpitch, const void *src, uint64_t spitch, uint64_t width,
                uint64_t height);
  int (*plan)(void *ctx, const uint64_t *vids, uint64_t n, uint64_t *shuffled, uint64_t *layer_sizes,
              uint32_t max_layers, uint32_t *layer_count);
  int (*layer_begin)(void *ctx, const uint64_t *vids, uint64_t n, uint64_t W, int *needs_phases, uint32_t *K);
  int (*layer_init_search)(void *ctx, uint64_t first, uint64_t count, void *ids, float *d, void *len);
  int (*layer_seed)(void *ctx, const void *init_ids, const float *init_d, const void *init_len, uint64_t first,
                    uint64_t count, void *rows, float *rows_d);
  int (*layer_finish)(void *ctx, const void *rows, const float *rows_d);
  uint32_t (*layer_count)(void *ctx);
  uint64_t (*layer_nodes)(void *ctx, uint32_t layer_from_top);
  int (*link_search)(void *ctx, uint32_t layer_from_top, const phnsw_search_params *sp, uint64_t link_count,
                     uint64_t first, uint64_t count, void *ids, float *d, void *len);
  int (*link_apply)(void *ctx, uint32_t layer_from_top, uint64_t link_count, const void *ids, const float *d,
                    const void *len, uint64_t *added);
  int (*recall_hits)(void *ctx, uint32_t layer_from_top, const phnsw_optimization_params *op, uint64_t first,
                     uint64_t count, uint64_t *hits, uint64_t *selection);
  int (*discover_hits)(void *ctx, uint32_t layer_from_top, const phnsw_search_params *sp, uint64_t first,
                       uint64_t count, void *hit);
  int (*promote_from_hits)(void *ctx, uint32_t layer_from_top, const void *hit, int *promoted);
  /* optional (NULL: layer_begin does it all): when layer_begin sets *needs_phases to 3 instead of 1 the driver
   * shards layer_cells (u32 per node) like any phase and hands the whole array to layer_set_cells */
  int (*layer_cells)(void *ctx, uint64_t first, uint64_t count, void *pos);
  int (*layer_set_cells)(void *ctx, const void *pos);
} phnsw_shard_engine;
int phnsw_build_sharded_engine(const phnsw_shard_engine *e, const uint64_t *vids, uint64_t n,
                               const phnsw_build_params *bp, const phnsw_comm *comm, phnsw_sharded_stats *stats);

/* ---- product quantisation (reference src/pq.rs; BASELINE config 5) ----
 * A PQ store holds u8 code rows [n][m] over per-sub-space codebooks [m][ksub][dim/m]
 * (random_centroids pq.rs:261-285 per sub-space; Quantizer::quantize pq.rs:61-71 as the exact
 * nearest centroid).  It is a phnsw_store: phnsw_build / phnsw_search_batch / phnsw_link_layer
 * ... run on it with quantised distances (a per-query lookup table in LDS; a Stored query is its
 * reconstruction, so code-vs-code distances are symmetric).  m % 4 == 0, dim % m == 0,
 * ksub <= 256, m*ksub*4 bytes must fit the LDS. */
int phnsw_store_create_pq(phnsw_store *full, uint32_t m, uint32_t ksub, uint64_t seed, phnsw_store **out);
/* the same with k-means codebooks (the reference's own k-means is dead code,
 * pq.rs:215-259): kmeans_iters Lloyd iterations from the random_centroids start, trained on the first
 * min(n, sample) vectors of the seeded shuffle (sample 0 = all); kmeans_iters 0 == phnsw_store_create_pq */
int phnsw_store_create_pq_kmeans(phnsw_store *full, uint32_t m, uint32_t ksub, uint64_t seed,
                                 uint32_t kmeans_iters, uint64_t sample, phnsw_store **out);
/* The reference's own quantizer shape (src/pq.rs:19-27, 61-81, 261-364): ONE codebook of n_centroids <= 65535
 * centroid sub-vectors of dsub floats shared by every sub-space (random_centroids: the sub-vectors of selected
 * vectors, sorted, de-duplicated, shuffled, truncated), u16 codes, quantize = the best result of an HNSW search
 * (quantized_search) over the centroids (built with centroid_bp, centroid_metric).  A stored vector IS its
 * reconstruction and distances apply the store's metric to reconstructions (the quantised comparators of
 * pq.rs:585-599), so the store is searched with phnsw_search_batch* / phnsw_pq_search_batch like any other.
 * Build the Hnsw over the quantised vectors on phnsw_pq_shared_reconstruct_store (same distance bits) and
 * adopt it with phnsw_index_from_layers.  dim % dsub == 0, dsub % 4 == 0. */
int phnsw_store_create_pq_shared(phnsw_store *full, uint32_t dsub, uint32_t n_centroids, uint64_t seed,
                                 const phnsw_build_params *centroid_bp, const phnsw_search_params *quantized_search,
                                 int centroid_metric, phnsw_store **out);
/* Both constructors with Quantizer::quantize split over the ranks of `comm` (PQ encode shards by
 * vector range, pq.rs:326-333): codebooks / the centroid index are computed identically on every rank, rank r
 * encodes vectors [r*chunk, (r+1)*chunk), the code rows are all-gathered (n x m code bytes, resp. n x m x 2).
 * comm == NULL is the single-GPU call.  Every rank passes the same store contents and parameters. */
int phnsw_store_create_pq_sharded(phnsw_store *full, uint32_t m, uint32_t ksub, uint64_t seed, uint32_t kmeans_iters,
                                  uint64_t sample, const phnsw_comm *comm, phnsw_store **out);
int phnsw_store_create_pq_shared_sharded(phnsw_store *full, uint32_t dsub, uint32_t n_centroids, uint64_t seed,
                                         const phnsw_build_params *centroid_bp,
                                         const phnsw_search_params *quantized_search, int centroid_metric,
                                         const phnsw_comm *comm, phnsw_store **out);
int phnsw_pq_shared_read(const phnsw_store *s, uint16_t *codes, float *codebook);
int phnsw_pq_shared_reconstruct_store(const phnsw_store *s, phnsw_store **out);
int phnsw_pq_info(const phnsw_store *s, uint32_t *m, uint32_t *ksub, uint32_t *dsub);
/* storage of the per-query lookup table: 0 = f32 (reference arithmetic), 1 = IEEE half entries,
 * 2 = 8-bit entries with a per-query scale (integer sums; fewest L2 requests per hop).  Modes 1
 * and 2 change the quantised distances.  Mode 1 must be set before building an index over the
 * store.  Mode 2 scales by the query's own table, so d(a,b) != d(b,a): it is for SEARCHING a graph
 * built in mode 0 or 1 (build entry points refuse it with PHNSW_E_UNSUPPORTED). */
int phnsw_pq_set_table_mode(phnsw_store *s, int mode);
int phnsw_pq_set_table_f16(phnsw_store *s, int on); /* = set_table_mode(s, on ? 1 : 0) */
int phnsw_pq_read(const phnsw_store *s, uint8_t *codes, float *codebook);
/* Quantizer::quantize / ::reconstruct  src/pq.rs:61-81 for n host vectors [n][dim] <-> codes [n][m] */
int phnsw_pq_quantize(const phnsw_store *s, const float *rows, uint64_t n, uint8_t *out_codes);
int phnsw_pq_reconstruct(const phnsw_store *s, const uint8_t *codes, uint64_t n, float *out_rows);
/* QuantizedHnsw::search  pq.rs:346-364 for a batch: search the index over the PQ store, re-rank
 * every result with the full-precision store, sort by (distance, id).  quantize_query != 0
 * quantises the query first like the reference (pq.rs:351-352); 0 = asymmetric (raw query
 * against codes).  Outputs as phnsw_search_batch. */
int phnsw_pq_search_batch(const phnsw_index *ix, const phnsw_store *full, const float *queries, uint64_t nq,
                          const phnsw_search_params *sp, int quantize_query, uint64_t *out_ids,
                          float *out_d, uint64_t *out_len, uint64_t *out_stats);

/* zero-copy form (asymmetric queries): search + re-rank kernels enqueued on `stream`, u32 ids */
int phnsw_pq_search_batch_device(const phnsw_index *ix, const phnsw_store *full, const float *queries_dev,
                                 uint32_t ldq, uint64_t nq, const phnsw_search_params *sp,
                                 uint32_t *out_ids_dev, float *out_d_dev, uint32_t *out_len_dev,
                                 uint32_t *out_stats_dev, uint32_t *status_dev, void *stream);

/* ---- on-disk interchange with the Rust crate: serialize_hnsw / deserialize_hnsw
 * (src/serialize.rs:33-209): <dir>/meta (JSON HNSWMeta), <dir>/comparator/ (this library's
 * store; a crate user substitutes their own Serializable comparator), layer.meta.N (JSON),
 * layer.nodes.N / layer.neighbors.N (raw native-endian u64, !0 = empty), N from the bottom.
 * Deserialising needs the store the index was built over (the crate's C::Params role); a
 * missing comparator entry is "Index not found" (serialize.rs:144-146). ---- */
int phnsw_index_serialize(const phnsw_index *ix, const char *path);
int phnsw_index_deserialize(phnsw_store *s, const char *path, phnsw_index **out);
int phnsw_index_build_params(const phnsw_index *ix, phnsw_build_params *bp);

/* Hnsw::knn  src/lib.rs:905-928 : bottom layer, out [node_count][k] */
int phnsw_knn(const phnsw_index *ix, uint64_t k, uint64_t probe_depth, uint64_t *out_ids,
              float *out_d, uint64_t *out_len);

/* ---- exact k nearest neighbours by brute force: the ground truth for recall@k (the reference
 * only measures self-recall, lib.rs:1485-1496).  A true GEMM (queries x base rows) on the f32
 * MFMA units, exact f32 with a k-ordered fma chain per score; dot-product metrics, k <= 16.
 * Results sorted by (distance, id). ---- */
int phnsw_bruteforce_topk(const phnsw_store *s, const float *queries, uint64_t nq, uint32_t k,
                          uint64_t *out_ids, float *out_d);
int phnsw_bruteforce_topk_device(const phnsw_store *s, const float *queries_dev, uint32_t ldq, uint64_t nq,
                                 uint32_t k, uint32_t *out_ids_dev, float *out_d_dev, void *stream);
float phnsw_bruteforce_last_gemm_ms(void); /* MFMA GEMM time of this thread's last pass */

/* Hnsw::threshold_nn  src/lib.rs:930-962 : bottom layer; out [node_count][max_out], entries
 * with distance < threshold, self removed.  The queue doubles without bound like the reference's
 * (resize_capacity, lib.rs:949-951): in LDS up to 1024 entries, in global memory beyond. */
int phnsw_threshold_nn(const phnsw_index *ix, float threshold, uint64_t probe_depth,
                       uint64_t initial_search_depth, uint64_t max_out, uint64_t *out_ids,
                       float *out_d, uint64_t *out_len);

#ifdef __cplusplus
}
#endif
#endif
