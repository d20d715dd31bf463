// Internal declarations of libphnsw (gfx950 only).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/phnsw.h"

#define PH_EMPTY32 0xFFFFFFFFu
#define PH_FMAX 3.4028234663852886e38f
#define PH_MAX_LAYERS 24
#define PH_WAVE 64
#define PH_MAX_DISPATCH 24
#define PH_DENSE_SPLIT_MIN 2048u  // batches from this size on walk their dense top layers in a launch of its own  // search launches of one descent (<= PH_MAX_LAYERS)
#define PH_TINY_MAX_NODES 8192u      // largest layer evaluated densely (tiny.hip)
#define PH_TINY_LDS_NODES 1024u      // up to here a query's table row is staged in LDS
#define PH_TINY_MAX_LAYERS 8

void ph_set_error(const char *fmt, ...);
// every extern "C" entry point is a function-try-block ending in `catch (...) { return ph_caught(); }`:
// nothing unwinds across the ABI (phnsw.h), a C++ exception becomes a status + message
int ph_caught() noexcept;
int ph_hip_fail(hipError_t e, const char *what, const char *file, int line);
#define PH_HIP(x)                                                       \
  do {                                                                  \
    hipError_t e__ = (x);                                               \
    if (e__ != hipSuccess) return ph_hip_fail(e__, #x, __FILE__, __LINE__); \
  } while (0)

// ---- device view of one layer (Layer<C>, src/lib.rs:85-91) ----
// node ids and vector ids are u32 on the device (n <= 2^31 - 1); the u64 <-> u32
// conversion and the !0 sentinel mapping happen at the ABI boundary.
struct PhLayerDev {
  uint32_t n_nodes;
  uint32_t W;                 // neighborhood_size
  const uint32_t *nodes;      // [n_nodes] NodeId -> VectorId, ascending
  const uint32_t *neighbors;  // [n_nodes * W], trailing PH_EMPTY32
  const uint32_t *vec2node;   // [store n] VectorId -> NodeId or PH_EMPTY32; nullptr = identity
};

struct PhLayerHost {
  uint32_t n_nodes = 0, W = 0;
  uint32_t *nodes = nullptr;
  uint32_t *neighbors = nullptr;
  float *nbr_dist = nullptr;  // [n_nodes * W] distance of each occupant to the row owner (build only)
  uint32_t *vec2node = nullptr;
  bool identity = false;
  uint32_t *recall_q = nullptr;  // cached stochastic_recall_at sample (build.hip), recall_n entries
  uint32_t recall_n = 0;
  // locality schedule: pos[node] = the node's coarse cell (chain rank of its nearest anchor,
  // anchors = a strided sample of the store; bruteforce.hip); ord = argsort(pos[ord_first .. +ord_count)), cached
  uint32_t *pos = nullptr;
  uint32_t *ord = nullptr;
  uint32_t ord_first = 0, ord_count = 0;
};

// what a distance evaluation needs: the f32 store, or -- for a product-quantised store --
// code rows + per-subspace codebooks (pq.hip)
struct PhDistArgs {
  const float *vecs;  // [n][ld] f32 rows (nullptr for a PQ store)
  uint32_t ld, nv4;
  int metric;
  const uint8_t *codes;   // [n][m] u8 codes (PQ store)
  const uint16_t *codes16;  // [n][m] u16 codes over ONE shared codebook [ksub][dsub] (the reference's shape, pq.rs:19-27)
  const float *codebook;  // [m][ksub][dsub]
  uint32_t m, ksub, dsub;
  uint32_t table_f16;  // 1: the per-query table holds IEEE half values (2 bytes per entry)
};

struct phnsw_store {
  int device = 0;
  float *rows = nullptr;  // [n][ld] flat, HBM
  // product-quantised store (phnsw_store_create_pq): rows == nullptr, dim/ld describe the
  // full vectors a query has
  uint8_t *codes = nullptr;
  uint16_t *codes16 = nullptr;  // shared-codebook PQ store (pq.hip): codes [n][pq_m], codebook [pq_ksub][pq_dsub]
  phnsw_store *centroid_store = nullptr;  // ... its centroids as a store of their own and the HNSW over them
  phnsw_index *centroid_index = nullptr;  //     (HnswQuantizer, pq.rs:29-81)
  phnsw_search_params quantized_search = {0, 0, 0};
  float *codebook = nullptr;
  uint32_t pq_m = 0, pq_ksub = 0, pq_dsub = 0;
  uint32_t pq_table_f16 = 0;
  // coarse cells of the locality schedule (bruteforce.hip): anchor rows + their chain ranks
  float *anchors = nullptr;
  uint32_t *anchor_rank = nullptr;
  uint32_t n_anchors = 0;
  bool owns_rows = true;
  uint64_t n = 0;
  uint32_t dim = 0, ld = 0;
  int metric = 0;
  int refcount = 1;
};

struct PhWorkspace {
  // per resident-wave search state: visited bitmap + overflow (frontier spill) list
  uint32_t *visited = nullptr;
  uint64_t visited_words = 0;  // per slot
  uint2 *ovf = nullptr;
  uint32_t ovf_cap = 0;  // entries per slot
  uint32_t n_slots = 0;
  uint32_t *counter = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t evc = nullptr;  // start of the last chunk of the descent (== ev0 when the list ran in one piece)
  bool timed = false;
  void *pq_tables = nullptr;  // n_slots x table bytes (PQ stores)
  size_t pq_tables_bytes = 0;
  // two-launch descents: per-query locality keys, their argsort, radix-sort scratch
  uint32_t *okey = nullptr, *okey_sorted = nullptr, *oiota = nullptr, *oorder = nullptr;
  void *sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  uint32_t order_cap = 0;
  // dense top layers (tiny.hip): distance table [positions][stride], neighbour rows in table ids,
  // per-node layer membership (+ one flag word)
  float *tiny_d = nullptr;
  size_t tiny_d_bytes = 0;
  uint32_t *tiny_nbr = nullptr;
  size_t tiny_nbr_bytes = 0;
  uint32_t *tiny_member = nullptr;
  size_t tiny_member_bytes = 0;
  float4 *tiny_pq = nullptr, *tiny_pn = nullptr;  // matrix-core operands: packed positions / nodes (tiny.hip)
  size_t tiny_pq_bytes = 0, tiny_pn_bytes = 0;
  uint2 *dense_ovf = nullptr;  // spill lists of the dense-only launch: [its resident waves][table nodes]
  size_t dense_ovf_bytes = 0;
  uint32_t *ovf_s = nullptr;  // search_instrumented: index sums of the spilled entries, [n_slots][ovf_cap]
  size_t ovf_s_cap = 0;
  // per-dispatch bookkeeping of the last descent (phnsw_last_search_dispatches): evd[0] closes the
  // dense-top-layer kernels, evd[1 + i] search launch i; dtotals[i] = {distance evaluations, hops}
  hipEvent_t evd[PH_MAX_DISPATCH + 1] = {};
  unsigned long long *dtotals = nullptr;  // device [PH_MAX_DISPATCH][2], then [PH_MAX_DISPATCH] table-served evaluations
  uint32_t n_dispatch = 0;
  uint32_t n_chunks = 0;  // chunks of the last descent (a query list longer than the dense table holds): the per-dispatch
                          // figures then describe the LAST chunk, times and counters alike
  int cus = 0;            // compute units of the workspace's device
  uint32_t d_lo[PH_MAX_DISPATCH] = {}, d_hi[PH_MAX_DISPATCH] = {};
  bool d_tiny = false;
};

// The dense distance table of the BUILD's searches, kept across rounds (tiny.hip).  A link round, a recall estimate and
// discover_unreachable all search with Stored(node of layer X) queries, and the distance of such a node to the nodes
// of the table layer does not change between rounds: row r = node r of layer X, computed once, reused until a node
// list changes (epoch).  17 % of a 1M build's GPU time was recomputing it every round.
struct PhBuildTable {
  float *D = nullptr;       // rows [lo_alloc, hi_alloc) x stride
  size_t bytes = 0;         // ... in use; `cap` allocated
  size_t cap = 0;
  const uint32_t *qnodes = nullptr, *tnodes = nullptr;  // identity of layer X and of the table layer
  uint32_t qn = 0, tn = 0, T = 0, stride = 0;
  uint32_t lo_alloc = 0, hi_alloc = 0, lo = 0, hi = 0;  // allocated rows; valid rows [lo, hi)
  uint64_t epoch = 0;
};
// which layer the Stored queries of a build search are nodes of (build.hip -> ph_search_device)
struct PhRowHint {
  const uint32_t *qnodes;    // device: node list of layer X
  uint32_t qn;
  const uint32_t *vec2node;  // device: VectorId -> NodeId of layer X, nullptr = identity
  uint32_t first, count;     // contiguous: the queries are nodes [first, first + count) of X
  bool contiguous;           // false: any nodes of X (a recall sample)
};

struct PhPendingLayer;
struct PhHostStage;  // hostpath.hip: persistent staging of the host-pointer search entry points
struct phnsw_index {
  std::vector<PhHostStage *> stages;  // handed out under stage_mutex, one per concurrent host-pointer call
  std::mutex stage_mutex;
  std::vector<PhBuildTable> bt;  // one per layer whose nodes query; build entry points only (exclusive access to the index)
  // buffers of dropped tables, reused by the next table that fits: a build re-makes its tables whenever a promotion
  // changes a node list, and device allocation is host work whose cost varies with what else the host is doing
  std::vector<std::pair<float *, size_t>> bt_spare;
  bool bt_enabled = false;    // set for the duration of a build / improve_index call: nothing else may keep 29 GB
  uint64_t nodes_epoch = 0;   // bumped whenever a layer's node list is created, replaced or dropped
  PhPendingLayer *pending = nullptr;  // layer under construction (phase API, build.hip)
  phnsw_store *store = nullptr;
  std::vector<PhLayerHost> layers;  // top first (src/lib.rs:587)
  phnsw_build_params bp;
  std::mutex ws_mutex;
  // two workspaces, used alternately: launches on different streams may overlap (the tail of
  // one batch with the head of the next); a third launch waits for the first
  PhWorkspace ws[2];
  uint32_t ws_next = 0, ws_last = 0;
  float last_kernel_ms = 0.f;
  unsigned long long *totals = nullptr;  // device [2]: distance evaluations, hops of every launch on this index
  const uint32_t *dbg_order = nullptr;  // experiment hook
  uint64_t dbg_order_n = 0;
};

// ---- kernel argument block for the batched greedy search ----
struct PhSearchArgs {
  PhDistArgs dist;
  const float *queries;  // [nq][ldq] or nullptr
  uint32_t ldq;
  const uint32_t *qids;     // or nullptr
  const uint32_t *exclude;  // or nullptr
  uint32_t nq;
  uint32_t n_layers;
  PhLayerDev layers[PH_MAX_LAYERS];
  uint32_t ef, upper, probe_depth;
  uint32_t *out_ids;
  float *out_d;
  uint32_t *out_len;
  uint32_t *out_stats;  // [nq][2] or nullptr
  uint32_t *status;     // [nq]
  uint32_t *visited;
  uint64_t visited_words;
  uint2 *ovf;
  uint32_t ovf_cap;
  uint32_t *counter;
  // 1 = Hnsw::knn (lib.rs:905-928): bottom layer only, query = node, seed (node, 0.0);
  // 2 = Hnsw::threshold_nn (lib.rs:930-962): the same, repeated with a doubling queue
  uint32_t knn_mode;
  float threshold;
  uint32_t first_node;  // knn modes: queries are nodes first_node .. first_node + nq
  uint32_t cap_max;     // threshold_nn: largest queue capacity the launch must support (0 = ef)
  // threshold_nn beyond the LDS queues (ph_search_kernel_big): an explicit node list and, per resident wave, a
  // queue of cap_max (id, distance) slots in global memory
  const uint32_t *knn_nodes;
  uint32_t *big_q;
  // dense layers: a query's table row is staged in LDS when the table layer has at most this many nodes
  // (PH_TINY_LDS_NODES; the one-wave-per-SIMD kernels of small batches raise it to what a CU's LDS holds)
  uint32_t tiny_lds_nodes;
  float hit_eps;        // out_hit: > 0 selects match_within_epsilon (search.rs:173-187)
  uint32_t out_stride;  // entries written per query (0 = ef); link rounds keep only the top M
  unsigned long long *totals;   // nullable: [2] running sums of distance evaluations / hops (all launches)
  unsigned long long *launch_totals;  // nullable: [2] the same for this launch alone
  unsigned long long *launch_tab;     // nullable: of this launch's evaluations, those the dense tables served
  void *pq_tables;              // PQ store: per-wave lookup-table slots in global memory (DistPQG)
  uint32_t pq_table_bytes;      // bytes per slot
  uint32_t layer_lo, layer_hi;  // layers of this launch (0, 0 = all); see search.hip
  uint32_t *out_key;            // nullable: locality key of each query after layer_hi - 1
  const uint32_t *key_pos;      // nullable: pos[] of layer layer_hi - 1
  // A split descent walks its dense top layers in a launch of its own (ph_search_kernel_dense: no row registers, so
  // twice the resident waves): dense_only marks that launch (layers [0, tiny_layers), spill lists of dense_ovf_cap
  // entries in dense_ovf); after_dense = T marks the launch that follows it.  Whether the table was usable (layers
  // nested) is only known on the device (dense_flag[0] == 0): if not, the dense launch does nothing and the
  // follow-up starts from layer 0 on the per-hop path.
  uint32_t dense_only, after_dense;
  const uint32_t *dense_flag;
  uint2 *dense_ovf;
  uint32_t dense_ovf_cap;
#ifdef PH_CELL_PROBE
  // experiment: how far, in cell-chain ranks, the bottom layer's evaluations stray from where the query landed
  const uint32_t *probe_pos;
  unsigned long long *probe_out;  // [12]: evaluations within 0,1,2,4,...,256 ranks; [10] all; [11] queries
#endif
  const uint32_t *order;  // nullable: processing order (a permutation of 0..nq-1), see search.hip
  uint32_t seg;           // order != nullptr: positions per XCD segment
  uint32_t *out_hit;    // nullable: 1 when a Stored query found itself (stochastic_recall lib.rs:1492)
  uint32_t *out_index;  // nullable: Hnsw::search_instrumented's index_distance per query (selects the INSTR kernels)
  uint32_t *ovf_s;      // ... and the index sums of the spilled entries, parallel to ovf
  // dense top layers (tiny.hip): layers [0, tiny_layers) of this launch hold <= PH_TINY_MAX_NODES
  // nodes; the distance of every query to every node of layer tiny_layers - 1 sits in tiny_d
  // (same bits as the per-hop evaluation), the traversal of those layers runs in the id space
  // of that layer ("table ids") with its visited set in LDS (and its table row, when small)
  uint32_t tiny_layers, tiny_n, tiny_stride;
  const float *tiny_d;          // [launch positions][tiny_stride]
  const uint32_t *tiny_nbr;     // layer l: [tiny_n][W_l] at tiny_off[l], table ids, PH_EMPTY32 padded
  const uint32_t *tiny_member;  // [tiny_n] bit l = node of layer l; [tiny_n] != 0: layers not nested, table off
  // tiny_rows != 0: tiny_d is the build's kept table -- the row of a Stored query is its NodeId in layer X
  // (tiny_row_map: VectorId -> NodeId, nullptr = identity) minus tiny_row_first, not its launch position
  uint32_t tiny_rows, tiny_row_first;
  const uint32_t *tiny_row_map;
  uint32_t tiny_off[PH_TINY_MAX_LAYERS];
};

uint32_t ph_default_ovf_cap(uint32_t ef);
// scratch allocator of the build path (misc.hip): hipMalloc / hipFree cost ~0.1 ms and a device
// sync each, a build issues thousands of them; freed blocks are kept by size class and handed
// out again (everything on the path runs on the null stream, so reuse is stream ordered)
hipError_t ph_timed_malloc(void **p, size_t bytes);  // hipMalloc / hipFree with their wall time accounted (misc.hip)
hipError_t ph_timed_free(void *p);
#ifndef PH_RAW_ALLOC  // every device allocation of the library goes through the accounted forms
#define hipMalloc(p, n) ph_timed_malloc((void **)(p), (n))
#define hipFree(p) ph_timed_free((void *)(p))
#endif
int ph_stream_beside(hipStream_t other, hipStream_t *io);  // a non-blocking stream on another hardware queue than `other` (misc.hip)
hipError_t ph_pool_alloc(void **p, size_t bytes);
void ph_pool_free(void *p);
void ph_pool_trim(void);  // give everything cached back to the driver
void ph_layer_free(PhLayerHost &l);
void ph_pending_free(phnsw_index *ix);
void ph_host_stages_free(phnsw_index *ix);  // hostpath.hip
int ph_layer_upload(phnsw_index *ix, const uint32_t *nodes, const uint32_t *neighbors, uint32_t n, uint32_t W,
                    PhLayerHost *out);
int ph_search_device(const phnsw_index *ix, const float *queries_dev, uint32_t ldq, const uint32_t *qids_dev,
                     uint64_t nq, const phnsw_search_params *sp, uint32_t upto, const uint32_t *exclude_dev,
                     uint32_t *out_ids, float *out_d, uint32_t *out_len, uint32_t *out_stats, uint32_t *status,
                     uint32_t ovf_cap, uint32_t knn_mode, hipStream_t stream, uint32_t out_stride = 0,
                     uint32_t *out_hit = nullptr, float threshold = 0.f, uint32_t first_node = 0,
                     float hit_eps = 0.f, const uint32_t *order = nullptr, uint32_t *out_index = nullptr,
                     const PhRowHint *hint = nullptr);
// locality schedule helpers (group.hip / api.hip)
#define PH_ORDER_MIN 16384u  // shorter query lists run in natural order
#define PH_POS_MIN 256u     // smaller layers carry no cells (their node id is the key)
// a layer whose vector rows do not fit one XCD's L2 (4 MiB) gets a launch of its own in a split
// descent, its queries sorted by the cell they arrive in (PHNSW_SPLIT_BYTES overrides: tuning knob)
#define PH_SPLIT_BYTES (4ull << 20)
static inline bool ph_layer_own_launch(uint64_t n_nodes, uint32_t ld) {
  const char *e = getenv("PHNSW_SPLIT_BYTES");  // read per call: the tests switch it
  const uint64_t limit = (e && atoll(e) > 0) ? (uint64_t)atoll(e) : (uint64_t)PH_SPLIT_BYTES;
  return n_nodes * (uint64_t)ld * 4u > limit;
}
int ph_layer_anchor_pos(const phnsw_store *s, PhLayerHost &L);  // bruteforce.hip
bool ph_layer_wants_cells(const phnsw_store *s, uint32_t n_nodes);
int ph_layer_cells_range(const phnsw_store *s, const PhLayerHost &L, uint32_t first, uint32_t count, uint32_t *out_pos);
void ph_store_anchors_free(phnsw_store *s);
int ph_order_by_keys_device(const uint32_t *keys, uint32_t n, uint32_t *order_out, hipStream_t st);
int ph_layer_range_order(PhLayerHost &L, uint32_t first, uint32_t count, const uint32_t **out);

// dense top layers (tiny.hip): decides how many leading layers of the launch described by `a` run
// against a distance table, fills a.tiny_* and enqueues the table kernels for launch positions
// [0, npos) (position p = query order[p], or p itself) on `stream`.  a.tiny_layers = 0 when unused.
int ph_tiny_prepare(const phnsw_index *ix, PhWorkspace &ws, PhSearchArgs &a, uint32_t max_layers, hipStream_t stream);
// the build's kept table: rows for the hinted queries are made available (computed where missing) and the launch is
// pointed at them; *used = false when the table cannot serve this launch (then ph_tiny_prepare does, per launch)
int ph_build_table_prepare(phnsw_index *ix, PhWorkspace &ws, PhSearchArgs &a, const PhRowHint &h, uint32_t T,
                           hipStream_t stream, bool *used);
void ph_build_table_free(phnsw_index *ix);
size_t ph_tiny_lds_bytes(const PhSearchArgs &a);
bool ph_tiny_matrix_cores(const phnsw_index *ix);  // the table of this index's store is built by the MFMA kernel
uint32_t ph_tiny_layer_count(const phnsw_index *ix, uint32_t n_layers, uint32_t ef);  // leading layers a launch may run densely
uint64_t ph_tiny_max_positions(const phnsw_index *ix, uint32_t n_layers, uint32_t ef);  // 0 = no dense layers for this launch shape
void ph_tiny_free(PhWorkspace &ws);

// launchers (search.hip)
int ph_search_begin(PhWorkspace &ws, hipStream_t stream);
int ph_search_launch(const phnsw_index *ix, PhWorkspace &ws, PhSearchArgs &a, hipStream_t stream, bool mark_end = true);
// threshold_nn's global-memory queues: `a` complete (visited / ovf / counter / big_q are the caller's, `grid` slots each)
int ph_search_launch_big(const phnsw_index *ix, PhSearchArgs &a, uint32_t grid, hipStream_t stream);
int ph_workspace_order_ensure(PhWorkspace &ws, uint32_t nq);                          // group.hip
int ph_workspace_order_sort(PhWorkspace &ws, uint32_t nq, hipStream_t stream);        // group.hip
void ph_workspace_order_free(PhWorkspace &ws);                                        // group.hip
#define PH_TWO_LAUNCH_MIN 32768u  // batches at least this large descend in two launches
int ph_workspace_ensure(const phnsw_index *ix, PhWorkspace &ws, uint32_t ef, uint32_t ovf_cap);
void ph_workspace_free(PhWorkspace &ws);
uint32_t ph_search_slots(uint32_t ef, uint32_t nv4, bool pq, size_t pq_lds, int pqr_m = 0);
static inline PhDistArgs ph_dist_args(const phnsw_store *s) {
  PhDistArgs d;
  d.vecs = s->rows;
  d.ld = s->ld;
  d.nv4 = s->ld / 4;
  d.metric = s->metric;
  d.codes = s->codes;
  d.codes16 = s->codes16;
  d.codebook = s->codebook;
  d.m = s->pq_m;
  d.ksub = s->pq_ksub;
  d.dsub = s->pq_dsub;
  d.table_f16 = s->pq_table_f16;
  return d;
}
// the batched search keeps PQ tables in global memory unless PHNSW_PQ_TABLE=lds
static inline bool ph_pq_global_tables() {
  const char *e = getenv("PHNSW_PQ_TABLE");
  return !(e && e[0] == 'l');
}
static inline size_t ph_pq_lds_bytes(const phnsw_store *s) {
  return s->codes ? (size_t)s->pq_m * s->pq_ksub * (s->pq_table_f16 == 2 ? 1 : (s->pq_table_f16 ? 2 : 4)) : 0;
}

// sharded.hip: rank r's share of n items, and a synchronous all-gather of device blocks over any transport
void ph_comm_range(const phnsw_comm *comm, uint32_t r, uint64_t n, uint64_t *chunk, uint64_t *first, uint64_t *count);
int ph_comm_all_gather_device(const phnsw_comm *comm, const void *send_dev, void *recv_dev, uint64_t bytes);

// misc kernels (misc.hip)
int ph_synth_rows(float *rows_dev, uint64_t first, uint64_t count, uint32_t dim, uint32_t ld,
                  uint64_t seed, int normalize, hipStream_t s);
int ph_synth_clustered_rows(float *rows_dev, uint64_t first, uint64_t count, uint32_t dim, uint32_t ld, uint64_t seed,
                            uint32_t n_clusters, float noise, hipStream_t s);
// q_dev == nullptr: the query is Stored(query_id)
int ph_distance_batch(const phnsw_store *st, const float *q_dev, uint32_t query_id, const uint32_t *ids_dev,
                      uint32_t k, float *out_dev, hipStream_t s);
int ph_fill_u32(uint32_t *p, uint32_t v, uint64_t n, hipStream_t s);
int ph_scatter_vec2node(const uint32_t *nodes, uint32_t n, uint32_t *vec2node, hipStream_t s);
int ph_count_nan(const float *rows, uint64_t n_floats, uint32_t *out_count_dev, hipStream_t s);

// deterministic generators shared by definition with the oracle (host + device)
__host__ __device__ inline uint64_t ph_mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBULL;
  x ^= x >> 31;
  return x;
}
__host__ __device__ inline uint64_t ph_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((__uint128_t)a * b) >> 64);
#endif
}
__host__ __device__ inline uint64_t ph_feistel_perm(uint64_t i, uint64_t domain, uint64_t key) {
  if (domain <= 1) return 0;
  uint32_t bits = 0;
  while (((uint64_t)1 << bits) < domain) bits++;
  uint32_t h = (bits + 1) / 2;
  if (h == 0) h = 1;
  uint64_t mask = ((uint64_t)1 << h) - 1;
  uint64_t x = i;
  do {
    uint64_t L = x >> h, R = x & mask;
    for (uint32_t r = 0; r < 4; r++) {
      uint64_t f = ph_mix64(R + key * 0x9E3779B97F4A7C15ULL + r * 0xC2B2AE3D27D4EB4FULL) & mask;
      uint64_t nl = R;
      R = L ^ f;
      L = nl;
    }
    x = (L << h) | R;
  } while (x >= domain);
  return x;
}
void ph_shuffle_u64(uint64_t *v, uint64_t n, uint64_t seed);
