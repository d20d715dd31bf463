/*
 * orc.h -- CPU ORACLE for the parallel-hnsw hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is a plain-C restatement of the reference crate's algorithm
 * (terminusdb-labs/parallel-hnsw, Rust) for the search / build hot path.  It is the
 * checker the HIP product path is compared against and the "port" CPU baseline that
 * bench.py times.  Nothing in the shipped package (parallel_hnsw_amd/, include/) may
 * include, link or call anything in oracle/: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do.
 *
 * Parity pinning: the reference is Rust and there is no Rust toolchain in the build
 * image, so there is no oracle/_ref build.  The restatement is pinned by the golden
 * vectors held in the reference's own unit tests (tests/golden/, transcribed from
 *   src/priority_queue.rs:229-439, src/lib.rs:1996-2006, 2046-2068, 2300-2304,
 *   2345-2354, 2358-2420, 2476-2512).
 * What those vectors do NOT pin (the rand-crate streams, the racy link order, traversal
 * on large graphs) is listed in DESIGN.md as "parity unpinned".
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/types.rs:8-13  VectorId::MAX / NodeId::MAX = !0 ; src/priority_queue.rs:170 f32::MAX */
#define ORC_EMPTY UINT64_MAX
#define ORC_FMAX 3.4028234663852886e38f

/* ---- metrics (the Comparator::compare_raw implementations in the tree) ---- */
enum {
  ORC_METRIC_COSINE_HALF = 0,   /* (1 - dot)/2   src/bigvec.rs:47-53   */
  ORC_METRIC_ONE_MINUS_DOT = 1, /* 1 - dot       src/lib.rs:1985-1991  */
  ORC_METRIC_L2 = 2             /* sqrt(sum (a-b)^2)  src/lib.rs:2431-2437, src/pq.rs:499-505 */
};
/* f32 summation order.  SEQ is the reference's (sequential, mul then add, no fma).
 * BLOCKED64 is the order the gfx950 kernel uses (one fma chain per lane of a 64-lane
 * wavefront over 16-byte chunks l, l+64, ..., then an xor-butterfly 32,16,..,1), so the
 * HIP path can be checked bit-exactly; see DESIGN.md "summation order". */
/* SEQFMA: sequential like the reference but with one rounding per step (fmaf chain) -- the
 * order of the f32 MFMA units, used by the brute-force ground-truth kernel */
enum { ORC_SUM_SEQ = 0, ORC_SUM_BLOCKED64 = 1, ORC_SUM_SEQFMA = 2 };

typedef struct {
  const float *rows; /* [n * ld], row i at rows + i*ld, floats dim..ld-1 are zero */
  uint64_t n;
  uint32_t dim;
  uint32_t ld;
  int metric;
  int sum_mode;
  /* product-quantised store (pq.rs): when codes != NULL the rows of the store are u8 code rows
   * [n][pq_m] over per-sub-space codebooks [pq_m][pq_ksub][pq_dsub]; queries stay f32 */
  const uint8_t *codes;
  const float *codebook;
  uint32_t pq_m, pq_ksub, pq_dsub;
  uint32_t pq_table_f16; /* table mode: 0 f32, 1 entries rounded to IEEE half once, 2 8-bit entries */
} orc_store;

float orc_distance(const orc_store *s, const float *a, const float *b);

/* ---- PriorityQueue  src/priority_queue.rs:28-223 ---- */
typedef struct {
  uint64_t *data;
  float *prio;
  uint64_t cap;
} orc_pq;
uint64_t orc_pq_len(const orc_pq *q);
uint64_t orc_pq_insert(orc_pq *q, uint64_t elt, float priority);
int orc_pq_merge(orc_pq *q, const uint64_t *ids, const float *prios, uint64_t m);
uint64_t orc_pq_iter_len(const orc_pq *q); /* number of items iter() yields */

/* ---- Layer  src/lib.rs:85-159 ---- */
typedef struct {
  uint64_t node_count;
  uint64_t neighborhood_size;
  uint64_t *nodes;     /* [node_count] sorted ascending VectorIds */
  uint64_t *neighbors; /* [node_count * neighborhood_size], trailing ORC_EMPTY */
} orc_layer;
uint64_t orc_final_neighbor_idx(uint64_t neighborhood_size, const uint64_t *neighbors, uint64_t n);

/* ---- parameters  src/parameters.rs ---- */
typedef struct {
  uint64_t number_of_candidates;
  uint64_t upper_layer_candidate_count;
  uint64_t probe_depth;
} orc_search_params;

typedef struct {
  float promotion_threshold;
  float neighborhood_threshold;
  float recall_proportion;
  float promotion_proportion;
  orc_search_params search;
} orc_opt_params;

typedef struct {
  uint64_t order;
  uint64_t zero_layer_neighborhood_size;
  uint64_t neighborhood_size;
  orc_opt_params optimization;
  orc_search_params initial_partition_search;
  uint64_t seed;           /* replaces thread_rng (lib.rs:832) */
  uint64_t max_link_rounds; /* 0 = reference loop (until improvement < threshold) */
  uint64_t promote;         /* 1 = promote_at_layer as the reference does (lib.rs:1580); 0 = skip */
} orc_build_params;
void orc_default_build_params(orc_build_params *bp);

/* ---- index handle ---- */
typedef struct orc_index orc_index;
orc_index *orc_index_new(const float *rows, uint64_t n, uint32_t dim, uint32_t ld, int metric,
                         int sum_mode);
void orc_index_free(orc_index *ix);
void orc_index_set_sum_mode(orc_index *ix, int sum_mode);
/* append a layer BELOW the existing ones (layers are stored top first, lib.rs:587) */
int orc_index_push_layer(orc_index *ix, const uint64_t *nodes, const uint64_t *neighbors,
                         uint64_t node_count, uint64_t neighborhood_size);
uint32_t orc_index_layer_count(const orc_index *ix);
const orc_layer *orc_index_layer(const orc_index *ix, uint32_t layer_from_top);
const orc_store *orc_index_store(const orc_index *ix);

typedef struct {
  uint64_t n_dist; /* Comparator::compare_vec calls */
  uint64_t n_hops; /* closest_nodes loop iterations */
} orc_stats;

/* search_layers  src/search.rs:84-140.  query==NULL => AbstractVector::Stored(qid).
 * upto_layers = number of layers from the top to use (Hnsw::search_upto lib.rs:654-661),
 * 0 = all.  exclude = ORC_EMPTY for None.  out_* sized number_of_candidates. */
int orc_search(const orc_index *ix, const float *query, uint64_t qid, orc_search_params sp,
               uint32_t upto_layers, uint64_t exclude, uint64_t *out_ids, float *out_d,
               uint64_t *out_len, orc_stats *st);
/* nq searches in parallel (rayon par_iter sites lib.rs:1107-1117, 2169-2184) */
int orc_search_batch(const orc_index *ix, const float *queries, uint32_t ldq,
                     const uint64_t *qids, uint64_t nq, orc_search_params sp,
                     const uint64_t *exclude, uint64_t *out_ids, float *out_d, uint64_t *out_len,
                     orc_stats *st, int threads);

/* Hnsw::search_instrumented lib.rs:667-673: results + index_distance per query */
int orc_search_batch_instrumented(const orc_index *ix, const float *queries, uint32_t ldq, const uint64_t *qids,
                                  uint64_t nq, orc_search_params sp, uint64_t *out_ids, float *out_d,
                                  uint64_t *out_len, uint64_t *out_index_distance, int threads);

/* Hnsw::knn lib.rs:905-928 ; out_[ids|d] sized node_count*k, out_len per node */
int orc_knn(const orc_index *ix, uint64_t k, uint64_t probe_depth, uint64_t *out_ids,
            float *out_d, uint64_t *out_len, int threads);
/* Hnsw::threshold_nn lib.rs:930-962 ; variable length results, max_out per node */
int orc_threshold_nn(const orc_index *ix, float threshold, uint64_t probe_depth,
                     uint64_t initial_search_depth, uint64_t max_out, uint64_t *out_ids,
                     float *out_d, uint64_t *out_len, int threads);

/* exact top-k by (d, id) -- ground truth for recall@k (not in the reference) */
int orc_bruteforce(const orc_store *s, const float *queries, uint32_t ldq, uint64_t nq,
                   uint64_t k, uint64_t *out_ids, float *out_d, int threads);

/* ---- build ---- */
/* calculate_partitions lib.rs:1883-1899 ; returns count, writes top-first sizes */
uint32_t orc_calculate_partitions(uint64_t total, uint64_t order, uint64_t *out, uint32_t max_out);
uint32_t orc_calculate_partitions_for_additions(const uint64_t *sizes_from_bottom, uint32_t n_sizes,
                                                uint64_t new_vecs, uint64_t order, uint64_t *out,
                                                uint32_t max_out);
/* Hnsw::generate lib.rs:825-893 (deterministic variant, promotion excluded) */
orc_index *orc_generate(const float *rows, uint64_t n_store, uint32_t dim, uint32_t ld, int metric,
                        int sum_mode, const uint64_t *vids, uint64_t n, const orc_build_params *bp,
                        int threads);
/* generate_layer lib.rs:675-823 appended to ix (no improve) */
int orc_generate_layer(orc_index *ix, const uint64_t *vs, uint64_t n, uint64_t neighborhood_size,
                       const orc_build_params *bp, int threads);
/* the same in phases over node ranges (what a multi-GPU driver shards; see orc_build.c) */
int orc_layer_begin(orc_index *ix, const uint64_t *vs, uint64_t n, uint64_t neighborhood_size,
                    const orc_build_params *bp);
uint64_t orc_layer_init_stride(const orc_index *ix);
int orc_layer_init_search(orc_index *ix, const orc_build_params *bp, uint64_t first, uint64_t count,
                          uint64_t *out_ids, float *out_d, uint64_t *out_len, int threads);
int orc_layer_seed(orc_index *ix, const orc_build_params *bp, const uint64_t *init_ids, const float *init_d,
                   const uint64_t *init_len, uint64_t first, uint64_t count, uint64_t *out_rows,
                   float *out_rows_d, int threads);
int orc_layer_finish(orc_index *ix, const uint64_t *rows, const float *rows_d, int threads);
int orc_link_search(orc_index *ix, uint32_t layer_from_top, orc_search_params sp, uint64_t link_count,
                    uint64_t first, uint64_t count, uint64_t *out_ids, float *out_d, uint64_t *out_len,
                    int threads);
uint64_t orc_link_apply(orc_index *ix, uint32_t layer_from_top, uint64_t link_count, const uint64_t *ids,
                        const float *d, const uint64_t *len, int threads);
int orc_recall_hits(const orc_index *ix, uint32_t at, const orc_opt_params *op, uint64_t first, uint64_t count,
                    uint64_t *out_hits, uint64_t *out_selection, int threads);
/* link_layer_to_better_neighbors lib.rs:1070-1154 ; returns new edges */
uint64_t orc_link_layer(orc_index *ix, uint32_t layer_from_top, orc_search_params sp,
                        uint64_t link_count, int threads);
/* stochastic_recall_at lib.rs:1463-1499 */
float orc_stochastic_recall_at(const orc_index *ix, uint32_t at, const orc_opt_params *op,
                               int threads);
/* improve_neighbors_upto lib.rs:1515-1544 */
float orc_improve_neighbors_upto(orc_index *ix, uint32_t upto, const orc_build_params *bp,
                                 float last_recall_or_nan, int threads);
/* improve_index lib.rs:1664-1686 minus promotion */
float orc_improve_index(orc_index *ix, const orc_build_params *bp, int threads);
float orc_improve_index_from(orc_index *ix, const orc_build_params *bp, float last_recall, int threads);
/* discover_unreachable_vectors lib.rs:1002-1037 ; returns count, *out malloc'd (caller frees) */
uint64_t orc_discover_unreachable(const orc_index *ix, uint32_t layer_from_top, orc_search_params sp,
                                  uint64_t **out, int threads);
/* promote_at_layer lib.rs:1273-1427 (deterministic tie order) ; 1 = promoted */
int orc_promote_at_layer(orc_index *ix, uint32_t layer_from_top, const orc_build_params *bp, int threads);
int orc_discover_hits(const orc_index *ix, uint32_t layer_from_top, orc_search_params sp, uint64_t first,
                      uint64_t count, uint64_t *out_hit, int threads);
int orc_promote_at_layer_hits(orc_index *ix, uint32_t layer_from_top, const orc_build_params *bp,
                              const uint64_t *hit, int threads);
/* extend_layer lib.rs:1039-1068 (layer counted from the top here) */
int orc_extend_layer(orc_index *ix, uint32_t layer_from_top, const uint64_t *vecs, uint64_t count);
/* assert_layer_invariants search.rs:142-171 ; 0 = ok */
int orc_check_layer_invariants(const orc_index *ix);

/* ---- deterministic generators shared by definition with the product ---- */
uint64_t orc_mix64(uint64_t x);
void orc_shuffle_u64(uint64_t *v, uint64_t n, uint64_t seed);
/* synthetic vector i (distribution of bigvec.rs:59-65): uniform(-1,1) components keyed
 * (seed+i, j), L2-normalised in f32 when normalize != 0 */
void orc_synth_rows(float *rows, uint64_t first, uint64_t count, uint32_t dim, uint32_t ld,
                    uint64_t seed, int normalize, int threads);
void orc_first_touch(float *p, uint64_t n_floats, int threads);
void orc_synth_clustered_rows(float *rows, uint64_t first, uint64_t count, uint32_t dim, uint32_t ld,
                              uint64_t seed, uint32_t n_clusters, float noise, int threads);
/* ---- product quantisation (pq.rs; per-sub-space codebooks, u8 codes: BASELINE config 5) ---- */
/* random_centroids (pq.rs:261-285) per sub-space + Quantizer::quantize (pq.rs:61-71, exact
 * nearest centroid) for every row; codes [n][m], codebook [m][ksub][dim/m] */
int orc_pq_create_kmeans(const float *rows, uint64_t n, uint32_t dim, uint32_t ld, uint32_t m, uint32_t ksub,
                         uint64_t seed, uint32_t kmeans_iters, uint64_t sample, uint8_t *codes, float *codebook,
                         int threads);
int orc_pq_create(const float *rows, uint64_t n, uint32_t dim, uint32_t ld, uint32_t m, uint32_t ksub,
                  uint64_t seed, uint8_t *codes, float *codebook, int threads);
void orc_pq_encode(const float *rows, uint64_t n, uint32_t ld, uint32_t m, uint32_t ksub, uint32_t dsub,
                   const float *codebook, uint8_t *codes, int threads);
/* turn the index's store into a PQ store (the arrays must outlive the index) */
void orc_index_set_pq(orc_index *ix, const uint8_t *codes, const float *codebook, uint32_t m, uint32_t ksub,
                      uint32_t dsub);
void orc_index_set_pq_table_f16(orc_index *ix, int on);
/* QuantizedHnsw::search pq.rs:346-364: search the code graph, re-rank with the full store
 * (sum_mode of `full`), sort (d, id) */
int orc_pq_search_batch(const orc_index *ix, const orc_store *full, const float *queries, uint32_t ldq,
                        uint64_t nq, orc_search_params sp, int quantize_query, uint64_t *out_ids, float *out_d,
                        uint64_t *out_len, orc_stats *st, int threads);
/* format-preserving permutation of [0,domain) used by the neighbour seeding step */
uint64_t orc_feistel_perm(uint64_t i, uint64_t domain, uint64_t key);

#ifdef __cplusplus
}
#endif
#endif
