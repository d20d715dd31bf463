/* ORACLE (test infrastructure): private structs shared by the oracle's C files. */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "orc.h"

struct orc_index {
  orc_store store;
  uint32_t layer_count;
  orc_layer *layers; /* top first  (src/lib.rs:587) */
  /* layer under construction (phase API of orc_build.c) */
  uint64_t *p_vs, *p_gm, *p_gstart, *p_gsize;
  uint64_t p_n, p_W, p_K;
  int p_grouped;
};
void orc_pending_free(orc_index *ix);

/* visit_queue entry: (NodeId, f32, NodeDistance{hops,index_sum})  src/lib.rs:182,162-173 */
typedef struct {
  float d;
  uint64_t id;
  uint64_t seq;
  uint64_t hops;
  uint64_t index_sum;
} orc_vq_entry;

typedef struct {
  uint32_t *visited; /* HashSet<NodeId> as epoch stamps */
  uint64_t visited_cap;
  uint32_t epoch;
  orc_vq_entry *heap;
  uint64_t heap_len, heap_cap;
  uint64_t *batch_ids;
  float *batch_d;
  uint64_t batch_cap;
  uint64_t *q_ids;
  float *q_d;
  uint64_t q_cap;
  uint64_t *c_ids;
  float *c_d;
  uint64_t c_cap;
  uint64_t *p_ids;
  float *p_d;
  uint64_t p_cap;
  /* the prepared query: f32 vector, or (PQ store) its lookup table T[m][ksub] */
  const float *qv;
  float *pq_table;
  float pq_bias, pq_scale; /* 8-bit table mode: distance = bias + scale * sum of entries */
  float *pq_recon;
} orc_scratch;

/* lookup_abstract + the start of compare_vec (lib.rs:60-73): fix the query side once */
void orc_query_prepare(const orc_store *S, orc_scratch *sc, const float *raw, uint64_t stored_id);
/* compare_vec(query, Stored(vid)) */
float orc_query_dist(const orc_store *S, const orc_scratch *sc, uint64_t vid);

orc_scratch *orc_scratch_new(const orc_index *ix, uint64_t extra_nodes);
void orc_scratch_free(orc_scratch *sc);
uint64_t orc_layer_get_node(const orc_layer *L, uint64_t v);
uint64_t orc_closest_nodes(const orc_index *ix, const orc_layer *L, orc_pq *cand, uint64_t probe_depth,
                           orc_scratch *sc, orc_stats *st);
int orc_search_sc(const orc_index *ix, const float *query, uint64_t qid, orc_search_params sp,
                  uint32_t upto_layers, uint64_t exclude, uint64_t *out_ids, float *out_d,
                  uint64_t *out_len, orc_stats *st, orc_scratch *sc, uint64_t *index_distance);
#endif
