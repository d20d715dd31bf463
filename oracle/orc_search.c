/* ORACLE (test infrastructure): Layer accessors, closest_nodes, closest_vectors,
 * search_layers, knn, threshold_nn restated from src/lib.rs:85-277,905-962 and
 * src/search.rs:9-140. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orc_internal.h"

/* ---------------------------------------------------------------- index handle */

orc_index *orc_index_new(const float *rows, uint64_t n, uint32_t dim, uint32_t ld, int metric,
                         int sum_mode) {
  if (ld < dim || (ld % 4) != 0) return NULL;
  orc_index *ix = (orc_index *)calloc(1, sizeof(orc_index));
  ix->store.rows = rows;
  ix->store.n = n;
  ix->store.dim = dim;
  ix->store.ld = ld;
  ix->store.metric = metric;
  ix->store.sum_mode = sum_mode;
  ix->store.codes = NULL;
  ix->store.codebook = NULL;
  ix->store.pq_table_f16 = 0;
  return ix;
}

void orc_index_free(orc_index *ix) {
  if (!ix) return;
  for (uint32_t i = 0; i < ix->layer_count; i++) {
    free(ix->layers[i].nodes);
    free(ix->layers[i].neighbors);
  }
  free(ix->layers);
  orc_pending_free(ix);
  free(ix);
}

void orc_index_set_sum_mode(orc_index *ix, int sum_mode) { ix->store.sum_mode = sum_mode; }

int orc_index_push_layer(orc_index *ix, const uint64_t *nodes, const uint64_t *neighbors,
                         uint64_t node_count, uint64_t neighborhood_size) {
  ix->layers = (orc_layer *)realloc(ix->layers, sizeof(orc_layer) * (ix->layer_count + 1));
  orc_layer *L = &ix->layers[ix->layer_count++];
  L->node_count = node_count;
  L->neighborhood_size = neighborhood_size;
  L->nodes = (uint64_t *)malloc(sizeof(uint64_t) * (node_count ? node_count : 1));
  L->neighbors =
      (uint64_t *)malloc(sizeof(uint64_t) * (node_count * neighborhood_size + 1));
  memcpy(L->nodes, nodes, sizeof(uint64_t) * node_count);
  memcpy(L->neighbors, neighbors, sizeof(uint64_t) * node_count * neighborhood_size);
  return 0;
}

uint32_t orc_index_layer_count(const orc_index *ix) { return ix->layer_count; }
const orc_layer *orc_index_layer(const orc_index *ix, uint32_t i) {
  return i < ix->layer_count ? &ix->layers[i] : NULL;
}
const orc_store *orc_index_store(const orc_index *ix) { return &ix->store; }

/* ---------------------------------------------------------------- Layer accessors */

/* get_final_neighbor_idx  src/lib.rs:108-119 */
uint64_t orc_final_neighbor_idx(uint64_t W, const uint64_t *neighbors, uint64_t n) {
  uint64_t final_idx = W * (n + 1);
  uint64_t current_idx = final_idx;
  for (uint64_t offset = 1; offset < W + 1; offset++) {
    if (neighbors[final_idx - offset] == ORC_EMPTY)
      current_idx--;
    else
      break;
  }
  return current_idx;
}

/* Layer::get_node  src/lib.rs:129-131 : nodes.binary_search(&v).ok() */
uint64_t orc_layer_get_node(const orc_layer *L, uint64_t v) {
  uint64_t lo = 0, hi = L->node_count;
  while (lo < hi) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (L->nodes[mid] == v) return mid;
    if (L->nodes[mid] < v)
      lo = mid + 1;
    else
      hi = mid;
  }
  return ORC_EMPTY;
}

/* ---------------------------------------------------------------- scratch */

orc_scratch *orc_scratch_new(const orc_index *ix, uint64_t extra_nodes) {
  orc_scratch *sc = (orc_scratch *)calloc(1, sizeof(orc_scratch));
  uint64_t mx = extra_nodes;
  for (uint32_t i = 0; i < ix->layer_count; i++)
    if (ix->layers[i].node_count > mx) mx = ix->layers[i].node_count;
  sc->visited_cap = mx + 1;
  sc->visited = (uint32_t *)calloc(sc->visited_cap, sizeof(uint32_t));
  sc->epoch = 0;
  return sc;
}

void orc_scratch_free(orc_scratch *sc) {
  if (!sc) return;
  free(sc->visited);
  free(sc->heap);
  free(sc->batch_ids);
  free(sc->batch_d);
  free(sc->q_ids);
  free(sc->q_d);
  free(sc->c_ids);
  free(sc->c_d);
  free(sc->p_ids);
  free(sc->p_d);
  free(sc->pq_table);
  free(sc->pq_recon);
  free(sc);
}

static void ensure_pairs(uint64_t **ids, float **d, uint64_t *cap, uint64_t need) {
  if (*cap >= need) return;
  uint64_t nc = need * 2 + 16;
  *ids = (uint64_t *)realloc(*ids, sizeof(uint64_t) * nc);
  *d = (float *)realloc(*d, sizeof(float) * nc);
  *cap = nc;
}

/* visit_queue: the reference keeps a Vec re-sorted every hop by (-d, MAX - id) and pops
 * the tail (src/lib.rs:191,243-244), i.e. it always pops the smallest (d, id); among equal
 * keys (only possible when a neighbour row holds a duplicate id) the stable sort leaves the
 * later-pushed entry at the tail.  A binary min-heap on (d, id, -seq) pops the identical
 * sequence without the O(F log F) re-sort. */
static int vq_less(const orc_vq_entry *a, const orc_vq_entry *b) {
  if (a->d != b->d) return a->d < b->d;
  if (a->id != b->id) return a->id < b->id;
  return a->seq > b->seq;
}

static void vq_push(orc_scratch *sc, orc_vq_entry e) {
  if (sc->heap_len == sc->heap_cap) {
    sc->heap_cap = sc->heap_cap * 2 + 64;
    sc->heap = (orc_vq_entry *)realloc(sc->heap, sizeof(orc_vq_entry) * sc->heap_cap);
  }
  uint64_t c = sc->heap_len++;
  sc->heap[c] = e;
  while (c > 0) {
    uint64_t p = (c - 1) / 2;
    if (vq_less(&sc->heap[c], &sc->heap[p])) {
      orc_vq_entry t = sc->heap[c];
      sc->heap[c] = sc->heap[p];
      sc->heap[p] = t;
      c = p;
    } else
      break;
  }
}

static orc_vq_entry vq_pop(orc_scratch *sc) {
  orc_vq_entry top = sc->heap[0];
  sc->heap[0] = sc->heap[--sc->heap_len];
  uint64_t c = 0;
  for (;;) {
    uint64_t l = 2 * c + 1, r = l + 1, b = c;
    if (l < sc->heap_len && vq_less(&sc->heap[l], &sc->heap[b])) b = l;
    if (r < sc->heap_len && vq_less(&sc->heap[r], &sc->heap[b])) b = r;
    if (b == c) break;
    orc_vq_entry t = sc->heap[b];
    sc->heap[b] = sc->heap[c];
    sc->heap[c] = t;
    c = b;
  }
  return top;
}

/* stable insertion sort by (OrderedFloat(d), id)  src/lib.rs:206 ; batches are <= W long */
static void sort_pairs(uint64_t *ids, float *d, uint64_t m) {
  for (uint64_t i = 1; i < m; i++) {
    uint64_t id = ids[i];
    float di = d[i];
    uint64_t j = i;
    while (j > 0 && (d[j - 1] > di || (d[j - 1] == di && ids[j - 1] > id))) {
      ids[j] = ids[j - 1];
      d[j] = d[j - 1];
      j--;
    }
    ids[j] = id;
    d[j] = di;
  }
}

/* ---------------------------------------------------------------- closest_nodes */

/* Layer::closest_nodes  src/lib.rs:175-248.  qv = the query vector (lookup_abstract of
 * Stored/Unstored already resolved, src/lib.rs:60-73). */
uint64_t orc_closest_nodes(const orc_index *ix, const orc_layer *L, orc_pq *cand, uint64_t probe_depth,
                           orc_scratch *sc, orc_stats *st) {
  const orc_store *S = &ix->store;
  /* assert!(!candidates.is_empty())  :175 */
  sc->epoch++;
  if (sc->epoch == 0) { /* wrapped */
    memset(sc->visited, 0, sizeof(uint32_t) * sc->visited_cap);
    sc->epoch = 1;
  }
  sc->heap_len = 0;
  uint64_t seq = 0;
  uint64_t ninit = orc_pq_iter_len(cand);
  /* visit_queue = candidates.iter().collect().reverse(); visited = candidates ids  :182-187 */
  for (uint64_t i = 0; i < ninit; i++) {
    orc_vq_entry e = {cand->prio[i], cand->data[i], ninit - i, 0, 0};
    vq_push(sc, e);
    sc->visited[cand->data[i]] = sc->epoch;
  }
  seq = ninit + 1;
  uint64_t highest_improvement = 0;
  uint64_t W = L->neighborhood_size;
  ensure_pairs(&sc->batch_ids, &sc->batch_d, &sc->batch_cap, W);
  while (sc->heap_len) { /* while let Some(..) = visit_queue.pop()  :191 */
    orc_vq_entry cur = vq_pop(sc);
    if (st) st->n_hops++;
    uint64_t first = W * cur.id;
    uint64_t final = orc_final_neighbor_idx(W, L->neighbors, cur.id); /* get_neighbors :195 */
    uint64_t m = 0;
    for (uint64_t k = first; k < final; k++) {
      uint64_t n = L->neighbors[k];
      if (sc->visited[n] == sc->epoch) continue; /* filter(!visited.contains) :198 */
      /* compare_vec(v, Stored(get_vector(n)))  :200-202 */
      float d = orc_query_dist(S, sc, L->nodes[n]);
      if (st) st->n_dist++;
      sc->batch_ids[m] = n;
      sc->batch_d[m] = d;
      m++;
    }
    sort_pairs(sc->batch_ids, sc->batch_d, m);                              /* :206 */
    for (uint64_t k = 0; k < m; k++) sc->visited[sc->batch_ids[k]] = sc->epoch; /* :209 */
    for (uint64_t k = 0; k < m; k++) {                                        /* :211-220 */
      orc_vq_entry e = {sc->batch_d[k], sc->batch_ids[k], seq++, cur.hops + 1,
                        cur.index_sum + k + 1};
      vq_push(sc, e);
    }
    /* current_best = candidates.first()  :225 */
    int had = orc_pq_len(cand) != 0;
    uint64_t best_id = had ? cand->data[0] : 0;
    float best_d = had ? cand->prio[0] : 0.0f;
    int did = orc_pq_merge(cand, sc->batch_ids, sc->batch_d, m); /* :226 */
    int has = orc_pq_len(cand) != 0;
    if (had != has || (has && (cand->data[0] != best_id || cand->prio[0] != best_d)))
      highest_improvement = cur.index_sum; /* :227-230 */
    if (!did) {                            /* :233-238 */
      probe_depth -= 1;
      if (probe_depth == 0) break;
    }
  }
  return highest_improvement;
}

/* ---------------------------------------------------------------- closest_vectors */

/* Layer::closest_vectors  src/lib.rs:250-277.  cand holds VectorIds; result pairs are
 * written to sc->p_* (VectorIds), returns count or -1 when get_node().unwrap() would panic */
static int64_t closest_vectors(const orc_index *ix, const orc_layer *L, const orc_pq *cand, uint64_t candidate_count, uint64_t probe_depth,
                               uint64_t exclude, orc_scratch *sc, orc_stats *st,
                               uint64_t *index_distance) {
  uint64_t cap = cand->cap; /* PriorityQueue::new(candidates.capacity())  :264 */
  ensure_pairs(&sc->q_ids, &sc->q_d, &sc->q_cap, cap);
  ensure_pairs(&sc->p_ids, &sc->p_d, &sc->p_cap, cap);
  uint64_t np = orc_pq_iter_len(cand);
  for (uint64_t i = 0; i < np; i++) { /* :258-262 */
    uint64_t node = orc_layer_get_node(L, cand->data[i]);
    if (node == ORC_EMPTY) return -1;
    sc->p_ids[i] = node;
    sc->p_d[i] = cand->prio[i];
  }
  orc_pq queue = {sc->q_ids, sc->q_d, cap};
  for (uint64_t i = 0; i < cap; i++) {
    queue.data[i] = ORC_EMPTY;
    queue.prio[i] = ORC_FMAX;
  }
  orc_pq_merge(&queue, sc->p_ids, sc->p_d, np);                                 /* :266 */
  *index_distance = orc_closest_nodes(ix, L, &queue, probe_depth, sc, st); /* :267 */
  uint64_t nq = orc_pq_iter_len(&queue);
  uint64_t out = 0;
  for (uint64_t i = 0; i < nq && out < candidate_count; i++) { /* :269-275 */
    uint64_t v = L->nodes[queue.data[i]];
    if (exclude != ORC_EMPTY && v == exclude) continue; /* include = |v| Some(v) != exclude */
    sc->p_ids[out] = v;
    sc->p_d[out] = queue.prio[i];
    out++;
  }
  return (int64_t)out;
}

/* ---------------------------------------------------------------- search_layers */

/* search_layers_instrumented  src/search.rs:93-140 */
int orc_search_sc(const orc_index *ix, const float *query, uint64_t qid, orc_search_params sp,
                  uint32_t upto_layers, uint64_t exclude, uint64_t *out_ids, float *out_d,
                  uint64_t *out_len, orc_stats *st, orc_scratch *sc, uint64_t *index_distance) {
  const orc_store *S = &ix->store;
  uint32_t nl = (upto_layers == 0 || upto_layers > ix->layer_count) ? ix->layer_count : upto_layers;
  if (nl == 0 || sp.number_of_candidates == 0 || sp.probe_depth == 0) return -3;
  orc_query_prepare(S, sc, query, qid);
  const orc_layer *layers = ix->layers;
  uint64_t cap = sp.number_of_candidates;
  ensure_pairs(&sc->c_ids, &sc->c_d, &sc->c_cap, cap);
  orc_pq cand = {sc->c_ids, sc->c_d, cap}; /* PriorityQueue::new(number_of_candidates) :110 */
  for (uint64_t i = 0; i < cap; i++) {
    cand.data[i] = ORC_EMPTY;
    cand.prio[i] = ORC_FMAX;
  }
  uint64_t entry = layers[0].nodes[0]; /* entry_vector  src/search.rs:9-11 */
  float d0 = orc_query_dist(S, sc, entry); /* :102-109 */
  if (st) st->n_dist++;
  orc_pq_insert(&cand, entry, d0); /* :111 */
  uint64_t last_index_distance = UINT64_MAX;
  for (uint32_t i = 0; i < nl; i++) { /* :113 */
    uint64_t candidate_count = (nl == 1 || i == nl - 1) ? sp.number_of_candidates
                                                        : sp.upper_layer_candidate_count; /* :122-126 */
    int64_t n = closest_vectors(ix, &layers[i], &cand, candidate_count, sp.probe_depth, exclude,
                                sc, st, &last_index_distance); /* :128-134 */
    if (n < 0) return -2;
    orc_pq_merge(&cand, sc->p_ids, sc->p_d, (uint64_t)n); /* :136 */
  }
  uint64_t len = orc_pq_iter_len(&cand); /* candidates.iter().collect()  :139 */
  for (uint64_t i = 0; i < len; i++) {
    out_ids[i] = cand.data[i];
    out_d[i] = cand.prio[i];
  }
  for (uint64_t i = len; i < cap; i++) {
    out_ids[i] = ORC_EMPTY;
    out_d[i] = ORC_FMAX;
  }
  *out_len = len;
  if (index_distance) *index_distance = last_index_distance;
  return 0;
}

int orc_search(const orc_index *ix, const float *query, uint64_t qid, orc_search_params sp,
               uint32_t upto_layers, uint64_t exclude, uint64_t *out_ids, float *out_d,
               uint64_t *out_len, orc_stats *st) {
  orc_scratch *sc = orc_scratch_new(ix, 0);
  int rc = orc_search_sc(ix, query, qid, sp, upto_layers, exclude, out_ids, out_d, out_len, st, sc, NULL);
  orc_scratch_free(sc);
  return rc;
}

/* Hnsw::search_instrumented  src/lib.rs:667-673 for a batch: the second return value of
 * search_layers_instrumented (src/search.rs:93-140) -- the index_sum of the last hop of the bottom layer's
 * closest_nodes that changed the best candidate (src/lib.rs:211-231) -- lands in out_index_distance[q] */
int orc_search_batch_instrumented(const orc_index *ix, const float *queries, uint32_t ldq, const uint64_t *qids,
                                  uint64_t nq, orc_search_params sp, uint64_t *out_ids, float *out_d,
                                  uint64_t *out_len, uint64_t *out_index_distance, int threads) {
  int rc_all = 0;
  uint64_t cap = sp.number_of_candidates;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    orc_scratch *sc = orc_scratch_new(ix, 0);
#pragma omp for schedule(dynamic, 8)
    for (uint64_t q = 0; q < nq; q++) {
      int rc = orc_search_sc(ix, queries ? queries + q * (uint64_t)ldq : NULL, qids ? qids[q] : 0, sp, 0, ORC_EMPTY,
                             out_ids + q * cap, out_d + q * cap, out_len + q, NULL, sc, out_index_distance + q);
      if (rc) {
#pragma omp critical
        rc_all = rc;
      }
    }
    orc_scratch_free(sc);
  }
  return rc_all;
}

int orc_search_batch(const orc_index *ix, const float *queries, uint32_t ldq, const uint64_t *qids,
                     uint64_t nq, orc_search_params sp, const uint64_t *exclude, uint64_t *out_ids,
                     float *out_d, uint64_t *out_len, orc_stats *st, int threads) {
  int rc_all = 0;
  uint64_t cap = sp.number_of_candidates;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    orc_scratch *sc = orc_scratch_new(ix, 0);
#pragma omp for schedule(dynamic, 8)
    for (uint64_t q = 0; q < nq; q++) {
      orc_stats s = {0, 0};
      int rc = orc_search_sc(ix, queries ? queries + q * (uint64_t)ldq : NULL, qids ? qids[q] : 0, sp, 0,
                             exclude ? exclude[q] : ORC_EMPTY, out_ids + q * cap, out_d + q * cap,
                             out_len + q, &s, sc, NULL);
      if (st) st[q] = s;
      if (rc) {
#pragma omp critical
        rc_all = rc;
      }
    }
    orc_scratch_free(sc);
  }
  return rc_all;
}

/* ---------------------------------------------------------------- knn / threshold_nn */

/* Hnsw::knn  src/lib.rs:905-928 */
int orc_knn(const orc_index *ix, uint64_t k, uint64_t probe_depth, uint64_t *out_ids, float *out_d,
            uint64_t *out_len, int threads) {
  if (ix->layer_count == 0 || k == 0 || probe_depth == 0) return -3;
  const orc_layer *L = &ix->layers[ix->layer_count - 1];
  const orc_store *S = &ix->store;
  uint64_t cap = k * 3; /* eff_factor = 3 */
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    orc_scratch *sc = orc_scratch_new(ix, 0);
    uint64_t *ids = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    float *pr = (float *)malloc(sizeof(float) * cap);
#pragma omp for schedule(dynamic, 8)
    for (uint64_t i = 0; i < L->node_count; i++) {
      orc_pq pq = {ids, pr, cap};
      for (uint64_t j = 0; j < cap; j++) {
        ids[j] = ORC_EMPTY;
        pr[j] = ORC_FMAX;
      }
      uint64_t self = i;
      float zero = 0.0f;
      orc_pq_merge(&pq, &self, &zero, 1); /* pq.merge_pairs(&[(node, 0.0)]) */
      orc_query_prepare(S, sc, NULL, L->nodes[i]);
      orc_closest_nodes(ix, L, &pq, probe_depth, sc, NULL);
      uint64_t n = orc_pq_iter_len(&pq), out = 0;
      for (uint64_t j = 0; j < n && out < k; j++) {
        if (ids[j] == self) continue;
        out_ids[i * k + out] = L->nodes[ids[j]];
        out_d[i * k + out] = pr[j];
        out++;
      }
      out_len[i] = out;
      for (uint64_t j = out; j < k; j++) {
        out_ids[i * k + j] = ORC_EMPTY;
        out_d[i * k + j] = ORC_FMAX;
      }
    }
    free(ids);
    free(pr);
    orc_scratch_free(sc);
  }
  return 0;
}

/* Hnsw::threshold_nn  src/lib.rs:930-962 */
int orc_threshold_nn(const orc_index *ix, float threshold, uint64_t probe_depth,
                     uint64_t initial_search_depth, uint64_t max_out, uint64_t *out_ids, float *out_d,
                     uint64_t *out_len, int threads) {
  if (ix->layer_count == 0 || initial_search_depth == 0 || probe_depth == 0) return -3;
  const orc_layer *L = &ix->layers[ix->layer_count - 1];
  const orc_store *S = &ix->store;
  int overflow = 0;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    orc_scratch *sc = orc_scratch_new(ix, 0);
#pragma omp for schedule(dynamic, 8)
    for (uint64_t i = 0; i < L->node_count; i++) {
      uint64_t cap = initial_search_depth;
      uint64_t *ids = (uint64_t *)malloc(sizeof(uint64_t) * cap);
      float *pr = (float *)malloc(sizeof(float) * cap);
      for (uint64_t j = 0; j < cap; j++) {
        ids[j] = ORC_EMPTY;
        pr[j] = ORC_FMAX;
      }
      orc_pq pq = {ids, pr, cap};
      uint64_t self = i;
      float zero = 0.0f;
      orc_pq_merge(&pq, &self, &zero, 1);
      float last = 0.0f;
      uint64_t last_size = 0;
      while (last < threshold && orc_pq_len(&pq) > last_size) {
        last_size = orc_pq_len(&pq);
        orc_query_prepare(S, sc, NULL, L->nodes[i]);
        orc_closest_nodes(ix, L, &pq, probe_depth, sc, NULL);
        uint64_t len = orc_pq_len(&pq);
        last = pq.prio[len - 1]; /* pq.last().expect(..).1 */
        if (last < threshold && len == pq.cap) {
          /* resize_capacity(capacity * 2)  src/priority_queue.rs:188-197 */
          uint64_t nc = pq.cap * 2;
          ids = (uint64_t *)realloc(ids, sizeof(uint64_t) * nc);
          pr = (float *)realloc(pr, sizeof(float) * nc);
          for (uint64_t j = pq.cap; j < nc; j++) {
            ids[j] = ORC_EMPTY;
            pr[j] = ORC_FMAX;
          }
          pq.data = ids;
          pq.prio = pr;
          pq.cap = nc;
        }
      }
      uint64_t n = orc_pq_iter_len(&pq), out = 0;
      for (uint64_t j = 0; j < n; j++) {
        if (ids[j] == self) continue;         /* filter(n != node) */
        if (!(pr[j] < threshold)) break;      /* take_while(d < threshold) */
        if (out == max_out) {
          overflow = 1;
          break;
        }
        out_ids[i * max_out + out] = L->nodes[ids[j]];
        out_d[i * max_out + out] = pr[j];
        out++;
      }
      out_len[i] = out;
      free(ids);
      free(pr);
    }
    orc_scratch_free(sc);
  }
  return overflow ? -4 : 0;
}
