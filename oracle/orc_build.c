/* ORACLE (test infrastructure): index construction restated from src/lib.rs:675-893
 * (generate_layer, generate), 1070-1154 (link), 1463-1544 (recall, improve_neighbors),
 * 1546-1686 (improve_index, promotion excluded), 1830-1852 (choose_n_1), 1883-1960
 * (partition arithmetic) and src/search.rs:13-82 (initial partitions).
 *
 * The reference build is not reproducible (thread_rng shuffle lib.rs:832, rand's StdRng
 * streams lib.rs:729,1847, lock-order races lib.rs:797-815,1107-1153), so this is the
 * DETERMINISTIC variant the HIP build is bit-compared with:
 *   - every PRNG draw comes from the counter-based generators in orc_util.c;
 *   - every "insert into another node's row under a lock" step is evaluated against a
 *     snapshot and resolved as  row' = best-W by (distance, id) of  row U proposals,
 *     which is what the sequential reference code produces for (d,id)-sorted rows.
 * "parity unpinned" vs the Rust crate at graph level; pinned properties are recall,
 * layer invariants and the toy-index golden tests. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orc_internal.h"

void orc_default_build_params(orc_build_params *bp) {
  /* src/parameters.rs:10-64 */
  bp->order = 12;
  bp->zero_layer_neighborhood_size = 48;
  bp->neighborhood_size = 24;
  bp->optimization.promotion_threshold = 0.01f;
  bp->optimization.neighborhood_threshold = 0.01f;
  bp->optimization.recall_proportion = 0.1f;
  bp->optimization.promotion_proportion = 1.0f;
  bp->optimization.search.number_of_candidates = 300;
  bp->optimization.search.upper_layer_candidate_count = 300;
  bp->optimization.search.probe_depth = 2;
  bp->initial_partition_search.number_of_candidates = 6;
  bp->initial_partition_search.upper_layer_candidate_count = 6;
  bp->initial_partition_search.probe_depth = 2;
  bp->seed = 0;
  bp->max_link_rounds = 0;
  bp->promote = 1; /* the reference always tries promote_at_layer (lib.rs:1575-1580) */
}

/* ---------------------------------------------------------------- partitions */

/* calculate_partitions_from_bottom  src/lib.rs:1883-1893 (f32 arithmetic as written) */
static uint32_t partitions_from_bottom(uint64_t total, uint64_t order, uint64_t *out, uint32_t max_out) {
  float lc = ceilf(logf((float)total) / logf((float)order));
  uint64_t layer_count = (lc != lc || lc < 0.0f) ? 0 : (uint64_t)lc; /* `as usize` saturates */
  if (layer_count < 1) layer_count = 1;
  uint64_t size = total;
  uint32_t n = 0;
  for (uint64_t i = 0; i < layer_count && n < max_out; i++) {
    out[n++] = size;
    size /= order;
  }
  return n;
}

/* calculate_partitions  src/lib.rs:1895-1899 */
uint32_t orc_calculate_partitions(uint64_t total, uint64_t order, uint64_t *out, uint32_t max_out) {
  uint32_t n = partitions_from_bottom(total, order, out, max_out);
  for (uint32_t i = 0; i < n / 2; i++) {
    uint64_t t = out[i];
    out[i] = out[n - 1 - i];
    out[n - 1 - i] = t;
  }
  return n;
}

/* calculate_partitions_for_additions  src/lib.rs:1901-1960 */
uint32_t orc_calculate_partitions_for_additions(const uint64_t *sizes, uint32_t n_sizes,
                                                uint64_t new_vecs, uint64_t order, uint64_t *out,
                                                uint32_t max_out) {
  uint64_t tmp[128];
  uint32_t n = partitions_from_bottom(sizes[0] + new_vecs, order, tmp, 128);
  while (n < n_sizes) tmp[n++] = 0;
  for (uint32_t i = 0; i < n; i++)
    if (i < n_sizes && sizes[i] > tmp[i]) tmp[i] = sizes[i];
  uint64_t last = 0;
  for (uint32_t i = n; i-- > 0;) {
    if (last > tmp[i]) tmp[i] = last;
    last = tmp[i];
  }
  for (uint32_t i = 0; i < n; i++)
    if (i < n_sizes) tmp[i] -= sizes[i];
  last = 0;
  for (uint32_t i = n; i-- > 0;) {
    if (last > tmp[i]) tmp[i] = last;
    last = tmp[i];
  }
  uint32_t m = n < max_out ? n : max_out;
  memcpy(out, tmp, sizeof(uint64_t) * m);
  return m;
}

/* ---------------------------------------------------------------- helpers */

typedef struct {
  float d;
  uint64_t id;
} nd_pair;

static int nd_cmp(const void *a, const void *b) {
  const nd_pair *x = (const nd_pair *)a, *y = (const nd_pair *)b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  if (x->id < y->id) return -1;
  if (x->id > y->id) return 1;
  return 0;
}

static int u64_cmp(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* sort (d,id), dedup(), drop `self`, take W, pad  (src/lib.rs:757-764) */
static void finish_row(nd_pair *list, uint64_t m, uint64_t self, uint64_t W, uint64_t *row_ids,
                       float *row_d) {
  qsort(list, m, sizeof(nd_pair), nd_cmp);
  uint64_t out = 0;
  for (uint64_t k = 0; k < m && out < W; k++) {
    if (k > 0 && list[k].id == list[k - 1].id && list[k].d == list[k - 1].d) continue; /* dedup */
    if (list[k].id == self) continue;
    row_ids[out] = list[k].id;
    row_d[out] = list[k].d;
    out++;
  }
  for (; out < W; out++) {
    row_ids[out] = ORC_EMPTY;
    row_d[out] = ORC_FMAX;
  }
}

/* row'[t] = best-W by (d,id) of row[t] U incoming[t]; incoming built from `prop_*`
 * (target, source, d) triples.  Deterministic form of the RwLock'd insert passes
 * (src/lib.rs:797-815 and 1118-1147). Returns the number of entries that are new. */
static uint64_t merge_proposals(uint64_t n, uint64_t W, uint64_t *rows, float *rows_d,
                                const uint64_t *prop_t, const uint64_t *prop_s, const float *prop_d,
                                uint64_t np, int threads) {
  uint64_t *start = (uint64_t *)calloc(n + 1, sizeof(uint64_t));
  for (uint64_t k = 0; k < np; k++) start[prop_t[k] + 1]++;
  for (uint64_t t = 0; t < n; t++) start[t + 1] += start[t];
  uint64_t *fill = (uint64_t *)malloc(sizeof(uint64_t) * (n + 1));
  memcpy(fill, start, sizeof(uint64_t) * (n + 1));
  nd_pair *inc = (nd_pair *)malloc(sizeof(nd_pair) * (np ? np : 1));
  for (uint64_t k = 0; k < np; k++) {
    nd_pair p = {prop_d[k], prop_s[k]};
    inc[fill[prop_t[k]]++] = p;
  }
  free(fill);
  uint64_t added = 0;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 64) reduction(+ : added)
  for (uint64_t t = 0; t < n; t++) {
    uint64_t cnt = start[t + 1] - start[t];
    if (cnt == 0) continue;
    nd_pair *list = (nd_pair *)malloc(sizeof(nd_pair) * (W + cnt));
    uint64_t m = 0;
    uint64_t *old = (uint64_t *)malloc(sizeof(uint64_t) * W);
    uint64_t nold = 0;
    for (uint64_t k = 0; k < W; k++) {
      if (rows[t * W + k] == ORC_EMPTY) continue;
      nd_pair p = {rows_d[t * W + k], rows[t * W + k]};
      list[m++] = p;
      old[nold++] = p.id;
    }
    for (uint64_t k = 0; k < cnt; k++) list[m++] = inc[start[t] + k];
    finish_row(list, m, t, W, rows + t * W, rows_d + t * W);
    for (uint64_t k = 0; k < W; k++) {
      uint64_t id = rows[t * W + k];
      if (id == ORC_EMPTY) break;
      int found = 0;
      for (uint64_t j = 0; j < nold; j++)
        if (old[j] == id) {
          found = 1;
          break;
        }
      if (!found) added++;
    }
    free(old);
    free(list);
  }
  free(inc);
  free(start);
  return added;
}

typedef struct {
  uint64_t key; /* ORC_EMPTY = None */
  float d;
  uint64_t node;
} gm_t;

static int gm_cmp(const void *a, const void *b) {
  const gm_t *x = (const gm_t *)a, *y = (const gm_t *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  if (x->d != y->d) return x->d < y->d ? -1 : 1;
  return x->node < y->node ? -1 : (x->node > y->node ? 1 : 0);
}

/* ---------------------------------------------------------------- generate_layer */

/* generate_layer runs in four phases (begin / init_search(range) / seed(range) / finish) so
 * that tests can drive the same node-range sharding the multi-GPU build uses; the
 * monolithic orc_generate_layer below is the phases over the whole range. */

static void pending_free(orc_index *ix) {
  free(ix->p_vs);
  free(ix->p_gm);
  free(ix->p_gstart);
  free(ix->p_gsize);
  ix->p_vs = NULL;
  ix->p_gm = NULL;
  ix->p_gstart = ix->p_gsize = NULL;
  ix->p_n = 0;
  ix->p_grouped = 0;
}
void orc_pending_free(orc_index *ix) { pending_free(ix); }

/* vs.sort(), allocate  src/lib.rs:683-693 */
int orc_layer_begin(orc_index *ix, const uint64_t *vs_in, uint64_t n, uint64_t W, const orc_build_params *bp) {
  if (n == 0 || W == 0) return -3; /* assert!(!vs.is_empty()) :683 */
  pending_free(ix);
  ix->p_vs = (uint64_t *)malloc(sizeof(uint64_t) * n);
  memcpy(ix->p_vs, vs_in, sizeof(uint64_t) * n);
  qsort(ix->p_vs, n, sizeof(uint64_t), u64_cmp); /* vs.sort() :685 */
  ix->p_n = n;
  ix->p_W = W;
  ix->p_K = ix->layer_count == 0 ? (n > 1 ? n - 1 : 1) : bp->initial_partition_search.number_of_candidates;
  return 0;
}
uint64_t orc_layer_init_stride(const orc_index *ix) { return ix->p_K; }

/* 1. generate_initial_partitions for nodes [first, first+count)  src/search.rs:32-71;
 * out_ids/out_d are [count][K], out_len [count] */
int orc_layer_init_search(orc_index *ix, const orc_build_params *bp, uint64_t first, uint64_t count,
                          uint64_t *out_ids, float *out_d, uint64_t *out_len, int threads) {
  const orc_store *S = &ix->store;
  const uint64_t *vs = ix->p_vs;
  uint64_t n = ix->p_n, K = ix->p_K;
  if (!vs || first + count > n) return -3;
  uint32_t layer_count = ix->layer_count;
  orc_search_params ips = bp->initial_partition_search;
  int rc_all = 0;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    orc_scratch *sc = layer_count ? orc_scratch_new(ix, 0) : NULL;
    orc_scratch *sc0 = orc_scratch_new(ix, 0);
    uint64_t *oi = (uint64_t *)malloc(sizeof(uint64_t) * (ips.number_of_candidates + 1));
    float *od = (float *)malloc(sizeof(float) * (ips.number_of_candidates + 1));
    nd_pair *tmp = (nd_pair *)malloc(sizeof(nd_pair) * (K + 1));
#pragma omp for schedule(dynamic, 16)
    for (uint64_t x = 0; x < count; x++) {
      uint64_t i = first + x;
      uint64_t m = 0;
      if (layer_count == 0) {
        /* compare_all  src/search.rs:13-30 */
        orc_query_prepare(S, sc0, NULL, vs[i]);
        for (uint64_t j = 0; j < n; j++) {
          if (vs[j] == vs[i]) continue;
          nd_pair p = {orc_query_dist(S, sc0, vs[j]), j};
          tmp[m++] = p;
        }
        qsort(tmp, m, sizeof(nd_pair), nd_cmp); /* ids are node ids; monotone in vector id */
      } else {
        /* initial_vector_distances  src/search.rs:73-82 */
        uint64_t len = 0;
        int rc = orc_search_sc(ix, NULL, vs[i], ips, 0, ORC_EMPTY, oi, od, &len, NULL, sc, NULL);
        if (rc) {
#pragma omp atomic write
          rc_all = rc;
          len = 0;
        }
        for (uint64_t k = 0; k < len; k++) {
          if (oi[k] == vs[i]) continue; /* filter(|(w,_)| v != *w) */
          /* NodeId(vs.binary_search(&inner_vector_id).unwrap())  src/search.rs:57-60 */
          uint64_t *hit = (uint64_t *)bsearch(&oi[k], vs, n, sizeof(uint64_t), u64_cmp);
          if (!hit) {
#pragma omp atomic write
            rc_all = -2;
            continue;
          }
          nd_pair p = {od[k], (uint64_t)(hit - vs)};
          tmp[m++] = p;
        }
      }
      for (uint64_t k = 0; k < m; k++) {
        out_ids[x * K + k] = tmp[k].id;
        out_d[x * K + k] = tmp[k].d;
      }
      for (uint64_t k = m; k < K; k++) {
        out_ids[x * K + k] = ORC_EMPTY;
        out_d[x * K + k] = ORC_FMAX;
      }
      out_len[x] = m;
    }
    free(oi);
    free(od);
    free(tmp);
    orc_scratch_free(sc);
    orc_scratch_free(sc0);
  }
  return rc_all;
}

/* 2. partition groups keyed by the nearest super node (src/lib.rs:711-713) from the FULL
 * init lists.  Member order = the sort of src/search.rs:67-69 (first distance; None
 * first), made total with the node id. */
static void layer_group(orc_index *ix, const uint64_t *init_ids, const float *init_d, const uint64_t *init_len) {
  uint64_t n = ix->p_n, K = ix->p_K;
  gm_t *gm = (gm_t *)malloc(sizeof(gm_t) * n);
  for (uint64_t i = 0; i < n; i++) {
    gm[i].node = i;
    if (init_len[i]) {
      gm[i].key = init_ids[i * K];
      gm[i].d = init_d[i * K];
    } else {
      gm[i].key = ORC_EMPTY;
      gm[i].d = 0.0f;
    }
  }
  qsort(gm, n, sizeof(gm_t), gm_cmp);
  ix->p_gm = (uint64_t *)malloc(sizeof(uint64_t) * n);
  ix->p_gstart = (uint64_t *)calloc(n + 1, sizeof(uint64_t)); /* per key node; slot n = None */
  ix->p_gsize = (uint64_t *)calloc(n + 1, sizeof(uint64_t));
  for (uint64_t p = 0; p < n; p++) {
    ix->p_gm[p] = gm[p].node;
    uint64_t slot = gm[p].key == ORC_EMPTY ? n : gm[p].key;
    if (ix->p_gsize[slot] == 0) ix->p_gstart[slot] = p;
    ix->p_gsize[slot]++;
  }
  free(gm);
  ix->p_grouped = 1;
}

/* 3. neighbourhood seeding for nodes [first, first+count)  src/lib.rs:719-787;
 * out_rows/out_rows_d are [count][W] */
int orc_layer_seed(orc_index *ix, const orc_build_params *bp, const uint64_t *init_ids, const float *init_d,
                   const uint64_t *init_len, uint64_t first, uint64_t count, uint64_t *out_rows, float *out_rows_d,
                   int threads) {
  const orc_store *S = &ix->store;
  const uint64_t *vs = ix->p_vs;
  uint64_t n = ix->p_n, K = ix->p_K, W = ix->p_W;
  if (!vs || first + count > n) return -3;
  if (!ix->p_grouped) layer_group(ix, init_ids, init_d, init_len);
  const uint64_t *gmem = ix->p_gm, *gstart = ix->p_gstart, *gsize = ix->p_gsize;
  uint32_t layer_count = ix->layer_count;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    uint64_t maxc = W * 5 + K + 8;
    orc_scratch *sc = orc_scratch_new(ix, 0);
    nd_pair *list = (nd_pair *)malloc(sizeof(nd_pair) * maxc);
    uint64_t *pstart = (uint64_t *)malloc(sizeof(uint64_t) * (K + 1));
    uint64_t *psize = (uint64_t *)malloc(sizeof(uint64_t) * (K + 1));
#pragma omp for schedule(dynamic, 16)
    for (uint64_t x = 0; x < count; x++) {
      uint64_t i = first + x;
      uint64_t m = 0;
      for (uint64_t k = 0; k < init_len[i]; k++) { /* distances.clone() */
        nd_pair p = {init_d[i * K + k], init_ids[i * K + k]};
        list[m++] = p;
      }
      uint64_t np = 0, total = 0;
      for (uint64_t k = 0; k < init_len[i]; k++) { /* filter_map(partition_groups.get(Some(n))) */
        uint64_t s = init_ids[i * K + k];
        if (gsize[s]) {
          pstart[np] = gstart[s];
          psize[np] = gsize[s];
          total += gsize[s];
          np++;
        }
      }
      if (np == 0) { /* partitions.push(partition) : our own group  :739-742 */
        uint64_t slot = init_len[i] ? init_ids[i * K] : n;
        pstart[0] = gstart[slot];
        psize[0] = gsize[slot];
        total = gsize[slot];
        np = 1;
      }
      uint64_t choice_count = W * 5 < total ? W * 5 : total; /* :745-746 */
      /* choose_n_1 (src/lib.rs:1830-1852; the Exp branch of choose_n :1861-1880 is
       * unreachable from here because choice_count <= sum(partition_maxes)): all
       * (partition, index) pairs except (0, exclude=node_id.0), shuffled, truncated. */
      uint64_t excl = i < psize[0] ? 1 : 0;
      uint64_t domain = total - excl;
      uint64_t picks = choice_count < domain ? choice_count : domain;
      /* StdRng::seed_from_u64(layer_count + vector_id + vs.len())  :729-731 */
      uint64_t key = orc_mix64((uint64_t)layer_count + vs[i] + n) ^ orc_mix64(bp->seed + 0x632BE59BD9B4E019ULL);
      orc_query_prepare(S, sc, NULL, vs[i]);
      for (uint64_t k = 0; k < picks; k++) {
        uint64_t f = orc_feistel_perm(k, domain, key);
        if (excl && f >= i) f += 1;
        uint64_t p = 0;
        while (f >= psize[p]) {
          f -= psize[p];
          p++;
        }
        uint64_t member = gmem[pstart[p] + f];
        /* compare_vec(Stored(vector_id), Stored(choice.1))  :750-754 */
        nd_pair c = {orc_query_dist(S, sc, vs[member]), member};
        list[m++] = c;
      }
      finish_row(list, m, i, W, out_rows + x * W, out_rows_d + x * W);
    }
    free(list);
    free(pstart);
    free(psize);
    orc_scratch_free(sc);
  }
  return 0;
}

/* 4. make neighbourhoods bidirectional  src/lib.rs:789-815 (snapshot form) + push */
int orc_layer_finish(orc_index *ix, const uint64_t *rows_in, const float *rows_d_in, int threads) {
  uint64_t n = ix->p_n, W = ix->p_W;
  if (!ix->p_vs) return -3;
  uint64_t *rows = (uint64_t *)malloc(sizeof(uint64_t) * n * W);
  float *rows_d = (float *)malloc(sizeof(float) * n * W);
  memcpy(rows, rows_in, sizeof(uint64_t) * n * W);
  memcpy(rows_d, rows_d_in, sizeof(float) * n * W);
  uint64_t np = 0;
  for (uint64_t i = 0; i < n * W; i++)
    if (rows[i] != ORC_EMPTY) np++;
  uint64_t *pt = (uint64_t *)malloc(sizeof(uint64_t) * (np ? np : 1));
  uint64_t *ps = (uint64_t *)malloc(sizeof(uint64_t) * (np ? np : 1));
  float *pd = (float *)malloc(sizeof(float) * (np ? np : 1));
  uint64_t c = 0;
  for (uint64_t i = 0; i < n; i++)
    for (uint64_t k = 0; k < W; k++)
      if (rows[i * W + k] != ORC_EMPTY) {
        pt[c] = rows[i * W + k];
        ps[c] = i;
        pd[c] = rows_d[i * W + k];
        c++;
      }
  merge_proposals(n, W, rows, rows_d, pt, ps, pd, np, threads > 0 ? threads : 1);
  free(pt); free(ps); free(pd);
  orc_index_push_layer(ix, ix->p_vs, rows, n, W);
  free(rows); free(rows_d);
  pending_free(ix);
  return 0;
}

/* Hnsw::generate_layer  src/lib.rs:675-823 (new_top = false) */
int orc_generate_layer(orc_index *ix, const uint64_t *vs_in, uint64_t n, uint64_t W,
                       const orc_build_params *bp, int threads) {
  int rc = orc_layer_begin(ix, vs_in, n, W, bp);
  if (rc) return rc;
  uint64_t K = ix->p_K;
  uint64_t *init_ids = (uint64_t *)malloc(sizeof(uint64_t) * n * K);
  float *init_d = (float *)malloc(sizeof(float) * n * K);
  uint64_t *init_len = (uint64_t *)malloc(sizeof(uint64_t) * n);
  uint64_t *rows = (uint64_t *)malloc(sizeof(uint64_t) * n * W);
  float *rows_d = (float *)malloc(sizeof(float) * n * W);
  rc = orc_layer_init_search(ix, bp, 0, n, init_ids, init_d, init_len, threads);
  if (!rc) rc = orc_layer_seed(ix, bp, init_ids, init_d, init_len, 0, n, rows, rows_d, threads);
  if (!rc) rc = orc_layer_finish(ix, rows, rows_d, threads);
  if (rc) pending_free(ix);
  free(init_ids); free(init_d); free(init_len); free(rows); free(rows_d);
  return rc;
}

/* ---------------------------------------------------------------- link round */

/* link round phase 1: searches of nodes [first, first+count) against the unmodified layer
 * (pseudo_layer = clone :1097-1100); out_ids [count][link_count] VectorIds */
int orc_link_search(orc_index *ix, uint32_t lft, orc_search_params sp, uint64_t link_count, uint64_t first,
                    uint64_t count, uint64_t *out_ids, float *out_d, uint64_t *out_len, int threads) {
  orc_layer *L = &ix->layers[lft];
  if (first + count > L->node_count) return -3;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    orc_scratch *sc = orc_scratch_new(ix, 0);
    uint64_t cap = sp.number_of_candidates;
    uint64_t *oi = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    float *od = (float *)malloc(sizeof(float) * cap);
#pragma omp for schedule(dynamic, 16)
    for (uint64_t x = 0; x < count; x++) {
      uint64_t vector = L->nodes[first + x];
      uint64_t len = 0;
      /* search_layers(Stored(vector), sp, &pseudo_stack, Some(vector))  :1112-1117 */
      orc_search_sc(ix, NULL, vector, sp, lft + 1, vector, oi, od, &len, NULL, sc, NULL);
      for (uint64_t k = 0; k < link_count; k++) {
        out_ids[x * link_count + k] = k < len ? oi[k] : ORC_EMPTY;
        out_d[x * link_count + k] = k < len ? od[k] : ORC_FMAX;
      }
      out_len[x] = len;
    }
    free(oi);
    free(od);
    orc_scratch_free(sc);
  }
  return 0;
}

/* link round phase 2: proposals -> rows  :1118-1147 */
uint64_t orc_link_apply(orc_index *ix, uint32_t lft, uint64_t link_count, const uint64_t *res_ids,
                        const float *res_d, const uint64_t *res_len, int threads) {
  const orc_store *S = &ix->store;
  orc_layer *L = &ix->layers[lft];
  uint64_t n = L->node_count, W = L->neighborhood_size;
  int T = threads > 0 ? threads : 1;
  /* distances of the current occupants to the row owner, recomputed as :1128-1133 does */
  float *rows_d = (float *)malloc(sizeof(float) * n * W);
#pragma omp parallel num_threads(T)
  {
    orc_scratch *sc = orc_scratch_new(ix, 0);
#pragma omp for schedule(dynamic, 64)
    for (uint64_t t = 0; t < n; t++) {
      orc_query_prepare(S, sc, NULL, L->nodes[t]); /* the metrics are symmetric bit for bit */
      for (uint64_t k = 0; k < W; k++) {
        uint64_t o = L->neighbors[t * W + k];
        rows_d[t * W + k] = o == ORC_EMPTY ? ORC_FMAX : orc_query_dist(S, sc, L->nodes[o]);
      }
    }
    orc_scratch_free(sc);
  }
  uint64_t *ft = (uint64_t *)malloc(sizeof(uint64_t) * (n * link_count + 1));
  uint64_t *fs = (uint64_t *)malloc(sizeof(uint64_t) * (n * link_count + 1));
  float *fd = (float *)malloc(sizeof(float) * (n * link_count + 1));
  uint64_t c = 0;
  for (uint64_t i = 0; i < n; i++) {
    uint64_t vector = L->nodes[i];
    for (uint64_t k = 0; k < res_len[i] && k < link_count; k++) { /* take(neighborhood_size) */
      uint64_t w = res_ids[i * link_count + k];
      if (w == vector) break; /* :1119-1121 */
      ft[c] = orc_layer_get_node(L, w);
      fs[c] = i;
      fd[c] = res_d[i * link_count + k];
      c++;
    }
  }
  uint64_t added = merge_proposals(n, W, L->neighbors, rows_d, ft, fs, fd, c, T);
  free(ft); free(fs); free(fd); free(rows_d);
  return added;
}

/* link_nodes_in_layer_to_better_neighbors over all nodes  src/lib.rs:1070-1154.
 * link_count = self.neighborhood_size() (bp.neighborhood_size, also on layer 0: :1093) */
uint64_t orc_link_layer(orc_index *ix, uint32_t lft, orc_search_params sp, uint64_t link_count,
                        int threads) {
  uint64_t n = ix->layers[lft].node_count;
  uint64_t *ids = (uint64_t *)malloc(sizeof(uint64_t) * n * link_count);
  float *d = (float *)malloc(sizeof(float) * n * link_count);
  uint64_t *len = (uint64_t *)malloc(sizeof(uint64_t) * n);
  orc_link_search(ix, lft, sp, link_count, 0, n, ids, d, len, threads);
  uint64_t added = orc_link_apply(ix, lft, link_count, ids, d, len, threads);
  free(ids); free(d); free(len);
  return added;
}

/* ---------------------------------------------------------------- recall / improve */

/* hits among sample[first, first+count) of stochastic_recall_at  src/lib.rs:1463-1499 */
int orc_recall_hits(const orc_index *ix, uint32_t at, const orc_opt_params *op, uint64_t first, uint64_t count,
                    uint64_t *out_hits, uint64_t *out_selection, int threads) {
  const orc_layer *L = &ix->layers[at];
  uint64_t total = L->node_count;
  uint64_t selection = (uint64_t)((float)total * op->recall_proportion);
  if (selection < 1) selection = 1;
  if (selection > total) selection = total;
  uint64_t *vecs = (uint64_t *)malloc(sizeof(uint64_t) * total);
  memcpy(vecs, L->nodes, sizeof(uint64_t) * total);
  if (selection != total) orc_shuffle_u64(vecs, total, 42); /* StdRng::seed_from_u64(42) */
  if (first > selection) first = selection;
  if (first + count > selection) count = selection - first;
  uint64_t cap = op->search.number_of_candidates;
  uint64_t relevant = 0;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    orc_scratch *sc = orc_scratch_new(ix, 0);
    uint64_t *oi = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    float *od = (float *)malloc(sizeof(float) * cap);
#pragma omp for schedule(dynamic, 16) reduction(+ : relevant)
    for (uint64_t k = first; k < first + count; k++) {
      uint64_t len = 0;
      /* self.search(Stored(vid), op.search): the whole stack  :1488-1491 */
      orc_search_sc(ix, NULL, vecs[k], op->search, 0, ORC_EMPTY, oi, od, &len, NULL, sc, NULL);
      for (uint64_t j = 0; j < len; j++)
        if (oi[j] == vecs[k]) {
          relevant++;
          break;
        }
    }
    free(oi);
    free(od);
    orc_scratch_free(sc);
  }
  free(vecs);
  *out_hits = relevant;
  if (out_selection) *out_selection = selection;
  return 0;
}

/* stochastic_recall_at  src/lib.rs:1463-1499 */
float orc_stochastic_recall_at(const orc_index *ix, uint32_t at, const orc_opt_params *op, int threads) {
  uint64_t hits = 0, selection = 0;
  orc_recall_hits(ix, at, op, 0, UINT64_MAX / 2, &hits, &selection, threads);
  return (float)hits / (float)selection;
}

/* improve_neighbors_upto  src/lib.rs:1515-1544 */
float orc_improve_neighbors_upto(orc_index *ix, uint32_t upto, const orc_build_params *bp,
                                 float last_recall_or_nan, int threads) {
  const orc_opt_params *op = &bp->optimization;
  float last_recall = (last_recall_or_nan != last_recall_or_nan) ? 0.0f : last_recall_or_nan;
  float last_improvement = 1.0f;
  uint64_t rounds = 0;
  while (last_improvement >= op->neighborhood_threshold && last_recall < 1.0f) {
    for (uint32_t lft = 0; lft < upto; lft++)
      orc_link_layer(ix, lft, op->search, bp->neighborhood_size, threads);
    float recall = orc_stochastic_recall_at(ix, upto - 1, op, threads);
    last_improvement = recall - last_recall;
    last_recall = recall;
    rounds++;
    if (bp->max_link_rounds && rounds >= bp->max_link_rounds) break;
  }
  return last_recall;
}

/* ---------------------------------------------------------------- promotion */

/* match_within_epsilon  src/search.rs:173-187 */
static int match_within_epsilon(uint64_t vector, const uint64_t *ids, const float *d, uint64_t len) {
  int found = 0;
  const float epsilon = 1e-5f;
  for (uint64_t k = 0; k < len; k++) {
    if (fabsf(d[k]) < epsilon) {
      if (ids[k] == vector) found = 1;
    } else
      break;
  }
  return found;
}

/* self-hit flags (match_within_epsilon) of nodes [first, first+count) of layer lft: the
 * searches of discover_unreachable_vectors  src/lib.rs:1017-1025 */
int orc_discover_hits(const orc_index *ix, uint32_t lft, orc_search_params sp, uint64_t first, uint64_t count,
                      uint64_t *out_hit, int threads) {
  const orc_layer *cur = &ix->layers[lft];
  if (first + count > cur->node_count) return -3;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    orc_scratch *sc = orc_scratch_new(ix, 0);
    uint64_t cap = sp.number_of_candidates;
    uint64_t *oi = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    float *od = (float *)malloc(sizeof(float) * cap);
#pragma omp for schedule(dynamic, 16)
    for (uint64_t x = 0; x < count; x++) {
      uint64_t vector = cur->nodes[first + x], len = 0;
      orc_search_sc(ix, NULL, vector, sp, lft + 1, ORC_EMPTY, oi, od, &len, NULL, sc, NULL);
      out_hit[x] = (uint64_t)match_within_epsilon(vector, oi, od, len);
    }
    free(oi);
    free(od);
    orc_scratch_free(sc);
  }
  return 0;
}

/* the filter of discover_unreachable_vectors  src/lib.rs:1026-1034 from the full flags */
static uint64_t discover_filter(const orc_index *ix, uint32_t lft, const uint64_t *hit, uint64_t **out) {
  const orc_layer *cur = &ix->layers[lft];
  const orc_layer *above = lft ? &ix->layers[lft - 1] : NULL;
  uint64_t n = cur->node_count, c = 0;
  uint64_t *v = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
  for (uint64_t i = 0; i < n; i++)
    if (!hit[i] && (!above || orc_layer_get_node(above, cur->nodes[i]) == ORC_EMPTY)) v[c++] = cur->nodes[i];
  *out = v;
  return c;
}

/* discover_unreachable_vectors  src/lib.rs:1002-1037 */
uint64_t orc_discover_unreachable(const orc_index *ix, uint32_t lft, orc_search_params sp, uint64_t **out,
                                  int threads) {
  uint64_t n = ix->layers[lft].node_count;
  uint64_t *hit = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
  orc_discover_hits(ix, lft, sp, 0, n, hit, threads);
  uint64_t c = discover_filter(ix, lft, hit, out);
  free(hit);
  return c;
}

/* extend_layer  src/lib.rs:1039-1068 with generate_node_maps :1767-1812,
 * copy_old_neighborhoods_into_layer :1737-1765, initialize_new_neighborhoods :1727-1735 */
int orc_extend_layer(orc_index *ix, uint32_t lft, const uint64_t *vecs_in, uint64_t count) {
  orc_layer *L = &ix->layers[lft];
  uint64_t W = L->neighborhood_size, n_old = L->node_count, n_new = n_old + count;
  uint64_t *vecs = (uint64_t *)malloc(sizeof(uint64_t) * (count ? count : 1));
  memcpy(vecs, vecs_in, sizeof(uint64_t) * count);
  qsort(vecs, count, sizeof(uint64_t), u64_cmp);
  uint64_t *nodes = (uint64_t *)malloc(sizeof(uint64_t) * n_new);
  uint64_t *old_map = (uint64_t *)malloc(sizeof(uint64_t) * (n_old ? n_old : 1));
  uint8_t *is_new = (uint8_t *)calloc(n_new, 1);
  uint64_t a = 0, b = 0, o = 0;
  while (a < n_old || b < count) {
    if (b >= count || (a < n_old && L->nodes[a] < vecs[b])) {
      old_map[a] = o;
      nodes[o++] = L->nodes[a++];
    } else {
      if (a < n_old && L->nodes[a] == vecs[b]) { /* panic!("tried to insert vector that already exists") */
        free(vecs); free(nodes); free(old_map); free(is_new);
        return -5;
      }
      is_new[o] = 1;
      nodes[o++] = vecs[b++];
    }
  }
  uint64_t *nb = (uint64_t *)malloc(sizeof(uint64_t) * n_new * W);
  for (uint64_t i = 0; i < n_new * W; i++) nb[i] = ORC_EMPTY;
  for (uint64_t i = 0; i < n_old; i++)
    for (uint64_t k = 0; k < W; k++) {
      uint64_t x = L->neighbors[i * W + k];
      nb[old_map[i] * W + k] = x == ORC_EMPTY ? ORC_EMPTY : old_map[x];
    }
  free(L->nodes);
  free(L->neighbors);
  L->nodes = nodes;
  L->neighbors = nb;
  L->node_count = n_new;
  free(vecs); free(old_map); free(is_new);
  return 0;
}

typedef struct {
  uint64_t node;
  uint64_t count;
} histo_t;
static int histo_cmp(const void *x, const void *y) {
  const histo_t *a = (const histo_t *)x, *b = (const histo_t *)y;
  if (a->count != b->count) return a->count < b->count ? -1 : 1; /* sort_by_key(count)  :1228 */
  return a->node < b->node ? -1 : (a->node > b->node ? 1 : 0);  /* HashMap order in the reference */
}

/* filter_promotion_candidates  src/lib.rs:1176-1271.  Every unreachable vector of layer
 * `lft` is absent from the layer above (discover_unreachable_vectors :1027-1029), so by the
 * nesting invariant its discover_order_from_top is `lft` and there is one histogram. */
static uint64_t filter_promotion_candidates(orc_index *ix, uint32_t lft, const uint64_t *vecs, uint64_t nv,
                                            orc_search_params sp, uint64_t **out, int threads) {
  *out = NULL;
  if (lft == 0) return 0; /* :1182-1184 */
  const orc_store *S = &ix->store;
  const orc_layer *L = &ix->layers[lft];
  uint64_t W = L->neighborhood_size;
  uint64_t *count = (uint64_t *)calloc(L->node_count, sizeof(uint64_t));
  for (uint64_t k = 0; k < nv; k++) { /* histogramming :1190-1223 */
    uint64_t node = orc_layer_get_node(L, vecs[k]);
    uint64_t fin = orc_final_neighbor_idx(W, L->neighbors, node);
    for (uint64_t x = node * W; x < fin; x++) {
      uint64_t nbr = L->neighbors[x];
      uint64_t nvct = L->nodes[nbr];
      if (bsearch(&nvct, vecs, nv, sizeof(uint64_t), u64_cmp)) count[nbr]++;
    }
  }
  uint64_t nh = 0;
  for (uint64_t i = 0; i < L->node_count; i++) nh += count[i] != 0;
  histo_t *h = (histo_t *)malloc(sizeof(histo_t) * (nh ? nh : 1));
  uint64_t c = 0;
  for (uint64_t i = 0; i < L->node_count; i++)
    if (count[i]) {
      h[c].node = i;
      h[c].count = count[i];
      c++;
    }
  free(count);
  qsort(h, nh, sizeof(histo_t), histo_cmp);
  uint64_t *sel = (uint64_t *)malloc(sizeof(uint64_t) * (nh ? nh : 1));
  float *radius = (float *)malloc(sizeof(float) * (nh ? nh : 1));
  uint64_t ns = 0;
  orc_scratch *sc = orc_scratch_new(ix, 0);
  uint64_t cap = sp.number_of_candidates;
  uint64_t *oi = (uint64_t *)malloc(sizeof(uint64_t) * cap);
  float *od = (float *)malloc(sizeof(float) * cap);
  while (nh) { /* while let Some((node, _)) = histogram.pop()  :1243 */
    uint64_t vec = L->nodes[h[--nh].node];
    int covered = 0;
    orc_query_prepare(S, sc, NULL, vec);
    for (uint64_t k = 0; k < ns && !covered; k++) /* compare_vec(Stored(v), Stored(vec)) < radius  :1244-1249 */
      if (orc_query_dist(S, sc, sel[k]) < radius[k]) covered = 1;
    if (covered) continue;
    uint64_t len = 0;
    /* self.search_upto(Stored(vec), search_parameters, layer_from_top)  :1254-1258 */
    orc_search_sc(ix, NULL, vec, sp, lft, ORC_EMPTY, oi, od, &len, NULL, sc, NULL);
    sel[ns] = vec;
    radius[ns] = len ? od[0] : 0.0f; /* result[0].1 */
    ns++;
  }
  orc_scratch_free(sc);
  free(oi); free(od); free(h); free(radius);
  (void)threads;
  *out = sel;
  return ns;
}

static uint32_t partitions_from_bottom(uint64_t total, uint64_t order, uint64_t *out, uint32_t max_out);
static orc_index *generate_impl(const float *rows, uint64_t n_store, uint32_t dim, uint32_t ld, int metric, int sum_mode,
                                const orc_store *pq_from, const uint64_t *vids, uint64_t n,
                                const orc_build_params *bp, int threads);

static int promote_impl(orc_index *ix, uint32_t lft, const orc_build_params *bp, const uint64_t *hit, int threads);
/* promote_at_layer  src/lib.rs:1273-1427 */
int orc_promote_at_layer(orc_index *ix, uint32_t lft, const orc_build_params *bp, int threads) {
  return promote_impl(ix, lft, bp, NULL, threads);
}
/* ... from precomputed self-hit flags (the sharded drivers all-gather them) */
int orc_promote_at_layer_hits(orc_index *ix, uint32_t lft, const orc_build_params *bp, const uint64_t *hit,
                              int threads) {
  return promote_impl(ix, lft, bp, hit, threads);
}
static int promote_impl(orc_index *ix, uint32_t lft, const orc_build_params *bp, const uint64_t *hit, int threads) {
  float max_proportion = bp->optimization.promotion_proportion;
  uint64_t *vecs = NULL;
  uint64_t nv = hit ? discover_filter(ix, lft, hit, &vecs)
                    : orc_discover_unreachable(ix, lft, bp->optimization.search, &vecs, threads);
  if (nv == 0) {
    free(vecs);
    return 0;
  }
  if (max_proportion < 1.0f) { /* :1288-1294 */
    nv = (uint64_t)((float)nv * max_proportion);
    if (nv == 0) {
      free(vecs);
      return 0;
    }
  }
  uint64_t *sel = NULL;
  uint64_t ns = filter_promotion_candidates(ix, lft, vecs, nv, bp->optimization.search, &sel, threads);
  free(vecs);
  if (lft == 0 || ns == 0) { /* order_vecs empty (or an empty selection): nothing to extend, still "true" */
    free(sel);
    return 1;
  }
  /* the else branch of :1332-1420 (layer_from_top = lft >= 1) */
  uint64_t sizes[128], new_sizes[128], promo[128];
  uint32_t nsz = lft;
  for (uint32_t i = 0; i < nsz; i++) sizes[i] = ix->layers[lft - 1 - i].node_count; /* reversed: [0] = just above */
  uint32_t nnew = partitions_from_bottom(sizes[0] + ns, bp->order, new_sizes, 128);
  while (nnew < nsz) new_sizes[nnew++] = 0; /* :1345-1349 */
  uint32_t retop_upto = nnew - nsz;
  uint32_t npromo = nsz;
  for (uint32_t i = 0; i < nsz; i++) promo[i] = new_sizes[i] > sizes[i] ? new_sizes[i] - sizes[i] : 0;
  uint32_t offset = 0;
  if (retop_upto != 0) { /* :1361-1397 */
    uint32_t retop_index = npromo - retop_upto;
    uint64_t into_top = promo[retop_index];
    if (into_top > ns) into_top = ns;
    npromo = retop_index;
    const orc_layer *T = &ix->layers[retop_upto - 1];
    uint64_t nt = T->node_count + into_top;
    uint64_t *top = (uint64_t *)malloc(sizeof(uint64_t) * nt);
    memcpy(top, T->nodes, sizeof(uint64_t) * T->node_count);
    memcpy(top + T->node_count, sel, sizeof(uint64_t) * into_top);
    qsort(top, nt, sizeof(uint64_t), u64_cmp);
    uint64_t u = 0;
    for (uint64_t k = 0; k < nt; k++)
      if (k == 0 || top[k] != top[k - 1]) top[u++] = top[k]; /* dedup */
    orc_build_params nbp = *bp;
    nbp.zero_layer_neighborhood_size = bp->neighborhood_size; /* :1377-1379 */
    nbp.seed = bp->seed + 0x51ED270B9F3ULL + ix->layer_count;  /* thread_rng in the reference */
    orc_index *nt_ix = generate_impl(ix->store.rows, ix->store.n, ix->store.dim, ix->store.ld, ix->store.metric,
                                     ix->store.sum_mode, &ix->store, top, u, &nbp, threads);
    free(top);
    if (!nt_ix) {
      free(sel);
      return -1;
    }
    uint32_t new_top_len = nt_ix->layer_count;
    uint32_t keep = ix->layer_count - retop_upto;
    orc_layer *nl = (orc_layer *)malloc(sizeof(orc_layer) * (new_top_len + keep));
    memcpy(nl, nt_ix->layers, sizeof(orc_layer) * new_top_len);
    memcpy(nl + new_top_len, ix->layers + retop_upto, sizeof(orc_layer) * keep);
    for (uint32_t i = 0; i < retop_upto; i++) {
      free(ix->layers[i].nodes);
      free(ix->layers[i].neighbors);
    }
    free(ix->layers);
    ix->layers = nl;
    ix->layer_count = new_top_len + keep;
    free(nt_ix->layers); /* the layer structs moved; free only the shell */
    nt_ix->layers = NULL;
    nt_ix->layer_count = 0;
    orc_index_free(nt_ix);
    offset = new_top_len;
  }
  /* promotion_sizes.reverse(); extend each remaining layer above  :1398-1412 */
  for (uint32_t i = 0; i < npromo; i++) {
    uint64_t size = promo[npromo - 1 - i];
    uint32_t cur = offset + i;
    const orc_layer *L = &ix->layers[cur];
    uint64_t *tp = (uint64_t *)malloc(sizeof(uint64_t) * (ns ? ns : 1));
    uint64_t c = 0;
    for (uint64_t k = 0; k < ns && c < size; k++)
      if (orc_layer_get_node(L, sel[k]) == ORC_EMPTY) tp[c++] = sel[k];
    int rc = orc_extend_layer(ix, cur, tp, c);
    free(tp);
    if (rc) {
      free(sel);
      return -1;
    }
  }
  free(sel);
  return 1;
}

/* improve_index_at  src/lib.rs:1546-1603; *lft may grow when promotion adds layers */
static float improve_index_at(orc_index *ix, uint32_t *lft_io, const orc_build_params *bp, int threads) {
  const orc_opt_params *op = &bp->optimization;
  uint32_t lft = *lft_io;
  float recall = orc_stochastic_recall_at(ix, lft, op, threads);
  float improvement = 1.0f;
  int bailout = 1;
  while (improvement >= op->promotion_threshold && recall < 1.0f && bailout != 0) {
    float last = recall;
    uint32_t cur = 0;
    while (cur <= lft && bailout != 0) {
      uint32_t layer_count = ix->layer_count;
      recall = orc_improve_neighbors_upto(ix, cur + 1, bp, NAN, threads);
      if (recall == 1.0f) { /* :1569-1572 */
        cur++;
        continue;
      }
      if (bp->promote) {
        int pr = orc_promote_at_layer(ix, cur, bp, threads); /* :1575 */
        if (pr > 0) {
          uint32_t delta = ix->layer_count - layer_count;
          cur += delta;
          lft += delta;
          recall = orc_improve_neighbors_upto(ix, cur + 1, bp, recall, threads); /* :1586-1587 */
        }
      }
      cur++;
    }
    bailout--;
    improvement = recall - last;
  }
  *lft_io = lft;
  return recall;
}

/* improve_index  src/lib.rs:1664-1686; last_recall NaN = None (unwrap_or_else :1671) */
float orc_improve_index_from(orc_index *ix, const orc_build_params *bp, float last_recall, int threads) {
  float recall = last_recall == last_recall ? last_recall
                                            : orc_stochastic_recall_at(ix, ix->layer_count - 1, &bp->optimization, threads);
  uint32_t lft = 0;
  while (lft < ix->layer_count) {
    recall = improve_index_at(ix, &lft, bp, threads);
    lft++;
  }
  return recall;
}
float orc_improve_index(orc_index *ix, const orc_build_params *bp, int threads) {
  return orc_improve_index_from(ix, bp, NAN, threads);
}

/* Hnsw::generate  src/lib.rs:825-893 */
static orc_index *generate_impl(const float *rows, uint64_t n_store, uint32_t dim, uint32_t ld, int metric, int sum_mode,
                                const orc_store *pq_from, const uint64_t *vids, uint64_t n,
                                const orc_build_params *bp, int threads) {
  if (n == 0 || bp->order < 2) return NULL; /* assert!(total_size > 0) :837 */
  orc_index *ix = orc_index_new(rows, n_store, dim, ld, metric, sum_mode);
  if (!ix) return NULL;
  if (pq_from && pq_from->codes)
  {
    orc_index_set_pq(ix, pq_from->codes, pq_from->codebook, pq_from->pq_m, pq_from->pq_ksub, pq_from->pq_dsub);
    orc_index_set_pq_table_f16(ix, (int)pq_from->pq_table_f16);
  }
  uint64_t *vs = (uint64_t *)malloc(sizeof(uint64_t) * n);
  memcpy(vs, vids, sizeof(uint64_t) * n);
  orc_shuffle_u64(vs, n, bp->seed); /* vs.shuffle(&mut thread_rng()) :832-833 */
  uint64_t parts[128];
  uint32_t np = orc_calculate_partitions(n, bp->order, parts, 128);
  uint32_t i = 0;
  while (i != np) { /* :854-890 */
    uint64_t length = parts[i] < n ? parts[i] : n; /* :858-860 */
    uint32_t level = np - i - 1;
    uint64_t W = level == 0 ? bp->zero_layer_neighborhood_size : bp->neighborhood_size;
    if (orc_generate_layer(ix, vs, length, W, bp, threads)) {
      orc_index_free(ix);
      free(vs);
      return NULL;
    }
    uint32_t old_count = ix->layer_count;
    orc_improve_index(ix, bp, threads); /* :877 */
    uint32_t delta = ix->layer_count - old_count;
    if (delta > 0) { /* new layers were added: fix the partitions  :880-887 */
      uint64_t suffix[128];
      uint32_t ns = np - (i + 1);
      memcpy(suffix, parts + i + 1, sizeof(uint64_t) * ns);
      for (uint32_t k = 0; k < ix->layer_count; k++) parts[k] = ix->layers[k].node_count;
      memcpy(parts + ix->layer_count, suffix, sizeof(uint64_t) * ns);
      np = ix->layer_count + ns;
      i += delta;
    }
    i++;
  }
  free(vs);
  return ix;
}

orc_index *orc_generate(const float *rows, uint64_t n_store, uint32_t dim, uint32_t ld, int metric,
                        int sum_mode, const uint64_t *vids, uint64_t n, const orc_build_params *bp,
                        int threads) {
  return generate_impl(rows, n_store, dim, ld, metric, sum_mode, NULL, vids, n, bp, threads);
}

/* assert_layer_invariants  src/search.rs:142-171 */
int orc_check_layer_invariants(const orc_index *ix) {
  for (uint32_t i = 0; i < ix->layer_count; i++) {
    const orc_layer *c = &ix->layers[i];
    for (uint64_t k = 1; k < c->node_count; k++)
      if (c->nodes[k] <= c->nodes[k - 1]) return -1;
    if (i + 1 < ix->layer_count)
      for (uint64_t k = 0; k < c->node_count; k++)
        if (orc_layer_get_node(&ix->layers[i + 1], c->nodes[k]) == ORC_EMPTY) return -2;
  }
  return 0;
}
