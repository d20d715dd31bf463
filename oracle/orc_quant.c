/* ORACLE (test infrastructure): the query side of compare_vec for f32 and product-quantised
 * stores, and the PQ flow of src/pq.rs:61-81 (quantize / reconstruct), 261-285
 * (random_centroids), 346-364 (QuantizedHnsw::search).
 *
 * The reference pins nothing numeric here: its quantised comparators reconstruct both
 * vectors and apply the full metric, PartialDistance::partial_distance is todo!()
 * (pq.rs:569-573, 751-755), and its tests assert recall only -- "parity unpinned".  The
 * definition below is the one the HIP path implements and is compared
 * with it bit for bit:
 *   table  T[j][k] = fma chain over e of  q_sub_j[e]*c_jk[e]   (or (q-c)^2 for L2)
 *   r      = T[0][code_0] + T[1][code_1] + ...   sequential f32 adds
 *   d      = metric epilogue(r)
 *   code_j = argmin_k sum_e (x_sub_j[e]-c_jk[e])^2 (fma chain), ties to the smaller k
 * A Stored query is its reconstruction, which makes code-vs-code distances symmetric. */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orc_internal.h"

/* f32 -> IEEE binary16 -> f32 with round to nearest even (integer arithmetic, the same
 * steps as the device code) */
static float round_to_f16(float f) {
  union { float f; uint32_t u; } v;
  v.f = f;
  uint32_t x = v.u, sign = (x >> 16) & 0x8000u, mant = x & 0x007FFFFFu, hbits;
  int32_t exp = (int32_t)((x >> 23) & 0xFF);
  if (exp == 0xFF)
    hbits = sign | 0x7C00u | (mant ? 0x200u : 0);
  else {
    int32_t e = exp - 127 + 15;
    if (e >= 0x1F)
      hbits = sign | 0x7C00u;
    else if (e <= 0) {
      if (e < -10)
        hbits = sign;
      else {
        mant |= 0x00800000u;
        uint32_t shift = (uint32_t)(14 - e), hm = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1u))) hm++;
        hbits = sign | hm;
      }
    } else {
      uint32_t hm = mant >> 13, rem = mant & 0x1FFFu;
      hbits = sign | ((uint32_t)e << 10) | hm;
      if (rem > 0x1000u || (rem == 0x1000u && (hm & 1u))) hbits++;
    }
  }
  uint32_t s2 = (hbits & 0x8000u) << 16, e2 = (hbits >> 10) & 0x1Fu, m2 = hbits & 0x3FFu, u;
  if (e2 == 0) {
    if (m2 == 0)
      u = s2;
    else {
      int s = 0;
      while (!(m2 & 0x400u)) {
        m2 <<= 1;
        s++;
      }
      u = s2 | ((uint32_t)(127 - 15 - s + 1) << 23) | ((m2 & 0x3FFu) << 13);
    }
  } else if (e2 == 0x1F)
    u = s2 | 0x7F800000u | (m2 << 13);
  else
    u = s2 | ((e2 - 15 + 127) << 23) | (m2 << 13);
  v.u = u;
  return v.f;
}

static float pq_entry(const orc_store *S, const float *qs, uint32_t j, uint32_t k, int l2) {
  const float *c = S->codebook + ((uint64_t)j * S->pq_ksub + k) * S->pq_dsub;
  float acc = 0.0f;
  for (uint32_t e = 0; e < S->pq_dsub; e++) {
    if (l2) {
      float df = qs[e] - c[e];
      acc = fmaf(df, df, acc);
    } else {
      acc = fmaf(qs[e], c[e], acc);
    }
  }
  return acc;
}

/* table modes: 0 = f32 entries; 1 = each entry rounded to IEEE half once;
 * 2 = 8-bit entries: u[j][k] = rint((T[j][k] - min_j) / scale), scale = (widest row range) / 255,
 * distance = bias + scale * (integer sum), bias = sum_j min_j added in j order.  T holds the
 * entries as floats in every mode (small integers in mode 2). */
static void pq_build_table(const orc_store *S, const float *raw, const uint8_t *qcodes, float *T, float *bias,
                           float *scale) {
  const int l2 = S->metric == ORC_METRIC_L2;
  *bias = 0.0f;
  *scale = 0.0f;
  for (uint32_t j = 0; j < S->pq_m; j++) {
    const float *qs = raw ? raw + (uint64_t)j * S->pq_dsub
                          : S->codebook + ((uint64_t)j * S->pq_ksub + qcodes[j]) * S->pq_dsub;
    for (uint32_t k = 0; k < S->pq_ksub; k++) {
      float acc = pq_entry(S, qs, j, k, l2);
      T[j * S->pq_ksub + k] = S->pq_table_f16 == 1 ? round_to_f16(acc) : acc;
    }
  }
  if (S->pq_table_f16 != 2) return;
  float widest = 0.0f;
  for (uint32_t j = 0; j < S->pq_m; j++) {
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (uint32_t k = 0; k < S->pq_ksub; k++) {
      float v = T[j * S->pq_ksub + k];
      lo = v < lo ? v : lo;
      hi = v > hi ? v : hi;
    }
    float range = hi - lo;
    widest = range > widest ? range : widest;
    *bias = *bias + lo;
  }
  *scale = widest / 255.0f;
  for (uint32_t j = 0; j < S->pq_m; j++) {
    float lo = FLT_MAX;
    for (uint32_t k = 0; k < S->pq_ksub; k++) lo = T[j * S->pq_ksub + k] < lo ? T[j * S->pq_ksub + k] : lo;
    for (uint32_t k = 0; k < S->pq_ksub; k++) {
      float v = T[j * S->pq_ksub + k];
      T[j * S->pq_ksub + k] = *scale > 0.0f ? (float)(uint8_t)rintf((v - lo) / *scale) : 0.0f;
    }
  }
}

void orc_query_prepare(const orc_store *S, orc_scratch *sc, const float *raw, uint64_t stored_id) {
  if (!S->codes) {
    sc->qv = raw ? raw : S->rows + stored_id * (uint64_t)S->ld;
    return;
  }
  if (!sc->pq_table) sc->pq_table = (float *)malloc(sizeof(float) * (size_t)S->pq_m * S->pq_ksub);
  pq_build_table(S, raw, raw ? NULL : S->codes + stored_id * (uint64_t)S->pq_m, sc->pq_table, &sc->pq_bias,
                 &sc->pq_scale);
  sc->qv = raw;
}

static float metric_epilogue(const orc_store *S, float r) {
  switch (S->metric) {
    case ORC_METRIC_COSINE_HALF:
      return (1.0f - r) / 2.0f;
    case ORC_METRIC_ONE_MINUS_DOT:
      return 1.0f - r;
    default:
      return sqrtf(r);
  }
}

float orc_query_dist(const orc_store *S, const orc_scratch *sc, uint64_t vid) {
  if (!S->codes) return orc_distance(S, sc->qv, S->rows + vid * (uint64_t)S->ld);
  const uint8_t *code = S->codes + vid * (uint64_t)S->pq_m;
  float r = 0.0f;
  if (S->pq_table_f16 == 2) {
    uint32_t sum = 0;
    for (uint32_t j = 0; j < S->pq_m; j++) sum += (uint32_t)sc->pq_table[j * S->pq_ksub + code[j]];
    float scaled = sc->pq_scale * (float)sum;
    r = sc->pq_bias + scaled;
  } else {
    for (uint32_t j = 0; j < S->pq_m; j++) r = r + sc->pq_table[j * S->pq_ksub + code[j]];
  }
  return metric_epilogue(S, r);
}

void orc_index_set_pq_table_f16(orc_index *ix, int on) { ix->store.pq_table_f16 = on == 2 ? 2u : (on ? 1u : 0u); }

void orc_index_set_pq(orc_index *ix, const uint8_t *codes, const float *codebook, uint32_t m, uint32_t ksub,
                      uint32_t dsub) {
  ix->store.codes = codes;
  ix->store.codebook = codebook;
  ix->store.pq_m = m;
  ix->store.pq_ksub = ksub;
  ix->store.pq_dsub = dsub;
}

/* Quantizer::quantize  pq.rs:61-71 (exact nearest centroid instead of the HNSW over centroids) */
void orc_pq_encode(const float *rows, uint64_t n, uint32_t ld, uint32_t m, uint32_t ksub, uint32_t dsub,
                   const float *codebook, uint8_t *codes, int threads) {
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (uint64_t i = 0; i < n; i++) {
    const float *x = rows + i * (uint64_t)ld;
    for (uint32_t j = 0; j < m; j++) {
      float best = 0.0f;
      uint32_t bk = 0;
      for (uint32_t k = 0; k < ksub; k++) {
        const float *c = codebook + ((uint64_t)j * ksub + k) * dsub;
        float acc = 0.0f;
        for (uint32_t e = 0; e < dsub; e++) {
          float df = x[j * dsub + e] - c[e];
          acc = fmaf(df, df, acc);
        }
        if (k == 0 || acc < best) {
          best = acc;
          bk = k;
        }
      }
      codes[i * m + j] = (uint8_t)bk;
    }
  }
}

/* Codebooks.  kmeans_iters == 0: random_centroids (pq.rs:261-285), the sub-vectors of ksub randomly
 * selected vectors.  kmeans_iters > 0 (per-sub-space k-means, own implementation;
 * the reference's linfa k-means is dead code, pq.rs:215-259): Lloyd iterations from that start over
 * the first min(n, sample) vectors of the same shuffle -- assign = the quantizer's exact nearest
 * centroid, update = mean of the members accumulated in f64 IN TRAINING ORDER and rounded to f32 once
 * (an empty cell keeps its centroid).  Every step is a fixed sequence of IEEE operations, so the
 * device path (csrc/pq.hip) reproduces the codebook bit for bit. */
int orc_pq_create_kmeans(const float *rows, uint64_t n, uint32_t dim, uint32_t ld, uint32_t m, uint32_t ksub,
                         uint64_t seed, uint32_t kmeans_iters, uint64_t sample, uint8_t *codes, float *codebook,
                         int threads) {
  if (m == 0 || dim % m || ksub == 0 || ksub > 256 || ksub > n) return -3;
  uint32_t dsub = dim / m;
  uint64_t *perm = (uint64_t *)malloc(sizeof(uint64_t) * n);
  for (uint64_t i = 0; i < n; i++) perm[i] = i;
  orc_shuffle_u64(perm, n, seed ^ 0x9C0DEB00C5ULL);
  for (uint32_t j = 0; j < m; j++)
    for (uint32_t k = 0; k < ksub; k++)
      memcpy(codebook + ((uint64_t)j * ksub + k) * dsub, rows + perm[k] * (uint64_t)ld + j * dsub, sizeof(float) * dsub);
  if (kmeans_iters) {
    uint64_t S = sample && sample < n ? sample : n;
    if (S < ksub) S = ksub;
    float *train = (float *)malloc(sizeof(float) * S * ld);
    for (uint64_t i = 0; i < S; i++) memcpy(train + i * ld, rows + perm[i] * (uint64_t)ld, sizeof(float) * ld);
    uint8_t *tc = (uint8_t *)malloc(S * m);
    double *sum = (double *)malloc(sizeof(double) * (size_t)ksub * dsub);
    uint64_t *cnt = (uint64_t *)malloc(sizeof(uint64_t) * ksub);
    for (uint32_t it = 0; it < kmeans_iters; it++) {
      orc_pq_encode(train, S, ld, m, ksub, dsub, codebook, tc, threads);
      for (uint32_t j = 0; j < m; j++) {
        memset(sum, 0, sizeof(double) * (size_t)ksub * dsub);
        memset(cnt, 0, sizeof(uint64_t) * ksub);
        for (uint64_t i = 0; i < S; i++) { /* every cell sees its members in training order */
          uint32_t k = tc[i * m + j];
          cnt[k]++;
          for (uint32_t e = 0; e < dsub; e++) sum[(size_t)k * dsub + e] += (double)train[i * ld + j * dsub + e];
        }
        for (uint32_t k = 0; k < ksub; k++)
          if (cnt[k])
            for (uint32_t e = 0; e < dsub; e++)
              codebook[((uint64_t)j * ksub + k) * dsub + e] = (float)(sum[(size_t)k * dsub + e] / (double)cnt[k]);
      }
    }
    free(cnt);
    free(sum);
    free(tc);
    free(train);
  }
  free(perm);
  orc_pq_encode(rows, n, ld, m, ksub, dsub, codebook, codes, threads);
  return 0;
}

int orc_pq_create(const float *rows, uint64_t n, uint32_t dim, uint32_t ld, uint32_t m, uint32_t ksub, uint64_t seed,
                  uint8_t *codes, float *codebook, int threads) {
  return orc_pq_create_kmeans(rows, n, dim, ld, m, ksub, seed, 0, 0, codes, codebook, threads);
}

typedef struct {
  float d;
  uint64_t id;
} rr_pair;
static int rr_cmp(const void *a, const void *b) {
  const rr_pair *x = (const rr_pair *)a, *y = (const rr_pair *)b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

/* QuantizedHnsw::search  pq.rs:346-364 */
int orc_pq_search_batch(const orc_index *ix, const orc_store *full, const float *queries, uint32_t ldq, uint64_t nq,
                        orc_search_params sp, int quantize_query, uint64_t *out_ids, float *out_d, uint64_t *out_len,
                        orc_stats *st, int threads) {
  const orc_store *S = &ix->store;
  if (!S->codes) return -3;
  uint64_t cap = sp.number_of_candidates;
  uint32_t dim = S->pq_m * S->pq_dsub;
  int rc_all = 0;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    orc_scratch *sc = orc_scratch_new(ix, 0);
    float *qq = (float *)malloc(sizeof(float) * dim);
    uint8_t *qc = (uint8_t *)malloc(S->pq_m);
    rr_pair *rr = (rr_pair *)malloc(sizeof(rr_pair) * cap);
#pragma omp for schedule(dynamic, 8)
    for (uint64_t q = 0; q < nq; q++) {
      const float *raw = queries + q * (uint64_t)ldq;
      const float *qs = raw;
      if (quantize_query) { /* quantizer.quantize(&raw_v)  :351-352 */
        orc_pq_encode(raw, 1, ldq, S->pq_m, S->pq_ksub, S->pq_dsub, S->codebook, qc, 1);
        for (uint32_t j = 0; j < S->pq_m; j++)
          memcpy(qq + j * S->pq_dsub, S->codebook + ((uint64_t)j * S->pq_ksub + qc[j]) * S->pq_dsub,
                 sizeof(float) * S->pq_dsub);
        qs = qq;
      }
      orc_stats s = {0, 0};
      uint64_t len = 0;
      int rc = orc_search_sc(ix, qs, 0, sp, 0, ORC_EMPTY, out_ids + q * cap, out_d + q * cap, &len, &s, sc, NULL);
      if (rc) {
#pragma omp atomic write
        rc_all = rc;
        len = 0;
      }
      /* re-rank with the full comparator, sort_by_key (d, id)  :354-361 */
      for (uint64_t k = 0; k < len; k++) {
        rr[k].id = out_ids[q * cap + k];
        rr[k].d = orc_distance(full, raw, full->rows + rr[k].id * (uint64_t)full->ld);
      }
      qsort(rr, len, sizeof(rr_pair), rr_cmp);
      for (uint64_t k = 0; k < len; k++) {
        out_ids[q * cap + k] = rr[k].id;
        out_d[q * cap + k] = rr[k].d;
      }
      out_len[q] = len;
      if (st) st[q] = s;
    }
    free(qq);
    free(qc);
    free(rr);
    orc_scratch_free(sc);
  }
  return rc_all;
}
