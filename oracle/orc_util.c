/* ORACLE (test infrastructure): metrics, deterministic generators, brute force. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orc.h"

/* ---------------------------------------------------------------- metrics */

/* Sequential f32 accumulation exactly as the reference loops:
 *   result += f1 * f2                 src/bigvec.rs:48-51, src/lib.rs:1986-1989
 *   result += (f1 - f2).powi(2)       src/lib.rs:2432-2435
 * Rust never contracts mul+add; this file is compiled with -ffp-contract=off. */
static float acc_seq(const orc_store *s, const float *a, const float *b) {
  float r = 0.0f;
  if (s->metric == ORC_METRIC_L2) {
    for (uint32_t i = 0; i < s->dim; i++) {
      float d = a[i] - b[i];
      r += d * d;
    }
  } else {
    for (uint32_t i = 0; i < s->dim; i++) r += a[i] * b[i];
  }
  return r;
}

/* The gfx950 kernel's order: lane l of a 64-lane wavefront owns the 16-byte chunks
 * l, l+64, l+128, ... of the (zero padded) row and runs ONE fma chain over them in
 * address order; the 64 partial sums are then combined by an xor butterfly with masks
 * 32,16,8,4,2,1 (every lane computes acc[l] + acc[l^m]; addition commutes, so all lanes
 * hold the same value).  fmaf here == v_fma_f32 there (both correctly rounded). */
static float acc_blocked64(const orc_store *s, const float *a, const float *b) {
  float acc[64];
  uint32_t nchunk = s->ld / 4;
  for (uint32_t l = 0; l < 64; l++) {
    float r = 0.0f;
    for (uint32_t c = l; c < nchunk; c += 64) {
      for (uint32_t e = 0; e < 4; e++) {
        uint32_t i = c * 4 + e;
        if (s->metric == ORC_METRIC_L2) {
          float d = a[i] - b[i];
          r = fmaf(d, d, r);
        } else {
          r = fmaf(a[i], b[i], r);
        }
      }
    }
    acc[l] = r;
  }
  for (uint32_t m = 32; m >= 1; m >>= 1) {
    float nxt[64];
    for (uint32_t l = 0; l < 64; l++) nxt[l] = acc[l] + acc[l ^ m];
    memcpy(acc, nxt, sizeof(acc));
  }
  return acc[0];
}

static float acc_seqfma(const orc_store *s, const float *a, const float *b) {
  float r = 0.0f;
  if (s->metric == ORC_METRIC_L2) {
    for (uint32_t i = 0; i < s->dim; i++) {
      float d = a[i] - b[i];
      r = fmaf(d, d, r);
    }
  } else {
    for (uint32_t i = 0; i < s->dim; i++) r = fmaf(a[i], b[i], r);
  }
  return r;
}

/* Comparator::compare_raw  (src/lib.rs:59) for the three in-tree metrics */
float orc_distance(const orc_store *s, const float *a, const float *b) {
  float r = (s->sum_mode == ORC_SUM_BLOCKED64) ? acc_blocked64(s, a, b)
            : (s->sum_mode == ORC_SUM_SEQFMA) ? acc_seqfma(s, a, b) : acc_seq(s, a, b);
  switch (s->metric) {
    case ORC_METRIC_COSINE_HALF:
      return (1.0f - r) / 2.0f; /* src/bigvec.rs:52 */
    case ORC_METRIC_ONE_MINUS_DOT:
      return 1.0f - r; /* src/lib.rs:1990 */
    default:
      return sqrtf(r); /* src/lib.rs:2436 result.powf(0.5) */
  }
}

/* ------------------------------------------------- deterministic generators */

/* splitmix64 finaliser */
uint64_t orc_mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBULL;
  x ^= x >> 31;
  return x;
}

static uint64_t stream_next(uint64_t *state) {
  *state += 0x9E3779B97F4A7C15ULL;
  return orc_mix64(*state);
}

/* unbiased-enough bounded draw: high 64 bits of x*n */
static uint64_t bounded(uint64_t x, uint64_t n) { return (uint64_t)(((__uint128_t)x * n) >> 64); }

/* slice.shuffle(&mut rng) shape (rand SliceRandom: for i in (1..len).rev() swap(i, 0..=i));
 * the stream itself is ours -- rand's ChaCha12 is "parity unpinned". */
void orc_shuffle_u64(uint64_t *v, uint64_t n, uint64_t seed) {
  uint64_t st = orc_mix64(seed ^ 0x5851F42D4C957F2DULL);
  for (uint64_t i = n; i-- > 1;) {
    uint64_t j = bounded(stream_next(&st), i + 1);
    uint64_t t = v[i];
    v[i] = v[j];
    v[j] = t;
  }
}

/* component j of vector with key k: uniform in [-1,1) from 24 random bits */
static float synth_component(uint64_t key, uint32_t j) {
  uint64_t x = orc_mix64(key * 0x9E3779B97F4A7C15ULL + ((uint64_t)j + 1) * 0xD1B54A32D192ED03ULL);
  uint32_t m = (uint32_t)(x >> 40); /* 24 bits */
  return (float)m * (1.0f / 8388608.0f) - 1.0f;
}

/* random_normed_vec  src/bigvec.rs:59-65: uniform(-1,1) components, norm = sqrt(sum f*f)
 * accumulated sequentially in f32, each component divided by it. */
void orc_synth_rows(float *rows, uint64_t first, uint64_t count, uint32_t dim, uint32_t ld,
                    uint64_t seed, int normalize, int threads) {
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (uint64_t r = 0; r < count; r++) {
    float *row = rows + r * (uint64_t)ld;
    uint64_t key = seed + first + r;
    float ss = 0.0f;
    for (uint32_t j = 0; j < dim; j++) {
      float f = synth_component(key, j);
      row[j] = f;
      ss += f * f;
    }
    if (normalize) {
      float norm = sqrtf(ss);
      for (uint32_t j = 0; j < dim; j++) row[j] = row[j] / norm;
    }
    for (uint32_t j = dim; j < ld; j++) row[j] = 0.0f;
  }
}

/* pages of a freshly allocated buffer touched by the threads that will read them (static
 * schedule, like the search loops): bench.py's cpu_baseline fills the buffer afterwards */
void orc_first_touch(float *p, uint64_t n_floats, int threads) {
  const uint64_t page = 1024; /* floats per 4 KiB page */
  const uint64_t pages = (n_floats + page - 1) / page;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (uint64_t i = 0; i < pages; i++) p[i * page] = 0.0f;
}

/* Clustered synthetic rows (SURVEY section 8d's second dataset): n_clusters unit centres
 * (synthetic normalised rows keyed seed ^ CENTRE_SALT), point i belongs to cluster
 * mulhi(mix64(..i..), n_clusters) and is normalise(centre + a * u), u_j uniform(-1,1) keyed
 * (seed + i, j), a = noise * sqrt(3/dim) so that |a*u| ~ noise.  Not in the reference. */
#define ORC_CENTRE_SALT 0xC1A55E5EEDULL
void orc_synth_clustered_rows(float *rows, uint64_t first, uint64_t count, uint32_t dim, uint32_t ld,
                              uint64_t seed, uint32_t n_clusters, float noise, int threads) {
  float *cent = (float *)malloc(sizeof(float) * (size_t)n_clusters * ld);
  orc_synth_rows(cent, 0, n_clusters, dim, ld, seed ^ ORC_CENTRE_SALT, 1, threads);
  float a = noise * sqrtf(3.0f / (float)dim);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (uint64_t r = 0; r < count; r++) {
    float *row = rows + r * (uint64_t)ld;
    uint64_t key = seed + first + r;
    uint64_t k = bounded(orc_mix64(key * 0xA24BAED4963EE407ULL + 0x9FB21C651E98DF25ULL), n_clusters);
    const float *c = cent + k * (uint64_t)ld;
    float ss = 0.0f;
    for (uint32_t j = 0; j < dim; j++) {
      float x = c[j] + a * synth_component(key, j);
      row[j] = x;
      ss += x * x;
    }
    float norm = sqrtf(ss);
    for (uint32_t j = 0; j < dim; j++) row[j] = row[j] / norm;
    for (uint32_t j = dim; j < ld; j++) row[j] = 0.0f;
  }
  free(cent);
}

/* 4-round Feistel network over 2*h bits with cycle walking: a keyed permutation of
 * [0,domain).  Stands in for choose_n_1's shuffle+truncate (src/lib.rs:1830-1852):
 * the first k images are a pseudo-random k-subset. */
uint64_t orc_feistel_perm(uint64_t i, uint64_t domain, uint64_t key) {
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
      uint64_t f = orc_mix64(R + key * 0x9E3779B97F4A7C15ULL + r * 0xC2B2AE3D27D4EB4FULL) & mask;
      uint64_t nl = R;
      R = L ^ f;
      L = nl;
    }
    x = (L << h) | R;
  } while (x >= domain);
  return x;
}

/* ------------------------------------------------------------- brute force */

typedef struct {
  float d;
  uint64_t id;
} pair_t;

static int pair_less(pair_t a, pair_t b) { return a.d < b.d || (a.d == b.d && a.id < b.id); }

/* exact k nearest by (distance, id) ascending; ground truth for recall@k.  The reference
 * has no such routine (it only measures self-recall, lib.rs:1485-1496). */
int orc_bruteforce(const orc_store *s, const float *queries, uint32_t ldq, uint64_t nq, uint64_t k,
                   uint64_t *out_ids, float *out_d, int threads) {
  if (k == 0 || k > s->n) return -1;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
  for (uint64_t q = 0; q < nq; q++) {
    pair_t *heap = (pair_t *)malloc(sizeof(pair_t) * k); /* max-heap of the k best */
    uint64_t hn = 0;
    const float *qv = queries + q * (uint64_t)ldq;
    for (uint64_t i = 0; i < s->n; i++) {
      pair_t p = {orc_distance(s, qv, s->rows + i * (uint64_t)s->ld), i};
      if (hn < k) {
        uint64_t c = hn++;
        heap[c] = p;
        while (c > 0) {
          uint64_t par = (c - 1) / 2;
          if (pair_less(heap[par], heap[c])) {
            pair_t t = heap[par];
            heap[par] = heap[c];
            heap[c] = t;
            c = par;
          } else
            break;
        }
      } else if (pair_less(p, heap[0])) {
        heap[0] = p;
        uint64_t c = 0;
        for (;;) {
          uint64_t l = 2 * c + 1, r = l + 1, b = c;
          if (l < hn && pair_less(heap[b], heap[l])) b = l;
          if (r < hn && pair_less(heap[b], heap[r])) b = r;
          if (b == c) break;
          pair_t t = heap[b];
          heap[b] = heap[c];
          heap[c] = t;
          c = b;
        }
      }
    }
    /* heap sort ascending */
    for (uint64_t e = hn; e-- > 0;) {
      pair_t top = heap[0];
      heap[0] = heap[e];
      uint64_t c = 0;
      for (;;) {
        uint64_t l = 2 * c + 1, r = l + 1, b = c;
        if (l < e && pair_less(heap[b], heap[l])) b = l;
        if (r < e && pair_less(heap[b], heap[r])) b = r;
        if (b == c) break;
        pair_t t = heap[b];
        heap[b] = heap[c];
        heap[c] = t;
        c = b;
      }
      out_ids[q * k + e] = top.id;
      out_d[q * k + e] = top.d;
    }
    free(heap);
  }
  return 0;
}
