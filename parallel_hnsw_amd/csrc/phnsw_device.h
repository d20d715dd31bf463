// Device helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "phnsw_internal.h"

#define EXPF 0x80000000u
#define IDM 0x7FFFFFFFu
#define KEY_NONE 0xFFFFFFFFFFFFFFFFull

#define ST_OK 0u
#define ST_MISSING 4u
#define ST_OVERFLOW 5u
#define ST_CAPACITY 6u

__device__ __forceinline__ uint32_t fkey(float d) {
  d += 0.0f;  // -0.0 -> +0.0 (OrderedFloat treats them as equal, src/types.rs:78-88)
  uint32_t u = __float_as_uint(d);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// total order of the reference: (OrderedFloat(d), id) ascending (src/lib.rs:206)
__device__ __forceinline__ uint64_t mkkey(float d, uint32_t id) {
  return ((uint64_t)fkey(d) << 32) | (uint64_t)(id & IDM);
}
__device__ __forceinline__ uint32_t rl32(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ uint64_t rl64(uint64_t v, int lane) {
  return ((uint64_t)rl32((uint32_t)(v >> 32), lane) << 32) | rl32((uint32_t)v, lane);
}
__device__ __forceinline__ uint32_t rfl32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint64_t lanemask_lt(uint32_t lane) { return (1ull << lane) - 1ull; }

__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 32);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

// one lane's share of Comparator::compare_raw for U candidate rows at once: chunks lane,
// lane+64, ... of each row in address order, ONE fma chain per row (the order the oracle's
// ORC_SUM_BLOCKED64 restates).  All U*NV loads are issued before the first use so that U
// rows are in flight per wave; out-of-range chunks (dim not a multiple of 256) load a
// clamped address and are skipped in the accumulate by a select, not a branch.
// the fma chain itself: chunks lane, lane+64, ... of one row (already in registers) against the
// query's chunks.  Every kernel that evaluates a distance goes through this function, so the
// bits of a distance do not depend on which kernel produced it (search, dense top-layer table,
// seeding, row distances).
template <int NV, bool EXACT, bool L2>
__device__ __forceinline__ float chain_partial(const float4 (&x)[NV], const float4 (&q)[NV], uint32_t nv4, uint32_t lane) {
  float a = 0.f;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    float t = a;
    if (L2) {
      float d0 = q[k].x - x[k].x, d1 = q[k].y - x[k].y, d2 = q[k].z - x[k].z, d3 = q[k].w - x[k].w;
      t = fmaf(d0, d0, t);
      t = fmaf(d1, d1, t);
      t = fmaf(d2, d2, t);
      t = fmaf(d3, d3, t);
    } else {
      t = fmaf(q[k].x, x[k].x, t);
      t = fmaf(q[k].y, x[k].y, t);
      t = fmaf(q[k].z, x[k].z, t);
      t = fmaf(q[k].w, x[k].w, t);
    }
    a = (EXACT || lane + 64u * k < nv4) ? t : a;
  }
  return a;
}

// The same chain for TWO queries against one row, as packed f32 operations (v_pk_fma_f32: two
// independent IEEE fmas per instruction, so each half carries exactly the bits of chain_partial;
// the plain v_fma_f32 issues at half the chip's f32 vector rate).  q2[k][e] = component e of chunk k
// of (query a, query b).  Used by the dense top-layer tile pass (tiny.hip).
typedef float ph_f2 __attribute__((ext_vector_type(2)));
// NP pairs at once, the independent chains interleaved (k, e outer; pair inner) so that consecutive
// instructions never depend on each other; every chain still sees its own operations in chain_partial's order
template <int NV, int NP, bool EXACT, bool L2>
__device__ __forceinline__ void chain_partial2(const float4 (&x)[NV], const ph_f2 (&q2)[NP][NV][4], uint32_t nv4,
                                               uint32_t lane, ph_f2 (&a)[NP]) {
#pragma unroll
  for (int j = 0; j < NP; j++) a[j] = ph_f2{0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NV; k++) {
    ph_f2 t[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) t[j] = a[j];
    const float xs[4] = {x[k].x, x[k].y, x[k].z, x[k].w};
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const ph_f2 xb = {xs[e], xs[e]};
#pragma unroll
      for (int j = 0; j < NP; j++) {
        if (L2) {
          const ph_f2 d = q2[j][k][e] - xb;
          t[j] = __builtin_elementwise_fma(d, d, t[j]);
        } else {
          t[j] = __builtin_elementwise_fma(q2[j][k][e], xb, t[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NP; j++) a[j] = (EXACT || lane + 64u * k < nv4) ? t[j] : a[j];
  }
}

template <int NV, int U, bool EXACT, bool L2>
__device__ __forceinline__ void rows_partial_impl(const float4 *const (&row)[U], const float4 (&q)[NV], uint32_t nv4,
                                                  uint32_t lane, float (&acc)[U]) {
  float4 x[U][NV];
#pragma unroll
  for (int u = 0; u < U; u++) {
#pragma unroll
    for (int k = 0; k < NV; k++) {
      uint32_t c = lane + 64u * k;
      if (!EXACT) c = c < nv4 ? c : nv4 - 1;
      x[u][k] = row[u][c];
    }
  }
#pragma unroll
  for (int u = 0; u < U; u++) acc[u] = chain_partial<NV, EXACT, L2>(x[u], q, nv4, lane);
}

template <int NV, int U>
__device__ __forceinline__ void rows_partial(const float4 *const (&row)[U], const float4 (&q)[NV], uint32_t nv4,
                                             uint32_t lane, bool l2, float (&acc)[U]) {
  const bool exact = nv4 == 64u * NV;  // wave-uniform
  if (exact) {
    if (l2)
      rows_partial_impl<NV, U, true, true>(row, q, nv4, lane, acc);
    else
      rows_partial_impl<NV, U, true, false>(row, q, nv4, lane, acc);
  } else {
    if (l2)
      rows_partial_impl<NV, U, false, true>(row, q, nv4, lane, acc);
    else
      rows_partial_impl<NV, U, false, false>(row, q, nv4, lane, acc);
  }
}

template <int NV>
__device__ __forceinline__ float row_partial(const float4 *__restrict__ row, const float4 (&q)[NV], uint32_t nv4,
                                             uint32_t lane, bool l2) {
  const float4 *const r[1] = {row};
  float acc[1];
  rows_partial<NV, 1>(r, q, nv4, lane, l2, acc);
  return acc[0];
}

// distances of the query to the (up to 64) rows whose ids sit in the lanes flagged by
// `mask`; 4 rows in flight; the result lands in the lane that held the id
template <int NV, int U>
__device__ __forceinline__ float batch_distances(const float *__restrict__ vecs, uint32_t ld, uint32_t nv4, int metric,
                                                 bool l2, const float4 (&qv)[NV], uint64_t mask, uint32_t vid,
                                                 uint32_t lane);

__device__ __forceinline__ float finalize_metric(float r, int metric) {
  if (metric == PHNSW_METRIC_COSINE_HALF) return (1.0f - r) / 2.0f;  // bigvec.rs:52
  if (metric == PHNSW_METRIC_ONE_MINUS_DOT) return 1.0f - r;         // lib.rs:1990
  return sqrtf(r);                                                    // lib.rs:2436
}

// partition_point over the sorted (d,id) queue held in LDS.  `len` is the same in every lane, so the halving
// loop is scalar control flow with one select per step (no divergent branches); every lane may call it, whatever
// its key.
__device__ __forceinline__ uint32_t lds_lower_bound(const uint32_t *ids, const float *ds, uint32_t len,
                                                    uint64_t key) {
  if (len == 0) return 0;
  uint32_t base = 0, n = len;
  while (n > 1) {
    const uint32_t half = n >> 1;
    const uint32_t probe = base + half - 1;
    base = mkkey(ds[probe], ids[probe]) < key ? base + half : base;
    n -= half;
  }
  return base + (mkkey(ds[base], ids[base]) < key ? 1u : 0u);
}

// minimum over the wave, in every lane (and uniform): four DPP rotations inside the 16-lane rows, then the four
// row minima through SGPRs
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xF, 0xF, false));  // row_ror:8
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xF, 0xF, false));  // row_ror:4
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xF, 0xF, false));  // row_ror:2
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xF, 0xF, false));  // row_ror:1
  return min(min(rl32(v, 0), rl32(v, 16)), min(rl32(v, 32), rl32(v, 48)));
}

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }


// wave_sum of FOUR rows at once: the first two butterfly steps fold the four registers into one
// (v_permlane32_swap / v_permlane16_swap, gfx950: a lane keeps the step of the row its 16-lane group will
// hold), the last four run once on that register with DPP moves -- 7 shuffles instead of 24, every sum formed
// from the same operands in the same tree as wave_sum (f32 add is commutative).  Afterwards lanes 0-15 hold
// row 0's sum, 16-31 row 2's, 32-47 row 1's, 48-63 row 3's.
__device__ __forceinline__ float wave_sum4(float p0, float p1, float p2, float p3) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(p2), __float_as_uint(p3), false, false);
  const float f0 = __uint_as_float(a[0]) + __uint_as_float(a[1]);  // lanes < 32: row 0, >= 32: row 1
  const float f1 = __uint_as_float(b[0]) + __uint_as_float(b[1]);  // lanes < 32: row 2, >= 32: row 3
  auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(f0), __float_as_uint(f1), false, false);
  float v = __uint_as_float(c[0]) + __uint_as_float(c[1]);
  int t = __builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x128, 0xF, 0xF, true);  // row_ror:8
  v += __uint_as_float((uint32_t)t);
  const int o = (int)__float_as_uint(v);
  t = __builtin_amdgcn_update_dpp(o, o, 0x114, 0xF, 0xA, false);  // row_shr:4 into lanes 4-7, 12-15
  t = __builtin_amdgcn_update_dpp(t, o, 0x104, 0xF, 0x5, false);  // row_shl:4 into lanes 0-3, 8-11
  v += __uint_as_float((uint32_t)t);
  t = __builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  v += __uint_as_float((uint32_t)t);
  t = __builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  v += __uint_as_float((uint32_t)t);
  return v;
}

// one round of batch_distances: candidates base .. base+U-1 (in lane order) of the compacted list
template <int NV, int U>
__device__ __forceinline__ void distance_round(const float *__restrict__ vecs, uint32_t nv4, int metric, bool l2,
                                               const float4 (&qv)[NV], uint32_t olo, uint32_t ohi, uint32_t m, uint32_t base,
                                               bool cand, uint32_t myrank, uint32_t lane, float &myd) {
    const float4 *r[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t k = min(base + (uint32_t)u, m - 1u);  // short tail: the last row again (same value, unused)
      const uint64_t o = ((uint64_t)rl32(ohi, (int)k) << 32) | rl32(olo, (int)k);
      r[u] = (const float4 *)((const char *)vecs + o);
    }
    float p[U];
    rows_partial<NV, U>(r, qv, nv4, lane, l2, p);
    const uint32_t mine = myrank - base;  // < U when this is the lane's round
    if constexpr (U % 4 == 0) {
      // row u of a group of four ends in lanes {0, 32, 16, 48}[u] + 0..15 (wave_sum4): every candidate lane
      // fetches its own row's sum with one ds_bpermute
#pragma unroll
      for (int g = 0; g < U; g += 4) {
        const float d4 = finalize_metric(wave_sum4(p[g], p[g + 1], p[g + 2], p[g + 3]), metric);
        const uint32_t u = (mine - g) & 3u;
        const uint32_t src = ((u & 1u) << 5) | ((u & 2u) << 3);  // 0, 32, 16, 48
        const float got = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)__float_as_uint(d4)));
        if (cand && mine >= (uint32_t)g && mine < (uint32_t)g + 4u) myd = got;
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) {
        float d = finalize_metric(wave_sum(p[u]), metric);
        if (cand && mine == (uint32_t)u) myd = d;
      }
    }
}

template <int NV, int U>
__device__ __forceinline__ float batch_distances(const float *__restrict__ vecs, uint32_t ld, uint32_t nv4, int metric,
                                                 bool l2, const float4 (&qv)[NV], uint64_t mask, uint32_t vid,
                                                 uint32_t lane) {
  // U rows in flight per wave: 4 for throughput (8 was measured: 183 VGPRs, 2 waves/SIMD, no faster on full
  // batches); the latency kernels of small batches take 12 (one load round per hop instead of four)
  float myd = 0.f;
  const uint32_t m = __popcll(mask);
  if (m == 0) return myd;
  // The candidates are compacted first: candidate number k (in lane order) leaves the byte offset of its row in
  // lane k (ds_permute), so that round r simply reads lanes r*U .. r*U + U-1 -- no bit scans, and the 64-bit
  // address arithmetic of a row is one v_mad_u64_u32 per lane instead of scalar multiplies per row.
  const bool cand = (mask >> lane) & 1ull;
  const uint32_t myrank = __popcll(mask & lanemask_lt(lane));
  const uint64_t off = (uint64_t)vid * ((uint64_t)ld * 4u);
  const int dst = (int)((cand ? myrank : 63u) << 2);  // non-candidates park in lane 63 (never read: m <= 63 there)
  uint32_t olo = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)(uint32_t)off);
  uint32_t ohi = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)(uint32_t)(off >> 32));
  if (m == 64) {  // every lane is a candidate: the permutation is the identity
    olo = (uint32_t)off;
    ohi = (uint32_t)(off >> 32);
  }
  if constexpr (U == 0) {
    // latency kernels: as few load rounds as the hop allows -- 24 rows in flight when more than 12 remain, else
    // 12, else 4 (a round computes all its U rows whatever the count)
    uint32_t base = 0;
    while (base < m) {
      const uint32_t left = m - base;
      constexpr int UBIG = NV <= 3 ? 24 : 12;  // 24 rows of 6 float4 per lane would not fit the register file
      if (left > 12u && UBIG > 12) {
        distance_round<NV, UBIG>(vecs, nv4, metric, l2, qv, olo, ohi, m, base, cand, myrank, lane, myd);
        base += (uint32_t)UBIG;
      } else if (left > 4u) {
        distance_round<NV, 12>(vecs, nv4, metric, l2, qv, olo, ohi, m, base, cand, myrank, lane, myd);
        base += 12u;
      } else {
        distance_round<NV, 4>(vecs, nv4, metric, l2, qv, olo, ohi, m, base, cand, myrank, lane, myd);
        base += 4u;
      }
    }
  } else {
    for (uint32_t base = 0; base < m; base += U)
      distance_round<NV, U>(vecs, nv4, metric, l2, qv, olo, ohi, m, base, cand, myrank, lane, myd);
  }
  return myd;
}

// ------------------------------------------------------------------ distance policies
// The traversal kernels are written once and instantiated per policy:
//   DistF32<NV>: the query in registers, candidates = f32 rows (wave per row, 4 in flight)
//   DistPQ:      product-quantised store: a per-query table T[m][ksub] in LDS
//                (T[j][k] = <q_sub_j, c_jk> or |q_sub_j - c_jk|^2), candidates = u8 code
//                rows, one LANE per candidate, distance = sum_j T[j][code_j] added in j order.
// DistNone: the policy of a launch that walks dense top layers only (every distance comes from the table)
struct DistNone {
  static constexpr bool GLOBAL_TABLE = false;
  static constexpr bool EARLY = false;
  __device__ __forceinline__ void prepare_raw(const PhDistArgs &, const float *, float *, uint32_t) {}
  __device__ __forceinline__ void prepare_stored(const PhDistArgs &, uint32_t, float *, uint32_t) {}
  __device__ __forceinline__ float batch(const PhDistArgs &, uint64_t, uint32_t, uint32_t) const { return 0.f; }
};
template <class D>
struct dist_is_none {
  static constexpr bool value = false;
};
template <>
struct dist_is_none<DistNone> {
  static constexpr bool value = true;
};

template <int NV, int U = 4>
struct DistF32 {
  static constexpr bool GLOBAL_TABLE = false;
  static constexpr bool EARLY = false;
  float4 qv[NV];
  __device__ __forceinline__ void prepare_raw(const PhDistArgs &d, const float *q, float *, uint32_t lane) {
#pragma unroll
    for (int k = 0; k < NV; k++) {
      uint32_t c = lane + 64u * k;
      qv[k] = (c < d.nv4) ? ((const float4 *)q)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void prepare_stored(const PhDistArgs &d, uint32_t vid, float *lds, uint32_t lane) {
    prepare_raw(d, d.vecs + (uint64_t)vid * d.ld, lds, lane);
  }
  __device__ __forceinline__ float batch(const PhDistArgs &d, uint64_t mask, uint32_t vid, uint32_t lane) const {
    return batch_distances<NV, U>(d.vecs, d.ld, d.nv4, d.metric, d.metric == PHNSW_METRIC_L2, qv, mask, vid, lane);
  }
};

// f32 -> IEEE binary16 bits, round to nearest even, written with integer operations so that
// the oracle's C version produces the same bits (no dependence on a denormal mode)
__host__ __device__ inline uint16_t ph_f32_to_f16_bits(float f) {
  union {
    float f;
    uint32_t u;
  } v;
  v.f = f;
  uint32_t x = v.u;
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t mant = x & 0x007FFFFFu;
  int32_t exp = (int32_t)((x >> 23) & 0xFF);
  if (exp == 0xFF) return (uint16_t)(sign | 0x7C00u | (mant ? 0x200u : 0));
  int32_t e = exp - 127 + 15;
  if (e >= 0x1F) return (uint16_t)(sign | 0x7C00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    mant |= 0x00800000u;
    uint32_t shift = (uint32_t)(14 - e);
    uint32_t hm = mant >> shift;
    uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1u))) hm++;
    return (uint16_t)(sign | hm);
  }
  uint32_t hm = mant >> 13, rem = mant & 0x1FFFu;
  uint32_t out = sign | ((uint32_t)e << 10) | hm;
  if (rem > 0x1000u || (rem == 0x1000u && (hm & 1u))) out++;  // may carry into the exponent: correct
  return (uint16_t)out;
}
__host__ __device__ inline float ph_f16_bits_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, u;
  if (e == 0) {
    if (m == 0)
      u = sign;
    else {
      int s = 0;
      while (!(m & 0x400u)) {
        m <<= 1;
        s++;
      }
      u = sign | ((uint32_t)(127 - 15 - s + 1) << 23) | ((m & 0x3FFu) << 13);
    }
  } else if (e == 0x1F)
    u = sign | 0x7F800000u | (m << 13);
  else
    u = sign | ((e - 15 + 127) << 23) | (m << 13);
  union {
    uint32_t u;
    float f;
  } v;
  v.u = u;
  return v.f;
}

template <bool GLOBAL>
struct DistPQT {
  // GLOBAL = false: the table lives in LDS (one per wave: 48-96 KiB, so 1-3 waves per CU);
  // GLOBAL = true: in a per-wave slot of HBM-backed memory that stays in L2 -- the batched search
  // uses this one: the lookups become 2- or 4-byte gathers from L2, and the CU holds as many
  // searching waves as its registers allow instead of as many tables as its LDS allows
  static constexpr bool GLOBAL_TABLE = GLOBAL;
  static constexpr bool EARLY = false;
  float *T;  // [m][ksub] (f32) or the same region viewed as uint16_t / uint8_t [m][ksub]
  float bias, scale;  // table mode 2 (8-bit entries): distance = bias + scale * sum of entries
  __device__ __forceinline__ float entry(const PhDistArgs &d, const float *qs, uint32_t j, uint32_t k, bool l2) const {
    const float *c = d.codebook + ((uint64_t)j * d.ksub + k) * d.dsub;
    float acc = 0.f;
    for (uint32_t e = 0; e < d.dsub; e++) {
      if (l2) {
        float df = qs[e] - c[e];
        acc = fmaf(df, df, acc);
      } else {
        acc = fmaf(qs[e], c[e], acc);
      }
    }
    return acc;
  }
  // q_sub_j comes from `q` (raw query, dim floats) or from the codebook entry of a stored code
  __device__ __forceinline__ void build(const PhDistArgs &d, const float *q, const uint8_t *qcodes, float *lds,
                                        uint32_t lane) {
    T = lds;
    uint16_t *T16 = (uint16_t *)lds;
    uint8_t *T8 = (uint8_t *)lds;
    const bool l2 = d.metric == PHNSW_METRIC_L2;
    bias = 0.f;
    scale = 0.f;
    float rmin0 = 0.f, rmin1 = 0.f;  // row minima: row j in lane j & 63 of register j >> 6 (m <= 128)
    if (d.table_f16 == 2) {
      // pass 1: per-row minimum and the widest row range (exact: min / max only)
      float widest = 0.f;
      for (uint32_t j = 0; j < d.m; j++) {
        const float *qs = q ? q + (uint64_t)j * d.dsub : d.codebook + ((uint64_t)j * d.ksub + qcodes[j]) * d.dsub;
        float lo = PH_FMAX, hi = -PH_FMAX;
        for (uint32_t k = lane; k < d.ksub; k += 64) {
          float v = entry(d, qs, j, k, l2);
          lo = fminf(lo, v);
          hi = fmaxf(hi, v);
        }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
          lo = fminf(lo, __shfl_xor(lo, sft));
          hi = fmaxf(hi, __shfl_xor(hi, sft));
        }
        widest = fmaxf(widest, __fsub_rn(hi, lo));
        if (lane == (j & 63u)) {
          if (j < 64)
            rmin0 = lo;
          else
            rmin1 = lo;
        }
        bias = __fadd_rn(bias, lo);  // in j order
      }
      scale = __fdiv_rn(widest, 255.0f);
    }
    for (uint32_t j = 0; j < d.m; j++) {
      const float *qs = q ? q + (uint64_t)j * d.dsub
                          : d.codebook + ((uint64_t)j * d.ksub + qcodes[j]) * d.dsub;
      float rowmin = 0.f;
      if (d.table_f16 == 2)
        rowmin = __uint_as_float(rl32(__float_as_uint(j < 64 ? rmin0 : rmin1), (int)(j & 63u)));
      for (uint32_t k = lane; k < d.ksub; k += 64) {
        float acc = entry(d, qs, j, k, l2);
        if (d.table_f16 == 2)
          T8[j * d.ksub + k] = scale > 0.f ? (uint8_t)rintf(__fdiv_rn(__fsub_rn(acc, rowmin), scale)) : (uint8_t)0;
        else if (d.table_f16)
          T16[j * d.ksub + k] = ph_f32_to_f16_bits(acc);
        else
          T[j * d.ksub + k] = acc;
      }
    }
    if (GLOBAL) {
      // own stores out to L2, then drop this CU's L1 lines (the slot held the previous query's table)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else {
      __syncthreads();
    }
  }
  __device__ __forceinline__ void prepare_raw(const PhDistArgs &d, const float *q, float *lds, uint32_t lane) {
    build(d, q, nullptr, lds, lane);
  }
  __device__ __forceinline__ void prepare_stored(const PhDistArgs &d, uint32_t vid, float *lds, uint32_t lane) {
    build(d, nullptr, d.codes + (uint64_t)vid * d.m, lds, lane);
  }
  __device__ __forceinline__ float at(const PhDistArgs &d, uint32_t idx) const {
    // half -> float is exact; v_cvt_f32_f16 does it in one instruction
    return d.table_f16 ? __half2float(__ushort_as_half(((const uint16_t *)T)[idx])) : T[idx];
  }
  __device__ __forceinline__ float batch(const PhDistArgs &d, uint64_t mask, uint32_t vid, uint32_t lane) const {
    float r = 0.f;
    if ((mask >> lane) & 1ull) {
      const uint32_t *row = (const uint32_t *)(d.codes + (uint64_t)vid * d.m);  // m % 4 == 0
      if (d.table_f16 == 2) {
        const uint8_t *T8 = (const uint8_t *)T;
        uint32_t sum = 0;
        for (uint32_t w = 0; w < d.m / 4; w++) {
          uint32_t cw = row[w];
          uint32_t j = 4 * w;
          sum += T8[(j + 0) * d.ksub + (cw & 0xFF)];
          sum += T8[(j + 1) * d.ksub + ((cw >> 8) & 0xFF)];
          sum += T8[(j + 2) * d.ksub + ((cw >> 16) & 0xFF)];
          sum += T8[(j + 3) * d.ksub + (cw >> 24)];
        }
        r = __fadd_rn(bias, __fmul_rn(scale, (float)sum));  // exact integer sum, two roundings
      } else {
        for (uint32_t w = 0; w < d.m / 4; w++) {
          uint32_t cw = row[w];
          uint32_t j = 4 * w;
          r = __fadd_rn(r, at(d, (j + 0) * d.ksub + (cw & 0xFF)));
          r = __fadd_rn(r, at(d, (j + 1) * d.ksub + ((cw >> 8) & 0xFF)));
          r = __fadd_rn(r, at(d, (j + 2) * d.ksub + ((cw >> 16) & 0xFF)));
          r = __fadd_rn(r, at(d, (j + 3) * d.ksub + (cw >> 24)));
        }
      }
      r = finalize_metric(r, d.metric);
    }
    return r;
  }
};
typedef DistPQT<false> DistPQ;
typedef DistPQT<true> DistPQG;

// DistPQR<M>: the 8-bit lookup table (table mode 2) of a query held in REGISTERS -- the register file of a
// CU (512 KiB) is larger than its LDS (160 KiB).  Row j of the table (256 one-byte entries) is exactly one
// VGPR: lane l holds entries 4l .. 4l+3.  A candidate sits in one lane; its entry for sub-space j is
// fetched from lane code_j >> 2 with ds_bpermute_b32 (the LDS crossbar, no LDS storage, no bank conflicts
// between rows) and byte code_j & 3 is extracted.  The sum of the M entries is an exact integer, so the
// distance bits equal DistPQT's mode 2 and the oracle's (ksub == 256 only).  The code rows of a hop's neighbours are requested
// BEFORE the visited test-and-set returns (EARLY): the two round trips overlap; rows of already visited
// neighbours are fetched for nothing (96 B each).
template <int M>
struct DistPQR {
  // the table is built by DistPQT<true> in the wave's slot of global memory (the same code, hence the same
  // bits, as table mode 2 everywhere else) and then read into registers once: row j = 256 bytes = one
  // coalesced dword per lane
  static constexpr bool GLOBAL_TABLE = true;
  static constexpr bool EARLY = true;
  uint32_t T[M];
  uint4 cw[M / 16];  // the candidate's code row: M bytes = M/16 16-byte loads (rows are 16-byte aligned: M % 32 == 0)
  float bias, scale;
  __device__ __forceinline__ void load_table(DistPQT<true> &b, uint32_t lane) {
    bias = b.bias;
    scale = b.scale;
    // The slot is the same for every query of this wave, so its M row addresses are loop invariants: hoisted out of
    // the query loop they cost 2 registers each and the kernel spilled 168 of them to scratch at M = 96.  The base is
    // made opaque per query (a wave-uniform SGPR pair), the rows are reached through immediate offsets.
    uint64_t base = (uint64_t)b.T;
    uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    asm volatile("" : "+s"(blo), "+s"(bhi));
    const uint32_t *t32 = (const uint32_t *)(((uint64_t)bhi << 32) | blo) + lane;
#pragma unroll
    for (int j = 0; j < M; j++) T[j] = t32[j * 64];
  }
  __device__ __forceinline__ void prepare_raw(const PhDistArgs &d, const float *q, float *slot, uint32_t lane) {
    DistPQT<true> b;
    b.prepare_raw(d, q, slot, lane);
    load_table(b, lane);
  }
  __device__ __forceinline__ void prepare_stored(const PhDistArgs &d, uint32_t vid, float *slot, uint32_t lane) {
    DistPQT<true> b;
    b.prepare_stored(d, vid, slot, lane);
    load_table(b, lane);
  }
  // request the code row of the candidate in this lane (valid lanes only)
  __device__ __forceinline__ void prefetch(const PhDistArgs &d, bool valid, uint32_t vid, uint32_t) {
    const uint4 *row = (const uint4 *)(d.codes + (uint64_t)(valid ? vid : 0u) * M);
#pragma unroll
    for (int w = 0; w < M / 16; w++) cw[w] = row[w];
  }
  // One lookup = the entry's dword from lane code >> 2 of row register j (ds_bpermute: the hardware takes
  // (address / 4) mod 64, so the shifted code word serves as the address without masking) and its byte code & 3
  // (v_bfe_u32 reads the low five bits of its offset operand).  Four VALU instructions and one crossbar pass per
  // lookup; the sum is an exact integer, whatever the order.
  __device__ __forceinline__ float finish(const PhDistArgs &d, uint64_t mask, uint32_t lane) const {
    uint32_t sum = 0;
#pragma unroll
    for (int w = 0; w < M / 4; w++) {
      const uint4 q4 = cw[w >> 2];
      const uint32_t c4 = (w & 3) == 0 ? q4.x : ((w & 3) == 1 ? q4.y : ((w & 3) == 2 ? q4.z : q4.w));
      const uint32_t e4 = c4 & 0x03030303u;  // byte i: index of the entry inside its dword
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t addr = i ? (c4 >> (8 * i)) : c4;
        const uint32_t sh = i ? (e4 >> (8 * i - 3)) : (e4 << 3);
        const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr, (int)T[4 * w + i]);
        sum += __builtin_amdgcn_ubfe(t, sh, 8u);
      }
      // eight lookups in flight are enough to cover the crossbar's latency; without the fence the
      // scheduler hoists all M of them and spills the table
      if ((w & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    float r = __fadd_rn(bias, __fmul_rn(scale, (float)sum));  // exact integer sum, two roundings
    return ((mask >> lane) & 1ull) ? finalize_metric(r, d.metric) : 0.f;
  }
  __device__ __forceinline__ float batch(const PhDistArgs &d, uint64_t mask, uint32_t vid, uint32_t lane) {
    prefetch(d, (mask >> lane) & 1ull, vid, lane);
    return finish(d, mask, lane);
  }
};

// DistPQS<NV>: the reference's quantised comparator (pq.rs:585-599, 767-782): a stored vector IS its
// reconstruction -- sub-vector j is centroid codes16[v][j] of ONE shared codebook -- and the distance is the
// full metric on it.  The query sits in registers like DistF32's; a candidate's chunk lane + 64k is fetched
// through its code (two dependent loads: u16 code, then 16 bytes of the centroid), and the fma chain and the
// butterfly are chain_partial / wave_sum, so every distance has exactly the bits DistF32 produces on the
// materialised reconstructions (which is how the graph over the codes is built and how the tests check it).
template <int NV>
struct DistPQS {
  static constexpr bool GLOBAL_TABLE = false;
  static constexpr bool EARLY = false;
  float4 qv[NV];
  uint32_t jk[NV], offk[NV];  // this lane's chunk k: sub-space and float offset inside the centroid
  __device__ __forceinline__ void lanes(const PhDistArgs &d, uint32_t lane) {
#pragma unroll
    for (int k = 0; k < NV; k++) {
      uint32_t c = lane + 64u * k;
      c = c < d.nv4 ? c : d.nv4 - 1u;
      jk[k] = (4u * c) / d.dsub;
      offk[k] = (4u * c) % d.dsub;
    }
  }
  __device__ __forceinline__ float4 chunk(const PhDistArgs &d, uint32_t vid, int k) const {
    const uint32_t code = d.codes16[(uint64_t)vid * d.m + jk[k]];
    return *(const float4 *)(d.codebook + (uint64_t)code * d.dsub + offk[k]);
  }
  __device__ __forceinline__ void prepare_raw(const PhDistArgs &d, const float *q, float *, uint32_t lane) {
    lanes(d, lane);
#pragma unroll
    for (int k = 0; k < NV; k++) {
      uint32_t c = lane + 64u * k;
      qv[k] = (c < d.nv4) ? ((const float4 *)q)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void prepare_stored(const PhDistArgs &d, uint32_t vid, float *, uint32_t lane) {
    lanes(d, lane);
#pragma unroll
    for (int k = 0; k < NV; k++) qv[k] = (lane + 64u * k < d.nv4) ? chunk(d, vid, k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ float batch(const PhDistArgs &d, uint64_t mask, uint32_t vid, uint32_t lane) const {
    constexpr int U = 4;
    const bool l2 = d.metric == PHNSW_METRIC_L2, exact = d.nv4 == 64u * NV;
    float myd = 0.f;
    uint64_t rem = mask;
    while (rem) {
      int l[U];
      l[0] = __builtin_ctzll(rem);
      rem &= rem - 1;
#pragma unroll
      for (int u = 1; u < U; u++) {
        l[u] = rem ? __builtin_ctzll(rem) : l[0];
        rem &= rem ? rem - 1 : 0;
      }
      float4 x[U][NV];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t v = rl32(vid, l[u]);
#pragma unroll
        for (int k = 0; k < NV; k++) x[u][k] = chunk(d, v, k);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        float p;
        if (exact)
          p = l2 ? chain_partial<NV, true, true>(x[u], qv, d.nv4, lane) : chain_partial<NV, true, false>(x[u], qv, d.nv4, lane);
        else
          p = l2 ? chain_partial<NV, false, true>(x[u], qv, d.nv4, lane) : chain_partial<NV, false, false>(x[u], qv, d.nv4, lane);
        const float dd = finalize_metric(wave_sum(p), d.metric);
        if ((int)lane == l[u]) myd = dd;
      }
    }
    return myd;
  }
};
