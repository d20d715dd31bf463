// Device helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>

#include "phnsw_internal.h"

#define EXPF 0x80000000u
#define IDM 0x7FFFFFFFu
#define KEY_NONE 0xFFFFFFFFFFFFFFFFull

#define ST_OK 0u
#define ST_MISSING 4u
#define ST_OVERFLOW 5u

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

// one lane's share of Comparator::compare_raw: chunks lane, lane+64, ... in address order
template <int NV>
__device__ __forceinline__ float row_partial(const float4 *__restrict__ row, const float4 (&q)[NV],
                                             uint32_t nv4, uint32_t lane, bool l2) {
  float4 x[NV];
#pragma unroll
  for (int k = 0; k < NV; k++) {
    uint32_t c = lane + 64u * k;
    if (c < nv4) x[k] = row[c];
  }
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    uint32_t c = lane + 64u * k;
    if (c < nv4) {
      if (l2) {
        float d0 = q[k].x - x[k].x, d1 = q[k].y - x[k].y, d2 = q[k].z - x[k].z, d3 = q[k].w - x[k].w;
        acc = fmaf(d0, d0, acc);
        acc = fmaf(d1, d1, acc);
        acc = fmaf(d2, d2, acc);
        acc = fmaf(d3, d3, acc);
      } else {
        acc = fmaf(q[k].x, x[k].x, acc);
        acc = fmaf(q[k].y, x[k].y, acc);
        acc = fmaf(q[k].z, x[k].z, acc);
        acc = fmaf(q[k].w, x[k].w, acc);
      }
    }
  }
  return acc;
}

__device__ __forceinline__ float finalize_metric(float r, int metric) {
  if (metric == PHNSW_METRIC_COSINE_HALF) return (1.0f - r) / 2.0f;  // bigvec.rs:52
  if (metric == PHNSW_METRIC_ONE_MINUS_DOT) return 1.0f - r;         // lib.rs:1990
  return sqrtf(r);                                                    // lib.rs:2436
}

// partition_point over the sorted (d,id) queue held in LDS
__device__ __forceinline__ uint32_t lds_lower_bound(const uint32_t *ids, const float *ds, uint32_t len,
                                                    uint64_t key) {
  uint32_t lo = 0, hi = len;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    uint64_t k = mkkey(ds[mid], ids[mid]);
    if (k < key)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

