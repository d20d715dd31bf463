// Product quantisation on gfx950 (BASELINE config 5; reference src/pq.rs).
//
// What the reference has (pq.rs:61-81, 261-364): a codebook of centroid sub-vectors picked at
// random from the data (random_centroids, :261-285), quantize = nearest centroid per
// sub-vector (through an HNSW over the centroids, :61-71), an Hnsw over the code rows whose
// comparator reconstructs both sides and applies the full metric (test comparators
// :585-599; PartialDistance is todo!()), and a search that re-ranks the quantised result
// with the full-precision comparator and sorts by (d, id) (:346-364).
//
// What is built here: m sub-spaces of dsub = dim/m floats, one
// codebook of ksub <= 256 centroids PER sub-space, u8 codes (m bytes per vector).  A PQ
// store is a `phnsw_store` whose rows are code rows; every traversal kernel runs on it
// through the DistPQ policy (phnsw_device.h): a per-query table T[m][ksub] in LDS (96 KiB at
// m=96, ksub=256) -- ADC for raw queries, and for Stored queries the table of the
// reconstruction, which makes code-vs-code distances symmetric sums of centroid-pair terms
// (the reference's PartialDistance idea).  Encoding is the exact nearest centroid (the
// reference's HNSW-over-centroids approximates it).
//
//   ph_pq_gather_codebook_kernel   random_centroids, per sub-space
//   ph_pq_encode_kernel            Quantizer::quantize for every vector, one wave per vector
//   ph_pq_reconstruct_kernel       Quantizer::reconstruct
//   ph_pq_rerank_kernel            the re-rank + sort tail of QuantizedHnsw::search
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "phnsw_device.h"

__global__ void ph_pq_gather_codebook_kernel(const float *rows, uint32_t ld, const uint32_t *sample, uint32_t m,
                                             uint32_t ksub, uint32_t dsub, float *codebook) {
  uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;  // (j, k, e)
  uint32_t total = m * ksub * dsub;
  if (x >= total) return;
  uint32_t e = x % dsub, k = (x / dsub) % ksub, j = x / (dsub * ksub);
  codebook[x] = rows[(uint64_t)sample[k] * ld + j * dsub + e];
}

// one wave per vector; per sub-space every lane scores ksub/64 centroids with a sequential
// fma chain over the dsub components, then the wave takes the minimum of (distance, k)
// (codes[i * si + j * sj]: si = m, sj = 1 is the store's row layout; the k-means trainer uses the
// transposed one, si = 1, sj = n)
__global__ __launch_bounds__(64) void ph_pq_encode_kernel(const float *rows, uint32_t ld, uint64_t n, uint32_t m,
                                                          uint32_t ksub, uint32_t dsub, const float *codebook,
                                                          uint8_t *codes, uint64_t si, uint64_t sj) {
  const uint32_t lane = threadIdx.x;
  for (uint64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const float *x = rows + i * ld;
    for (uint32_t j = 0; j < m; j++) {
      const float *xs = x + j * dsub;
      uint64_t best = KEY_NONE;
      for (uint32_t k = lane; k < ksub; k += 64) {
        const float *c = codebook + ((uint64_t)j * ksub + k) * dsub;
        float acc = 0.f;
        for (uint32_t e = 0; e < dsub; e++) {
          float df = xs[e] - c[e];
          acc = fmaf(df, df, acc);
        }
        uint64_t key = ((uint64_t)__float_as_uint(acc) << 32) | k;  // acc >= 0: bits order as floats
        best = key < best ? key : best;
      }
#pragma unroll
      for (int s = 32; s >= 1; s >>= 1) {
        uint64_t o = ((uint64_t)__shfl_xor((uint32_t)(best >> 32), s) << 32) | __shfl_xor((uint32_t)best, s);
        best = o < best ? o : best;
      }
      if (lane == 0) codes[i * si + j * sj] = (uint8_t)(best & 0xFF);
    }
  }
}

// k-means update of one cell (sub-space j, centroid k) per wave: the members' sub-vectors summed in f64
// in training order (the ballot's set bits, ascending), mean rounded to f32 once; an empty cell keeps its
// centroid.  tcT = training codes, transposed [m][S].  Definition shared with oracle/orc_quant.c.
__global__ __launch_bounds__(64) void ph_pq_kmeans_update_kernel(const float *train, uint32_t ld, uint32_t S, uint32_t m,
                                                                 uint32_t ksub, uint32_t dsub, const uint8_t *tcT,
                                                                 float *codebook) {
  const uint32_t lane = threadIdx.x, j = blockIdx.x / ksub, k = blockIdx.x % ksub;
  double sum[4] = {0.0, 0.0, 0.0, 0.0};  // components lane, lane + 64, ... (dsub <= 256)
  uint32_t cnt = 0;
  for (uint32_t base = 0; base < S; base += 64) {
    const uint32_t i = base + lane;
    const bool mine = i < S && tcT[(uint64_t)j * S + i] == (uint8_t)k;
    uint64_t mm = __ballot(mine);
    cnt += __popcll(mm);
    while (mm) {
      const uint32_t b = __builtin_ctzll(mm);
      mm &= mm - 1;
      const float *x = train + (uint64_t)(base + b) * ld + (uint64_t)j * dsub;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const uint32_t e = lane + 64u * c;
        if (e < dsub) sum[c] += (double)x[e];
      }
    }
  }
  if (cnt) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint32_t e = lane + 64u * c;
      if (e < dsub) codebook[((uint64_t)j * ksub + k) * dsub + e] = (float)(sum[c] / (double)cnt);
    }
  }
}

__global__ void ph_pq_gather_train_kernel(const float *rows, uint32_t ld, const uint32_t *ids, uint32_t S, float *out) {
  const uint32_t r = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
  if (r >= S) return;
  const float *src = rows + (uint64_t)ids[r] * ld;
  for (uint32_t c = lane; c < ld; c += 64) out[(uint64_t)r * ld + c] = src[c];
}

__global__ void ph_pq_reconstruct_kernel(const uint8_t *codes, uint64_t n, uint32_t m, uint32_t ksub, uint32_t dsub,
                                         const float *codebook, float *out, uint32_t ld) {
  uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t dim = (uint64_t)m * dsub;
  if (x >= n * dim) return;
  uint64_t i = x / dim;
  uint32_t c = (uint32_t)(x % dim), j = c / dsub, e = c % dsub;
  out[i * ld + c] = codebook[((uint64_t)j * ksub + codes[i * m + j]) * dsub + e];
}

// re-rank: full-precision distance of the query to each of its quantised results, then
// sort by (d, id)  pq.rs:354-361.  One wave per query, ids in/out [nq][ef].
template <int NV>
__global__ __launch_bounds__(64) void ph_pq_rerank_kernel(PhDistArgs full, const float *queries, uint32_t ldq,
                                                          uint32_t nq, uint32_t ef, const uint32_t *len,
                                                          uint32_t *ids, float *d) {
  extern __shared__ uint64_t rk[];  // [ef] keys, [ef] sorted
  const uint32_t lane = threadIdx.x;
  for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
    DistF32<NV> dist;
    dist.prepare_raw(full, queries + (uint64_t)q * ldq, nullptr, lane);
    const uint32_t cnt = min(len[q], ef);
    for (uint32_t base = 0; base < cnt; base += 64) {
      uint32_t i = base + lane;
      uint32_t id = i < cnt ? ids[(uint64_t)q * ef + i] : 0u;
      float dd = dist.batch(full, __ballot(i < cnt), id, lane);
      if (i < cnt) rk[i] = mkkey(dd, id);
    }
    __syncthreads();
    for (uint32_t c = lane; c < cnt; c += 64) {
      uint64_t kc = rk[c];
      uint32_t rank = 0;
      for (uint32_t e = 0; e < cnt; e++) {
        uint64_t ke = rk[e];
        rank += (ke < kc || (ke == kc && e < c)) ? 1u : 0u;
      }
      rk[ef + rank] = kc;
    }
    __syncthreads();
    for (uint32_t i = lane; i < ef; i += 64) {
      bool live = i < cnt;
      uint64_t k = live ? rk[ef + i] : 0;
      uint32_t fk = (uint32_t)(k >> 32);
      uint32_t u = (fk & 0x80000000u) ? (fk ^ 0x80000000u) : ~fk;
      ids[(uint64_t)q * ef + i] = live ? ((uint32_t)k & IDM) : PH_EMPTY32;
      d[(uint64_t)q * ef + i] = live ? __uint_as_float(u) : PH_FMAX;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ C ABI

#define PH_TRYQ(x)         \
  do {                     \
    int rc__ = (x);        \
    if (rc__) return rc__; \
  } while (0)

// Codebooks: random_centroids (pq.rs:261-285) when kmeans_iters == 0, else Lloyd iterations from that
// start over the first min(n, sample) vectors of the same shuffle (per-sub-space
// k-means, own implementation -- the reference's linfa k-means is dead code, pq.rs:215-259).
// Quantizer::quantize for every vector, split over the ranks of `comm` (pq.rs:326-333 runs it as
// one par_iter over the vectors): rank r encodes the vector range [r*chunk, (r+1)*chunk) into its place of the
// code array, the ranges are all-gathered (n x code bytes in all), every rank ends with the full array.  An
// emulated world computes the ranges in turn.  `encode(first, count, dst)` writes the codes of `count` vectors.
template <class Encode>
static int pq_encode_sharded(const phnsw_comm *comm, uint64_t n, uint64_t row_bytes, void **codes_io, Encode &&encode) {
  const uint32_t w = (comm && comm->world > 1) ? comm->world : 1;
  if (w == 1) return encode((uint64_t)0, n, *codes_io);
  if (comm->rank >= w || (!comm->all_gather && !comm->emulate)) {
    ph_set_error("sharded PQ encode: rank %u of world %u without a transport", comm->rank, w);
    return PHNSW_E_INVALID;
  }
  uint64_t chunk, first, count;
  ph_comm_range(comm, comm->rank, n, &chunk, &first, &count);
  // the code array is re-allocated padded to world * chunk rows so that it IS the receive buffer
  void *padded = nullptr, *send = nullptr;
  PH_HIP(hipMalloc(&padded, (size_t)w * chunk * row_bytes));
  int rc = 0;
  if (!comm->all_gather) {
    for (uint32_t r = 0; r < w && !rc; r++) {
      ph_comm_range(comm, r, n, &chunk, &first, &count);
      if (count) rc = encode(first, count, (char *)padded + first * row_bytes);
    }
  } else {
    hipError_t e = hipMalloc(&send, (size_t)chunk * row_bytes);
    if (e != hipSuccess) rc = ph_hip_fail(e, "pq send block", __FILE__, __LINE__);
    if (!rc && count) rc = encode(first, count, send);
    if (!rc) rc = ph_comm_all_gather_device(comm, send, padded, chunk * row_bytes);
    if (send) hipFree(send);
  }
  if (rc) {
    hipFree(padded);
    return rc;
  }
  hipFree(*codes_io);
  *codes_io = padded;
  return 0;
}

extern "C" int phnsw_store_create_pq_sharded(phnsw_store *full, uint32_t m, uint32_t ksub, uint64_t seed,
                                             uint32_t kmeans_iters, uint64_t sample, const phnsw_comm *comm,
                                             phnsw_store **out) try {
  if (!full || !out || full->codes || !full->rows || m == 0 || ksub == 0 || ksub > 256 || (m % 4) ||
      (full->dim % m) || ksub > full->n) {
    ph_set_error("phnsw_store_create_pq: need an f32 store, m %% 4 == 0, dim %% m == 0, 1 <= ksub <= min(256, n)");
    return PHNSW_E_INVALID;
  }
  if ((size_t)m * ksub * 4 > 150 * 1024) {
    ph_set_error("phnsw_store_create_pq: table m*ksub*4 = %zu bytes does not fit the 160 KiB LDS of a CU",
                 (size_t)m * ksub * 4);
    return PHNSW_E_UNSUPPORTED;
  }
  if (kmeans_iters && full->dim / m > 256) {
    ph_set_error("phnsw_store_create_pq_kmeans: sub-vectors of more than 256 floats are not supported");
    return PHNSW_E_UNSUPPORTED;
  }
  PH_HIP(hipSetDevice(full->device));
  const uint32_t dsub = full->dim / m;
  phnsw_store *s = new phnsw_store();
  s->device = full->device;
  s->n = full->n;
  s->dim = full->dim;
  s->ld = full->ld;
  s->metric = full->metric;
  s->rows = nullptr;
  s->pq_m = m;
  s->pq_ksub = ksub;
  s->pq_dsub = dsub;
  uint64_t S = kmeans_iters ? ((sample && sample < s->n) ? sample : s->n) : ksub;
  if (S < ksub) S = ksub;
  uint32_t *sample_d = nullptr;
  float *train = nullptr;
  uint8_t *tcT = nullptr;
  int rc = 0;
  hipError_t e = hipMalloc(&s->codes, (size_t)s->n * m);
  if (e == hipSuccess) e = hipMalloc(&s->codebook, (size_t)m * ksub * dsub * 4);
  if (e == hipSuccess) e = hipMalloc(&sample_d, (size_t)S * 4);
  if (e != hipSuccess) rc = ph_hip_fail(e, "pq alloc", __FILE__, __LINE__);
  if (!rc) {
    std::vector<uint64_t> perm(s->n);
    for (uint64_t i = 0; i < s->n; i++) perm[i] = i;
    ph_shuffle_u64(perm.data(), s->n, seed ^ 0x9C0DEB00C5ULL);
    std::vector<uint32_t> smp(S);
    for (uint64_t k = 0; k < S; k++) smp[k] = (uint32_t)perm[k];
    e = hipMemcpy(sample_d, smp.data(), (size_t)S * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = ph_hip_fail(e, "pq sample upload", __FILE__, __LINE__);
  }
  if (!rc) {
    uint32_t total = m * ksub * dsub;
    hipLaunchKernelGGL(ph_pq_gather_codebook_kernel, dim3((total + 255) / 256), dim3(256), 0, 0, full->rows, full->ld,
                       sample_d, m, ksub, dsub, s->codebook);
    e = hipGetLastError();
    if (e != hipSuccess) rc = ph_hip_fail(e, "pq codebook", __FILE__, __LINE__);
  }
  if (!rc && kmeans_iters) {
    e = hipMalloc(&train, (size_t)S * full->ld * 4);
    if (e == hipSuccess) e = hipMalloc(&tcT, (size_t)S * m);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(ph_pq_gather_train_kernel, dim3((uint32_t)((S + 3) / 4)), dim3(256), 0, 0, full->rows, full->ld,
                         sample_d, (uint32_t)S, train);
      for (uint32_t it = 0; it < kmeans_iters; it++) {
        hipLaunchKernelGGL(ph_pq_encode_kernel, dim3((uint32_t)std::min<uint64_t>(S, 256u * 32u)), dim3(64), 0, 0, train,
                           full->ld, S, m, ksub, dsub, s->codebook, tcT, (uint64_t)1, S);
        hipLaunchKernelGGL(ph_pq_kmeans_update_kernel, dim3(m * ksub), dim3(64), 0, 0, train, full->ld, (uint32_t)S, m, ksub,
                           dsub, tcT, s->codebook);
      }
      e = hipGetLastError();
    }
    if (e != hipSuccess) rc = ph_hip_fail(e, "pq k-means", __FILE__, __LINE__);
  }
  if (!rc) {
    // codebooks are replicated (computed identically on every rank from the seed); the encode is what shards
    rc = pq_encode_sharded(comm, s->n, m, (void **)&s->codes, [&](uint64_t first, uint64_t count, void *dst) {
      uint32_t grid = (uint32_t)std::min<uint64_t>(count, 256u * 32u);
      hipLaunchKernelGGL(ph_pq_encode_kernel, dim3(grid), dim3(64), 0, 0, full->rows + first * full->ld, full->ld, count, m,
                         ksub, dsub, s->codebook, (uint8_t *)dst, (uint64_t)m, (uint64_t)1);
      hipError_t e2 = hipGetLastError();
      if (e2 == hipSuccess) e2 = hipDeviceSynchronize();
      return e2 == hipSuccess ? 0 : ph_hip_fail(e2, "pq encode", __FILE__, __LINE__);
    });
  }
  if (sample_d) hipFree(sample_d);
  if (train) hipFree(train);
  if (tcT) hipFree(tcT);
  if (rc) {
    if (s->codes) hipFree(s->codes);
    if (s->codebook) hipFree(s->codebook);
    delete s;
    return rc;
  }
  *out = s;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_store_create_pq_kmeans(phnsw_store *full, uint32_t m, uint32_t ksub, uint64_t seed,
                                            uint32_t kmeans_iters, uint64_t sample, phnsw_store **out) try {
  return phnsw_store_create_pq_sharded(full, m, ksub, seed, kmeans_iters, sample, nullptr, out);
} catch (...) { return ph_caught(); }

extern "C" int phnsw_store_create_pq(phnsw_store *full, uint32_t m, uint32_t ksub, uint64_t seed, phnsw_store **out) try {
  return phnsw_store_create_pq_sharded(full, m, ksub, seed, 0, 0, nullptr, out);
} catch (...) { return ph_caught(); }

// ------------------------------------------------------------------ the reference's quantizer shape
// pq.rs:19-27, 61-81, 261-364: ONE codebook of centroid sub-vectors shared by every sub-space, u16 codes
// (up to 65 535 centroids -- a per-query table would be m x 65 535 entries, which is why the reference
// quantises with an HNSW over the centroids and why distances here go through the codes, DistPQS).
__global__ void ph_ids_to_u16_kernel(const uint32_t *ids, uint64_t n, uint16_t *out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = (uint16_t)ids[i];
}

__global__ void ph_pq_shared_reconstruct_kernel(const uint16_t *codes, uint64_t n, uint32_t m, uint32_t dsub,
                                                const float *codebook, float *out, uint32_t ld) {
  const uint64_t dim = (uint64_t)m * dsub;
  for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n * dim; x += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = x / dim;
    const uint32_t c = (uint32_t)(x % dim), j = c / dsub, e = c % dsub;
    out[i * ld + c] = codebook[(uint64_t)codes[i * m + j] * dsub + e];
  }
}

// HnswQuantizer::quantize (pq.rs:61-71) for n vectors [n][m * dsub] resident on the device: every sub-vector is
// a query against the HNSW over the centroids (quantized_search parameters), its best result is the code
static int pq_shared_encode_device(const phnsw_store *s, const float *rows_dev, uint64_t n, uint16_t *codes_dev) {
  const uint64_t total = n * s->pq_m, CH = 1ull << 22;
  const uint64_t cap = std::min(total, CH);
  uint32_t *oid = nullptr, *olen = nullptr, *ostat = nullptr;
  float *od = nullptr;
  hipError_t e = hipMalloc(&oid, cap * 4);
  if (e == hipSuccess) e = hipMalloc(&od, cap * 4);
  if (e == hipSuccess) e = hipMalloc(&olen, cap * 4);
  if (e == hipSuccess) e = hipMalloc(&ostat, cap * 4);
  int rc = e == hipSuccess ? 0 : ph_hip_fail(e, "pq quantize staging", __FILE__, __LINE__);
  std::vector<uint32_t> h_status;
  for (uint64_t c0 = 0; !rc && c0 < total; c0 += CH) {
    const uint64_t cnt = std::min(CH, total - c0);
    rc = ph_search_device(s->centroid_index, rows_dev + c0 * s->pq_dsub, s->pq_dsub, nullptr, cnt, &s->quantized_search, 0,
                          nullptr, oid, od, olen, nullptr, ostat, 0, 0, 0, /*out_stride=*/1);
    if (rc) break;
    hipLaunchKernelGGL(ph_ids_to_u16_kernel, dim3(1024), dim3(256), 0, 0, oid, cnt, codes_dev + c0);
    h_status.resize(cnt);
    e = hipMemcpy(h_status.data(), ostat, cnt * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = ph_hip_fail(e, "pq quantize", __FILE__, __LINE__);
    for (uint64_t i = 0; !rc && i < cnt; i++)
      if (h_status[i]) {
        ph_set_error("pq quantize: sub-vector %llu failed with status %u", (unsigned long long)(c0 + i), h_status[i]);
        rc = h_status[i] == 4 ? PHNSW_E_MISSING_NODE : PHNSW_E_OVERFLOW;
      }
  }
  for (void *p : {(void *)oid, (void *)od, (void *)olen, (void *)ostat})
    if (p) hipFree(p);
  return rc;
}

extern "C" int phnsw_store_create_pq_shared_sharded(phnsw_store *full, uint32_t dsub, uint32_t n_centroids, uint64_t seed,
                                                    const phnsw_build_params *centroid_bp,
                                                    const phnsw_search_params *quantized_search, int centroid_metric,
                                                    const phnsw_comm *comm, phnsw_store **out) try {
  if (!full || !out || !full->rows || !centroid_bp || !quantized_search || dsub == 0 || (dsub % 4) || (full->dim % dsub) ||
      full->ld != full->dim || n_centroids == 0 || n_centroids > 65535 || n_centroids > full->n ||
      quantized_search->number_of_candidates == 0 || quantized_search->number_of_candidates > 1024 ||
      quantized_search->probe_depth == 0 || centroid_metric < 0 || centroid_metric > 2) {
    ph_set_error("phnsw_store_create_pq_shared: need an f32 store with dim %% 4 == 0, dsub %% 4 == 0 dividing dim, "
                 "1 <= n_centroids <= min(65535, n), valid parameters");
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(full->device));
  const uint32_t m = full->dim / dsub;
  // random_centroids  pq.rs:261-285: the sub-vectors of n_centroids selected vectors, sorted, de-duplicated,
  // shuffled, truncated
  std::vector<uint64_t> perm(full->n);
  for (uint64_t i = 0; i < full->n; i++) perm[i] = i;
  ph_shuffle_u64(perm.data(), full->n, seed ^ 0x9C0DEB00C5ULL);
  std::vector<uint32_t> sel(n_centroids);
  for (uint32_t k = 0; k < n_centroids; k++) sel[k] = (uint32_t)perm[k];
  std::vector<float> picked((size_t)n_centroids * full->ld);
  {
    uint32_t *sel_d = nullptr;
    float *rows_d = nullptr;
    hipError_t e = hipMalloc(&sel_d, (size_t)n_centroids * 4);
    if (e == hipSuccess) e = hipMalloc(&rows_d, picked.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(sel_d, sel.data(), (size_t)n_centroids * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(ph_pq_gather_train_kernel, dim3((n_centroids + 3) / 4), dim3(256), 0, 0, full->rows, full->ld, sel_d,
                         n_centroids, rows_d);
      e = hipMemcpy(picked.data(), rows_d, picked.size() * 4, hipMemcpyDeviceToHost);
    }
    if (sel_d) hipFree(sel_d);
    if (rows_d) hipFree(rows_d);
    if (e != hipSuccess) return ph_hip_fail(e, "pq centroid selection", __FILE__, __LINE__);
  }
  const size_t cand = (size_t)n_centroids * m;
  std::vector<uint64_t> order(cand);
  for (size_t i = 0; i < cand; i++) order[i] = i;
  auto subv = [&](uint64_t c) { return picked.data() + (c / m) * full->ld + (c % m) * dsub; };
  auto less = [&](uint64_t a, uint64_t b) {  // OrderedFloat order component by component (pq.rs:277)
    const float *x = subv(a), *y = subv(b);
    for (uint32_t e = 0; e < dsub; e++) {
      if (x[e] < y[e]) return true;
      if (x[e] > y[e]) return false;
    }
    return false;
  };
  std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return less(a, b) || (!less(b, a) && a < b); });
  order.erase(std::unique(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return !less(a, b) && !less(b, a); }),
              order.end());
  ph_shuffle_u64(order.data(), order.size(), seed ^ 0x5EEDCE17ULL);
  const uint32_t C = (uint32_t)std::min<size_t>(order.size(), n_centroids);
  std::vector<float> cb((size_t)C * dsub);
  for (uint32_t k = 0; k < C; k++) memcpy(cb.data() + (size_t)k * dsub, subv(order[k]), (size_t)dsub * 4);

  phnsw_store *s = new phnsw_store();
  s->device = full->device;
  s->n = full->n;
  s->dim = full->dim;
  s->ld = full->ld;
  s->metric = full->metric;
  s->rows = nullptr;
  s->pq_m = m;
  s->pq_ksub = C;
  s->pq_dsub = dsub;
  s->quantized_search = *quantized_search;
  auto fail = [&](int rc) {
    phnsw_store_destroy(s);
    return rc;
  };
  // CentroidComparator + Hnsw::generate over the centroids + improve_index  pq.rs:306-312
  int rc = phnsw_store_create(cb.data(), C, dsub, centroid_metric, full->device, &s->centroid_store);
  if (rc) return fail(rc);
  std::vector<uint64_t> cids(C);
  for (uint32_t k = 0; k < C; k++) cids[k] = k;
  rc = phnsw_build(s->centroid_store, cids.data(), C, centroid_bp, nullptr, nullptr, &s->centroid_index);
  if (rc) return fail(rc);
  rc = phnsw_improve_index(s->centroid_index, centroid_bp, NAN, nullptr, nullptr, nullptr);
  if (rc) return fail(rc);
  hipError_t e = hipMalloc(&s->codebook, cb.size() * 4);
  if (e == hipSuccess) e = hipMemcpy(s->codebook, cb.data(), cb.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&s->codes16, (size_t)s->n * m * 2);
  if (e != hipSuccess) return fail(ph_hip_fail(e, "pq shared alloc", __FILE__, __LINE__));
  // centroid_quantizer.quantize(&v) for every vector  pq.rs:326-333 -- over the ranks of comm; the centroid index
  // above is replicated (built identically on every rank: 65 535 x 16 floats at most)
  rc = pq_encode_sharded(comm, s->n, (uint64_t)m * 2, (void **)&s->codes16, [&](uint64_t first, uint64_t count, void *dst) {
    int r2 = pq_shared_encode_device(s, full->rows + first * full->ld, count, (uint16_t *)dst);
    if (r2) return r2;
    hipError_t e2 = hipDeviceSynchronize();
    return e2 == hipSuccess ? 0 : ph_hip_fail(e2, "pq shared encode", __FILE__, __LINE__);
  });
  if (rc) return fail(rc);
  *out = s;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_store_create_pq_shared(phnsw_store *full, uint32_t dsub, uint32_t n_centroids, uint64_t seed,
                                            const phnsw_build_params *centroid_bp,
                                            const phnsw_search_params *quantized_search, int centroid_metric,
                                            phnsw_store **out) try {
  return phnsw_store_create_pq_shared_sharded(full, dsub, n_centroids, seed, centroid_bp, quantized_search, centroid_metric,
                                              nullptr, out);
} catch (...) { return ph_caught(); }

extern "C" int phnsw_pq_shared_read(const phnsw_store *s, uint16_t *codes, float *codebook) try {
  if (!s || !s->codes16) {
    ph_set_error("not a shared-codebook product-quantised store");
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(s->device));
  if (codes) PH_HIP(hipMemcpy(codes, s->codes16, (size_t)s->n * s->pq_m * 2, hipMemcpyDeviceToHost));
  if (codebook) PH_HIP(hipMemcpy(codebook, s->codebook, (size_t)s->pq_ksub * s->pq_dsub * 4, hipMemcpyDeviceToHost));
  return 0;
} catch (...) { return ph_caught(); }

// the reconstructions of a shared-codebook store as an f32 store of their own: distances over it have the same
// bits as over the codes (DistPQS), so the Hnsw over the quantised vectors (pq.rs:336-338) is built on it with
// the full-precision kernels and adopted over the codes (phnsw_index_from_layers); destroy it afterwards
extern "C" int phnsw_pq_shared_reconstruct_store(const phnsw_store *s, phnsw_store **out) try {
  if (!s || !s->codes16 || !out) {
    ph_set_error("phnsw_pq_shared_reconstruct_store: needs a shared-codebook product-quantised store");
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(s->device));
  float *rows = nullptr;
  PH_HIP(hipMalloc(&rows, (size_t)s->n * s->ld * 4));
  hipLaunchKernelGGL(ph_pq_shared_reconstruct_kernel, dim3(4096), dim3(256), 0, 0, s->codes16, s->n, s->pq_m, s->pq_dsub,
                     s->codebook, rows, s->ld);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    hipFree(rows);
    return ph_hip_fail(e, "pq shared reconstruct", __FILE__, __LINE__);
  }
  phnsw_store *r = new phnsw_store();
  r->device = s->device;
  r->rows = rows;
  r->owns_rows = true;
  r->n = s->n;
  r->dim = s->dim;
  r->ld = s->ld;
  r->metric = s->metric;
  *out = r;
  return 0;
} catch (...) { return ph_caught(); }

// How the per-query lookup table T[m][ksub] is stored.  0: f32, the
// reference arithmetic.  1: every entry rounded once to IEEE half.  2: 8-bit entries
// u = rint((T - min_row) / scale), scale = widest row range / 255; a distance is then
// bias + scale * (exact integer sum), bias = sum of the row minima.  Modes 1 and 2 change the
// quantised distances; a graph must be built and searched in the same mode.  The oracle mirrors
// every mode bit for bit.
extern "C" int phnsw_pq_set_table_mode(phnsw_store *s, int mode) try {
  if (!s || !s->codes || mode < 0 || mode > 2) {
    ph_set_error("phnsw_pq_set_table_mode: needs a product-quantised store and mode 0 (f32), 1 (f16) or 2 (8-bit)");
    return PHNSW_E_INVALID;
  }
  if (mode == 2 && s->pq_m > 128) {
    ph_set_error("8-bit tables support at most 128 sub-spaces (got %u)", s->pq_m);
    return PHNSW_E_UNSUPPORTED;
  }
  s->pq_table_f16 = (uint32_t)mode;
  return 0;
} catch (...) { return ph_caught(); }
extern "C" int phnsw_pq_set_table_f16(phnsw_store *s, int on) try { return phnsw_pq_set_table_mode(s, on ? 1 : 0); } catch (...) { return ph_caught(); }

// Quantizer::quantize  pq.rs:61-71 for arbitrary vectors (exact nearest centroid per sub-space,
// ties to the smaller centroid id) and Quantizer::reconstruct  pq.rs:73-81
extern "C" int phnsw_pq_quantize(const phnsw_store *s, const float *rows, uint64_t n, uint8_t *out_codes) try {
  if (!s || !s->codes || !rows || !out_codes) {
    ph_set_error("phnsw_pq_quantize: needs a product-quantised store, rows and an output buffer");
    return PHNSW_E_INVALID;
  }
  if (n == 0) return 0;
  PH_HIP(hipSetDevice(s->device));
  const uint32_t dim = s->dim;
  float *rd = nullptr;
  uint8_t *cd = nullptr;
  hipError_t e = hipMalloc(&rd, (size_t)n * dim * 4);
  if (e == hipSuccess) e = hipMalloc(&cd, (size_t)n * s->pq_m);
  if (e == hipSuccess) e = hipMemcpy(rd, rows, (size_t)n * dim * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    uint32_t grid = (uint32_t)std::min<uint64_t>(n, 256u * 32u);
    hipLaunchKernelGGL(ph_pq_encode_kernel, dim3(grid), dim3(64), 0, 0, rd, dim, n, s->pq_m, s->pq_ksub, s->pq_dsub,
                       s->codebook, cd, (uint64_t)s->pq_m, (uint64_t)1);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out_codes, cd, (size_t)n * s->pq_m, hipMemcpyDeviceToHost);
  if (rd) hipFree(rd);
  if (cd) hipFree(cd);
  if (e != hipSuccess) return ph_hip_fail(e, "pq quantize", __FILE__, __LINE__);
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_pq_reconstruct(const phnsw_store *s, const uint8_t *codes, uint64_t n, float *out_rows) try {
  if (!s || !s->codes || !codes || !out_rows) {
    ph_set_error("phnsw_pq_reconstruct: needs a product-quantised store, codes and an output buffer");
    return PHNSW_E_INVALID;
  }
  if (n == 0) return 0;
  PH_HIP(hipSetDevice(s->device));
  const uint32_t dim = s->dim;
  float *rd = nullptr;
  uint8_t *cd = nullptr;
  hipError_t e = hipMalloc(&rd, (size_t)n * dim * 4);
  if (e == hipSuccess) e = hipMalloc(&cd, (size_t)n * s->pq_m);
  if (e == hipSuccess) e = hipMemcpy(cd, codes, (size_t)n * s->pq_m, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    uint64_t total = n * dim;
    hipLaunchKernelGGL(ph_pq_reconstruct_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, 0, cd, n, s->pq_m,
                       s->pq_ksub, s->pq_dsub, s->codebook, rd, dim);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out_rows, rd, (size_t)n * dim * 4, hipMemcpyDeviceToHost);
  if (rd) hipFree(rd);
  if (cd) hipFree(cd);
  if (e != hipSuccess) return ph_hip_fail(e, "pq reconstruct", __FILE__, __LINE__);
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_pq_info(const phnsw_store *s, uint32_t *m, uint32_t *ksub, uint32_t *dsub) try {
  if (!s || (!s->codes && !s->codes16)) {
    ph_set_error("not a product-quantised store");
    return PHNSW_E_INVALID;
  }
  if (m) *m = s->pq_m;
  if (ksub) *ksub = s->pq_ksub;
  if (dsub) *dsub = s->pq_dsub;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_pq_read(const phnsw_store *s, uint8_t *codes, float *codebook) try {
  if (!s || !s->codes) {
    ph_set_error("not a product-quantised store");
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(s->device));
  if (codes) PH_HIP(hipMemcpy(codes, s->codes, (size_t)s->n * s->pq_m, hipMemcpyDeviceToHost));
  if (codebook)
    PH_HIP(hipMemcpy(codebook, s->codebook, (size_t)s->pq_m * s->pq_ksub * s->pq_dsub * 4, hipMemcpyDeviceToHost));
  return 0;
} catch (...) { return ph_caught(); }

// QuantizedHnsw::search  pq.rs:346-364 for a batch: (optionally quantise the query like the
// reference :351-352, default = asymmetric: the raw query meets the codes), search the graph
// over the code rows, re-rank with the full-precision store, sort by (d, id).
extern "C" int phnsw_pq_search_batch(const phnsw_index *ix, const phnsw_store *full, const float *queries, uint64_t nq,
                                     const phnsw_search_params *sp, int quantize_query, uint64_t *out_ids,
                                     float *out_d, uint64_t *out_len, uint64_t *out_stats) try {
  if (!ix || !full || !queries || !sp || !out_ids || !out_d || !out_len || (!ix->store->codes && !ix->store->codes16) || !full->rows ||
      full->n != ix->store->n || full->dim != ix->store->dim || nq > 0xFFFFFFFFull || sp->number_of_candidates == 0 ||
      sp->number_of_candidates > 1024 || sp->probe_depth == 0) {
    ph_set_error("phnsw_pq_search_batch: need an index over a PQ store, its full-precision store and valid parameters");
    return PHNSW_E_INVALID;
  }
  if (nq == 0) return 0;
  const phnsw_store *ps = ix->store;
  PH_HIP(hipSetDevice(ps->device));
  const uint32_t ef = (uint32_t)sp->number_of_candidates;
  float *qd = nullptr, *qq = nullptr, *od = nullptr;
  uint32_t *oid = nullptr, *olen = nullptr, *ost = nullptr, *ostat = nullptr;
  uint8_t *qcodes = nullptr;
  int rc = 0;
  hipError_t e = hipMalloc(&qd, (size_t)nq * full->ld * 4);
  if (e == hipSuccess && full->ld != full->dim) e = hipMemset(qd, 0, (size_t)nq * full->ld * 4);
  if (e == hipSuccess)
    e = hipMemcpy2D(qd, (size_t)full->ld * 4, queries, (size_t)full->dim * 4, (size_t)full->dim * 4, nq,
                    hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&oid, (size_t)nq * ef * 4);
  if (e == hipSuccess) e = hipMalloc(&od, (size_t)nq * ef * 4);
  if (e == hipSuccess) e = hipMalloc(&olen, nq * 4);
  if (e == hipSuccess) e = hipMalloc(&ost, nq * 8);
  if (e == hipSuccess) e = hipMalloc(&ostat, nq * 4);
  const float *qsearch = qd;
  if (e == hipSuccess && quantize_query && ps->codes16) {
    // quantizer.quantize(&raw_v) through the HNSW over the centroids, then its reconstruction  pq.rs:351-353
    uint16_t *qc16 = nullptr;
    e = hipMalloc(&qc16, (size_t)nq * ps->pq_m * 2);
    if (e == hipSuccess) e = hipMalloc(&qq, (size_t)nq * full->ld * 4);
    if (e == hipSuccess) {
      rc = pq_shared_encode_device(ps, qd, nq, qc16);
      if (!rc) {
        hipLaunchKernelGGL(ph_pq_shared_reconstruct_kernel, dim3(1024), dim3(256), 0, 0, qc16, nq, ps->pq_m, ps->pq_dsub,
                           ps->codebook, qq, full->ld);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        qsearch = qq;
      }
    }
    if (qc16) hipFree(qc16);
  } else if (e == hipSuccess && quantize_query) {
    e = hipMalloc(&qcodes, (size_t)nq * ps->pq_m);
    if (e == hipSuccess) e = hipMalloc(&qq, (size_t)nq * full->ld * 4);
    if (e == hipSuccess) e = hipMemset(qq, 0, (size_t)nq * full->ld * 4);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(ph_pq_encode_kernel, dim3((uint32_t)std::min<uint64_t>(nq, 8192)), dim3(64), 0, 0, qd,
                         full->ld, nq, ps->pq_m, ps->pq_ksub, ps->pq_dsub, ps->codebook, qcodes, (uint64_t)ps->pq_m,
                         (uint64_t)1);
      uint64_t tot = nq * full->dim;
      hipLaunchKernelGGL(ph_pq_reconstruct_kernel, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, 0, qcodes, nq,
                         ps->pq_m, ps->pq_ksub, ps->pq_dsub, ps->codebook, qq, full->ld);
      e = hipGetLastError();
      qsearch = qq;
    }
  }
  if (e != hipSuccess && !rc) rc = ph_hip_fail(e, "pq search staging", __FILE__, __LINE__);
  if (!rc)
    rc = ph_search_device(ix, qsearch, full->ld, nullptr, nq, sp, 0, nullptr, oid, od, olen, ost, ostat, 0, 0, 0);
  std::vector<uint32_t> h_status(nq);
  if (!rc) {
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(h_status.data(), ostat, nq * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = ph_hip_fail(e, "pq search", __FILE__, __LINE__);
    for (uint64_t i = 0; !rc && i < nq; i++)
      if (h_status[i]) {
        ph_set_error("pq search: query %llu failed with status %u", (unsigned long long)i, h_status[i]);
        rc = h_status[i] == 4 ? PHNSW_E_MISSING_NODE : PHNSW_E_OVERFLOW;
      }
  }
  if (!rc) {
    // full_comparator().compare_vec(Stored(id), v) for every result, sort_by_key (d, id)
    PhDistArgs fa = ph_dist_args(full);
    uint32_t grid = (uint32_t)std::min<uint64_t>(nq, 256u * 16u);
    size_t lds = (size_t)ef * 16;
    uint32_t nv4 = full->ld / 4;
    if (nv4 <= 64)
      hipLaunchKernelGGL(ph_pq_rerank_kernel<1>, dim3(grid), dim3(64), lds, 0, fa, qd, full->ld, (uint32_t)nq, ef, olen, oid, od);
    else if (nv4 <= 192)
      hipLaunchKernelGGL(ph_pq_rerank_kernel<3>, dim3(grid), dim3(64), lds, 0, fa, qd, full->ld, (uint32_t)nq, ef, olen, oid, od);
    else if (nv4 <= 384)
      hipLaunchKernelGGL(ph_pq_rerank_kernel<6>, dim3(grid), dim3(64), lds, 0, fa, qd, full->ld, (uint32_t)nq, ef, olen, oid, od);
    else {
      ph_set_error("dim %u unsupported (max 1536)", full->dim);
      rc = PHNSW_E_UNSUPPORTED;
    }
    if (!rc) {
      e = hipGetLastError();
      if (e == hipSuccess) e = hipDeviceSynchronize();
      if (e != hipSuccess) rc = ph_hip_fail(e, "pq rerank", __FILE__, __LINE__);
    }
  }
  if (!rc) {
    std::vector<uint32_t> h_ids((size_t)nq * ef), h_len(nq), h_st(2 * nq);
    e = hipMemcpy(h_ids.data(), oid, h_ids.size() * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_d, od, (size_t)nq * ef * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(h_len.data(), olen, nq * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(h_st.data(), ost, nq * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
      rc = ph_hip_fail(e, "pq readback", __FILE__, __LINE__);
    else {
      for (size_t i = 0; i < h_ids.size(); i++) out_ids[i] = h_ids[i] == PH_EMPTY32 ? PHNSW_EMPTY : h_ids[i];
      for (uint64_t i = 0; i < nq; i++) out_len[i] = h_len[i];
      if (out_stats)
        for (uint64_t i = 0; i < 2 * nq; i++) out_stats[i] = h_st[i];
    }
  }
  for (void *p : {(void *)qd, (void *)qq, (void *)od, (void *)oid, (void *)olen, (void *)ost, (void *)ostat, (void *)qcodes})
    if (p) hipFree(p);
  return rc;
} catch (...) { return ph_caught(); }

// zero-copy form of phnsw_pq_search_batch (asymmetric queries): search kernel + re-rank
// kernel enqueued on `stream`, u32 ids, no synchronisation.
extern "C" int phnsw_pq_search_batch_device(const phnsw_index *ix, const phnsw_store *full, const float *queries_dev,
                                            uint32_t ldq, uint64_t nq, const phnsw_search_params *sp,
                                            uint32_t *out_ids_dev, float *out_d_dev, uint32_t *out_len_dev,
                                            uint32_t *out_stats_dev, uint32_t *status_dev, void *stream) try {
  if (!ix || !full || !queries_dev || !sp || !out_ids_dev || !out_d_dev || !out_len_dev || !status_dev ||
      (!ix->store->codes && !ix->store->codes16) || !full->rows || full->n != ix->store->n || full->dim != ix->store->dim ||
      nq > 0xFFFFFFFFull || sp->number_of_candidates == 0 || sp->number_of_candidates > 1024 || sp->probe_depth == 0 ||
      ldq < full->ld || (ldq % 4)) {
    ph_set_error("phnsw_pq_search_batch_device: invalid argument");
    return PHNSW_E_INVALID;
  }
  if (nq == 0) return 0;
  PH_HIP(hipSetDevice(full->device));
  const uint32_t ef = (uint32_t)sp->number_of_candidates;
  PH_TRYQ(ph_search_device(ix, queries_dev, ldq, nullptr, nq, sp, 0, nullptr, out_ids_dev, out_d_dev, out_len_dev,
                           out_stats_dev, status_dev, 0, 0, (hipStream_t)stream));
  PhDistArgs fa = ph_dist_args(full);
  uint32_t grid = (uint32_t)std::min<uint64_t>(nq, 256u * 16u);
  size_t lds = (size_t)ef * 16;
  uint32_t nv4 = full->ld / 4;
  hipStream_t st = (hipStream_t)stream;
  if (nv4 <= 64)
    hipLaunchKernelGGL(ph_pq_rerank_kernel<1>, dim3(grid), dim3(64), lds, st, fa, queries_dev, ldq, (uint32_t)nq, ef, out_len_dev, out_ids_dev, out_d_dev);
  else if (nv4 <= 192)
    hipLaunchKernelGGL(ph_pq_rerank_kernel<3>, dim3(grid), dim3(64), lds, st, fa, queries_dev, ldq, (uint32_t)nq, ef, out_len_dev, out_ids_dev, out_d_dev);
  else if (nv4 <= 384)
    hipLaunchKernelGGL(ph_pq_rerank_kernel<6>, dim3(grid), dim3(64), lds, st, fa, queries_dev, ldq, (uint32_t)nq, ef, out_len_dev, out_ids_dev, out_d_dev);
  else {
    ph_set_error("dim %u unsupported (max 1536)", full->dim);
    return PHNSW_E_UNSUPPORTED;
  }
  PH_HIP(hipGetLastError());
  return 0;
} catch (...) { return ph_caught(); }
