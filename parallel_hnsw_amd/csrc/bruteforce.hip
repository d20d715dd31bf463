// G1: exact k nearest neighbours by brute force -- the ground truth for recall@k
// (the reference has no such routine, it only measures
// self-recall, /root/reference/src/lib.rs:1485-1496).
//
// This is the one place on the path where a batched query x candidate block IS a true GEMM
// (north_star): scores[q][b] = <query q, base row b>, M = queries, N = base rows, K = dim.
// It runs on the matrix cores with the f32-input MFMA (v_mfma_f32_32x32x2_f32): exact f32, and
// bit for bit a k-ordered fma chain, i.e. the oracle's ORC_SUM_SEQFMA order, so the result
// can be checked exactly.  Peak 157 TFLOP/s (MI355X_MICROARCH.md); a 10 000 x 1M x 768 pass
// is 15.4 TFLOP.
//
//   ph_gemm_nt_mfma_kernel  128 queries x 128 base rows per 256-thread block, K in chunks of
//                           32 staged through LDS k-major (conflict-free fragment reads),
//                           each wave 2x2 tiles of 32x32, scores written in 128-B segments
//   ph_topk_chunk_kernel    one wave per query: running top-k by (distance, id) over a chunk
//                           of scores, merged with the result of the previous chunks
//
// Dot-product metrics only ((1-dot)/2 and 1-dot); scores of a chunk of base rows at a time
// (64 Ki rows: 2.6 GB of f32 for 10 000 queries) instead of the full n x nq matrix.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "phnsw_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BF_TM 128   // queries per block
#define BF_TN 128   // base rows per block
#define BF_TK 32    // k per LDS stage
#define BF_LD 132   // LDS row stride in floats (k-major rows of 128 + pad)

__global__ __launch_bounds__(256) void ph_gemm_nt_mfma_kernel(const float *__restrict__ Q, uint32_t ldq, uint32_t nq,
                                                              const float *__restrict__ B, uint32_t ldb, uint32_t nb,
                                                              uint32_t K, float *__restrict__ C, uint64_t ldc) {
  __shared__ float Qs[BF_TK][BF_LD];
  __shared__ float Bs[BF_TK][BF_LD];
  const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
  const uint32_t wm = w >> 1, wn = w & 1;            // wave tile: 64 queries x 64 base rows
  const uint32_t q0 = blockIdx.y * BF_TM, b0 = blockIdx.x * BF_TN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const uint32_t lrow = t >> 3, lk = (t & 7) * 4;    // loader: 32 rows x 8 float4 per pass, 4 passes
  float4 qa[4], ba[4];
  auto load_stage = [&](uint32_t k0) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      uint32_t row = lrow + 32 * p, kk = k0 + lk;
      qa[p] = (q0 + row < nq && kk < K) ? *(const float4 *)(Q + (uint64_t)(q0 + row) * ldq + kk) : make_float4(0, 0, 0, 0);
      ba[p] = (b0 + row < nb && kk < K) ? *(const float4 *)(B + (uint64_t)(b0 + row) * ldb + kk) : make_float4(0, 0, 0, 0);
    }
  };
  load_stage(0);
  for (uint32_t k0 = 0; k0 < K; k0 += BF_TK) {
    __syncthreads();  // the previous stage's fragment reads are done
#pragma unroll
    for (int p = 0; p < 4; p++) {
      uint32_t row = lrow + 32 * p;
      Qs[lk + 0][row] = qa[p].x;
      Qs[lk + 1][row] = qa[p].y;
      Qs[lk + 2][row] = qa[p].z;
      Qs[lk + 3][row] = qa[p].w;
      Bs[lk + 0][row] = ba[p].x;
      Bs[lk + 1][row] = ba[p].y;
      Bs[lk + 2][row] = ba[p].z;
      Bs[lk + 3][row] = ba[p].w;
    }
    __syncthreads();
    if (k0 + BF_TK < K) load_stage(k0 + BF_TK);  // in flight under this stage's MFMAs
    // A operand = queries: lane l supplies A[i = l&31][k = l>>5]; B operand = base rows:
    // B[k = l>>5][j = l&31]; k ascends through the loop => one fma chain per output element
#pragma unroll
    for (int kk = 0; kk < BF_TK; kk += 2) {
      const uint32_t kr = kk + (lane >> 5), c = lane & 31;
      float a0 = Qs[kr][wm * 64 + c], a1 = Qs[kr][wm * 64 + 32 + c];
      float b0_ = Bs[kr][wn * 64 + c], b1 = Bs[kr][wn * 64 + 32 + c];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0_, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0_, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  // D: lane l holds column j = l&31 (base row) and rows i = (r&3) + 8*(r>>2) + 4*(l>>5) (queries)
#pragma unroll
  for (int mi = 0; mi < 2; mi++)
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        uint32_t qi = q0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        uint32_t bj = b0 + wn * 64 + ni * 32 + (lane & 31);
        if (qi < nq && bj < nb) C[(uint64_t)qi * ldc + bj] = acc[mi][ni][r];
      }
}

// running top-k of one query over a chunk of scores; keys = (distance, id) like the search
#define BF_KMAX 16
__global__ __launch_bounds__(64) void ph_topk_chunk_kernel(const float *__restrict__ scores, uint64_t ldc, uint32_t nb,
                                                           uint32_t base_first, int metric, uint32_t k, uint32_t nq,
                                                           uint64_t *__restrict__ best /* [nq][k] running keys */) {
  __shared__ uint64_t cand[64 * BF_KMAX + BF_KMAX];
  const uint32_t lane = threadIdx.x;
  for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
    uint64_t top[BF_KMAX];
#pragma unroll
    for (int j = 0; j < BF_KMAX; j++) top[j] = KEY_NONE;
    const float *row = scores + (uint64_t)q * ldc;
    for (uint32_t i = lane; i < nb; i += 64) {
      uint64_t key = mkkey(finalize_metric(row[i], metric), base_first + i);
      if (key < top[BF_KMAX - 1]) {  // sorted insert (ascending), fully unrolled
#pragma unroll
        for (int j = 0; j < BF_KMAX; j++) {
          uint64_t lo = key < top[j] ? key : top[j];
          uint64_t hi = key < top[j] ? top[j] : key;
          top[j] = lo;
          key = hi;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < BF_KMAX; j++) cand[lane * BF_KMAX + j] = top[j];
    if (lane < BF_KMAX) cand[64 * BF_KMAX + lane] = lane < k ? best[(uint64_t)q * k + lane] : KEY_NONE;
    __syncthreads();
    const uint32_t total = 64 * BF_KMAX + BF_KMAX;
    for (uint32_t c = lane; c < total; c += 64) {
      uint64_t kc = cand[c];
      if (kc == KEY_NONE) continue;
      uint32_t rank = 0;
      for (uint32_t e = 0; e < total && rank < k; e++) {
        uint64_t ke = cand[e];
        rank += (ke < kc || (ke == kc && e < c)) ? 1u : 0u;
      }
      if (rank < k) best[(uint64_t)q * k + rank] = kc;
    }
    __syncthreads();
  }
}

__global__ void ph_fill_u64_kernel(uint64_t *p, uint64_t v, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void ph_unpack_keys_kernel(const uint64_t *keys, uint64_t n, uint32_t *ids, float *d) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t k = keys[i];
    uint32_t fk = (uint32_t)(k >> 32);
    uint32_t u = (fk & 0x80000000u) ? (fk ^ 0x80000000u) : ~fk;
    ids[i] = k == KEY_NONE ? PH_EMPTY32 : ((uint32_t)k & IDM);
    d[i] = k == KEY_NONE ? PH_FMAX : __uint_as_float(u);
  }
}

static thread_local float g_bf_gemm_ms = 0.f;

// everything on the device: queries_dev [nq][ldq], outputs [nq][k] (u32 ids, f32 distances)
extern "C" int phnsw_bruteforce_topk_device(const phnsw_store *s, const float *queries_dev, uint32_t ldq, uint64_t nq,
                                            uint32_t k, uint32_t *out_ids_dev, float *out_d_dev, void *stream) try {
  if (!s || !s->rows || !queries_dev || !out_ids_dev || !out_d_dev || k == 0 || k > BF_KMAX || k > s->n ||
      nq == 0 || nq > 0xFFFFFFFFull || ldq < s->ld || (ldq % 4) || s->metric == PHNSW_METRIC_L2) {
    ph_set_error("phnsw_bruteforce_topk: need an f32 store with a dot-product metric, 1 <= k <= %d, ldq %% 4 == 0",
                 BF_KMAX);
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  // chunk of base rows whose score matrix stays around 2.5 GB
  uint64_t chunk = std::max<uint64_t>(BF_TN, std::min<uint64_t>(s->n, (uint64_t)(640ull << 20) / nq / BF_TN * BF_TN));
  chunk = std::min<uint64_t>(chunk, 1u << 20);
  float *scores = nullptr;
  uint64_t *best = nullptr;
  hipError_t e = hipMalloc(&scores, (size_t)nq * chunk * 4);
  if (e == hipSuccess) e = hipMalloc(&best, (size_t)nq * k * 8);
  if (e != hipSuccess) {
    if (scores) hipFree(scores);
    return ph_hip_fail(e, "bruteforce alloc", __FILE__, __LINE__);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(ph_fill_u64_kernel, dim3(256), dim3(256), 0, st, best, KEY_NONE, nq * k);
  float gemm_ms = 0.f;
  int rc = 0;
  for (uint64_t first = 0; first < s->n && !rc; first += chunk) {
    uint32_t nb = (uint32_t)std::min<uint64_t>(chunk, s->n - first);
    dim3 grid((nb + BF_TN - 1) / BF_TN, (uint32_t)((nq + BF_TM - 1) / BF_TM));
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(ph_gemm_nt_mfma_kernel, grid, dim3(256), 0, st, queries_dev, ldq, (uint32_t)nq,
                       s->rows + first * s->ld, s->ld, nb, s->ld, scores, chunk);
    hipEventRecord(e1, st);
    hipLaunchKernelGGL(ph_topk_chunk_kernel, dim3((uint32_t)std::min<uint64_t>(nq, 256u * 16u)), dim3(64), 0, st, scores,
                       chunk, nb, (uint32_t)first, s->metric, k, (uint32_t)nq, best);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e != hipSuccess) {
      rc = ph_hip_fail(e, "bruteforce chunk", __FILE__, __LINE__);
      break;
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    gemm_ms += ms;
  }
  if (!rc) {
    hipLaunchKernelGGL(ph_unpack_keys_kernel, dim3(256), dim3(256), 0, st, best, nq * k, out_ids_dev, out_d_dev);
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = ph_hip_fail(e, "bruteforce unpack", __FILE__, __LINE__);
  }
  g_bf_gemm_ms = gemm_ms;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(scores);
  hipFree(best);
  return rc;
} catch (...) { return ph_caught(); }

// ---- coarse cells for the locality schedule (phnsw_internal.h PhLayerHost::pos) ----
__global__ void ph_gather_rows_f32_kernel(const float *rows, uint32_t ld, const uint32_t *ids, uint32_t stride,
                                          uint32_t cnt, float *out) {
  // one wave per row, float4 per lane
  uint32_t r = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
  if (r >= cnt) return;
  const float4 *src = (const float4 *)(rows + (uint64_t)ids[(uint64_t)r * stride] * ld);
  float4 *dst = (float4 *)(out + (uint64_t)r * ld);
  for (uint32_t j = lane; j < ld / 4; j += 64) dst[j] = src[j];
}

// argmax of each score row (largest dot product = nearest for the dot-product metrics)
__global__ __launch_bounds__(64) void ph_argmax_rows_kernel(const float *scores, uint64_t ldc, uint32_t na, uint32_t nq,
                                                            uint32_t *out) {
  const uint32_t lane = threadIdx.x;
  for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
    const float *row = scores + (uint64_t)q * ldc;
    float best = -PH_FMAX;
    uint32_t bi = 0;
    for (uint32_t i = lane; i < na; i += 64) {
      float v = row[i];
      if (v > best) {
        best = v;
        bi = i;
      }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      float ov = __shfl_xor(best, s);
      uint32_t oi = __shfl_xor(bi, s);
      if (ov > best || (ov == best && oi < bi)) {
        best = ov;
        bi = oi;
      }
    }
    if (lane == 0) out[q] = bi;
  }
}

// Greedy nearest-neighbour chain over the anchors: rank[a] = position of anchor a on a path
// that always moves to the most similar anchor not yet visited, so cells that are close in
// space get close ranks.  One 1024-thread block; scores is the A x A anchor Gram matrix.
__global__ __launch_bounds__(1024) void ph_anchor_chain_kernel(const float *scores, uint32_t A, uint32_t *rank) {
  extern __shared__ uint32_t sm[];
  uint32_t *seen = sm;                 // [A] flags
  float *wbest = (float *)(sm + A);    // [16]
  uint32_t *widx = sm + A + 16;        // [16]
  __shared__ uint32_t cur_s;
  const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (uint32_t i = t; i < A; i += 1024) seen[i] = 0;
  if (t == 0) {
    cur_s = 0;
    seen[0] = 1;
    rank[0] = 0;
  }
  __syncthreads();
  for (uint32_t step = 1; step < A; step++) {
    const uint32_t cur = cur_s;
    const float *row = scores + (uint64_t)cur * A;
    float best = -PH_FMAX;
    uint32_t bi = PH_EMPTY32;
    for (uint32_t j = t; j < A; j += 1024)
      if (!seen[j]) {
        float v = row[j];
        if (bi == PH_EMPTY32 || v > best) {
          best = v;
          bi = j;
        }
      }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
      float ov = __shfl_xor(best, sft);
      uint32_t oi = __shfl_xor(bi, sft);
      if (oi != PH_EMPTY32 && (bi == PH_EMPTY32 || ov > best || (ov == best && oi < bi))) {
        best = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      wbest[w] = best;
      widx[w] = bi;
    }
    __syncthreads();
    if (t == 0) {
      float b = -PH_FMAX;
      uint32_t i = PH_EMPTY32;
      for (int k = 0; k < 16; k++)
        if (widx[k] != PH_EMPTY32 && (i == PH_EMPTY32 || wbest[k] > b || (wbest[k] == b && widx[k] < i))) {
          b = wbest[k];
          i = widx[k];
        }
      cur_s = i;
      seen[i] = 1;
      rank[i] = step;
    }
    __syncthreads();
  }
}

__global__ void ph_rank_of_cell_kernel(uint32_t *pos, uint32_t n, const uint32_t *rank) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos[i] = rank[pos[i]];
}

// The store's coarse cells: A anchors (every (n/A)-th row) and their chain ranks, made once.
static int ph_store_anchors(phnsw_store *s) {
  if (s->anchors) return 0;
  uint32_t amax = 4096u;
  if (const char *e = getenv("PHNSW_ANCHORS")) amax = atoi(e) >= 64 ? (uint32_t)atoi(e) : amax;  // tuning knob
  const uint32_t A = (uint32_t)std::min<uint64_t>(amax, std::max<uint64_t>(s->n / 16u, 1u)), ld = s->ld;
  const uint64_t stride = s->n / A;
  float *anchors = nullptr, *gram = nullptr;
  uint32_t *rank = nullptr, *ids = nullptr;
  std::vector<uint32_t> h(A);
  for (uint32_t a = 0; a < A; a++) h[a] = (uint32_t)(a * stride);
  hipError_t e = hipMalloc(&anchors, (size_t)A * ld * 4);
  if (e == hipSuccess) e = hipMalloc(&gram, (size_t)A * A * 4);
  if (e == hipSuccess) e = hipMalloc(&rank, (size_t)A * 4);
  if (e == hipSuccess) e = hipMalloc(&ids, (size_t)A * 4);
  if (e == hipSuccess) e = hipMemcpy(ids, h.data(), (size_t)A * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(ph_gather_rows_f32_kernel, dim3((A + 3) / 4), dim3(256), 0, 0, s->rows, ld, ids, 1u, A, anchors);
    dim3 grid((A + BF_TN - 1) / BF_TN, (A + BF_TM - 1) / BF_TM);
    hipLaunchKernelGGL(ph_gemm_nt_mfma_kernel, grid, dim3(256), 0, 0, anchors, ld, A, anchors, ld, A, ld, gram, (uint64_t)A);
    hipLaunchKernelGGL(ph_anchor_chain_kernel, dim3(1), dim3(1024), (A + 32) * 4, 0, gram, A, rank);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  if (gram) hipFree(gram);
  if (ids) hipFree(ids);
  if (e != hipSuccess) {
    if (anchors) hipFree(anchors);
    if (rank) hipFree(rank);
    return ph_hip_fail(e, "store cells (anchors)", __FILE__, __LINE__);
  }
  s->anchors = anchors;
  s->anchor_rank = rank;
  s->n_anchors = A;
  return 0;
}

void ph_store_anchors_free(phnsw_store *s) {
  if (s->anchors) hipFree(s->anchors);
  if (s->anchor_rank) hipFree(s->anchor_rank);
  s->anchors = nullptr;
  s->anchor_rank = nullptr;
}

// whether a layer carries cells at all: not the small ones, not PQ stores, not the L2 metric (the GEMM scores dot
// products)
bool ph_layer_wants_cells(const phnsw_store *s, uint32_t n_nodes) {
  return n_nodes >= PH_POS_MIN && s->rows && s->n >= 65536 && s->metric != PHNSW_METRIC_L2 && !getenv("PHNSW_NO_LOCALITY");
}

// out_pos[i] = chain rank of the nearest anchor of node first + i, for count nodes of the layer (the sharded build
// gives every rank a node range and all-gathers the ranks' pieces: the GEMM is the one step of layer_begin that costs)
int ph_layer_cells_range(const phnsw_store *cs, const PhLayerHost &L, uint32_t first, uint32_t count, uint32_t *out_pos) {
  phnsw_store *s = const_cast<phnsw_store *>(cs);
  if (count == 0) return 0;
  int rc = ph_store_anchors(s);
  if (rc) return rc;
  const uint32_t A = s->n_anchors, ld = s->ld;
  const uint32_t QC = 65536;  // layer vectors per GEMM pass
  float *qrows = nullptr, *scores = nullptr;
  hipError_t e = hipMalloc(&scores, (size_t)std::min(QC, count) * A * 4);
  if (e == hipSuccess && !L.identity) e = hipMalloc(&qrows, (size_t)std::min(QC, count) * ld * 4);
  if (e == hipSuccess) {
    for (uint32_t at = 0; at < count; at += QC) {
      const uint32_t cnt = std::min(QC, count - at);
      const float *Q = s->rows + (uint64_t)(first + at) * ld;  // identity layer: node i is row i
      if (!L.identity) {
        hipLaunchKernelGGL(ph_gather_rows_f32_kernel, dim3((cnt + 3) / 4), dim3(256), 0, 0, s->rows, ld, L.nodes + first + at,
                           1u, cnt, qrows);
        Q = qrows;
      }
      dim3 grid((A + BF_TN - 1) / BF_TN, (cnt + BF_TM - 1) / BF_TM);
      // the leading 256 dimensions are enough for a coarse cell (a hint, not a result)
      hipLaunchKernelGGL(ph_gemm_nt_mfma_kernel, grid, dim3(256), 0, 0, Q, ld, cnt, s->anchors, ld, A,
                         std::min<uint32_t>(ld, 256u), scores, (uint64_t)A);
      hipLaunchKernelGGL(ph_argmax_rows_kernel, dim3(std::min<uint32_t>(cnt, 256u * 16u)), dim3(64), 0, 0, scores,
                         (uint64_t)A, A, cnt, out_pos + at);
    }
    hipLaunchKernelGGL(ph_rank_of_cell_kernel, dim3((count + 255) / 256), dim3(256), 0, 0, out_pos, count, s->anchor_rank);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  if (e != hipSuccess) rc = ph_hip_fail(e, "layer cells (anchor GEMM)", __FILE__, __LINE__);
  if (qrows) hipFree(qrows);
  if (scores) hipFree(scores);
  return rc;
}

// L.pos[node] = chain rank of the node's nearest anchor, for the whole layer.  No-op when the layer has its
// cells or wants none.
int ph_layer_anchor_pos(const phnsw_store *cs, PhLayerHost &L) {
  const uint32_t n = L.n_nodes;
  if (L.pos || !ph_layer_wants_cells(cs, n)) return 0;
  hipError_t e = hipMalloc(&L.pos, (size_t)n * 4);
  if (e != hipSuccess) return ph_hip_fail(e, "layer cells", __FILE__, __LINE__);
  int rc = ph_layer_cells_range(cs, L, 0, n, L.pos);
  if (rc) {
    hipFree(L.pos);
    L.pos = nullptr;
  }
  return rc;
}

// milliseconds the MFMA GEMM launches of the calling thread's last brute-force pass took
extern "C" float phnsw_bruteforce_last_gemm_ms(void) { return g_bf_gemm_ms; }

extern "C" int phnsw_bruteforce_topk(const phnsw_store *s, const float *queries, uint64_t nq, uint32_t k,
                                     uint64_t *out_ids, float *out_d) try {
  if (!s || !queries || !out_ids || !out_d || nq == 0) {
    ph_set_error("phnsw_bruteforce_topk: invalid argument");
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(s->device));
  float *qd = nullptr, *od = nullptr;
  uint32_t *oid = nullptr;
  hipError_t e = hipMalloc(&qd, (size_t)nq * s->ld * 4);
  if (e == hipSuccess && s->ld != s->dim) e = hipMemset(qd, 0, (size_t)nq * s->ld * 4);
  if (e == hipSuccess)
    e = hipMemcpy2D(qd, (size_t)s->ld * 4, queries, (size_t)s->dim * 4, (size_t)s->dim * 4, nq, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&oid, (size_t)nq * k * 4);
  if (e == hipSuccess) e = hipMalloc(&od, (size_t)nq * k * 4);
  int rc = e == hipSuccess ? 0 : ph_hip_fail(e, "bruteforce staging", __FILE__, __LINE__);
  if (!rc) rc = phnsw_bruteforce_topk_device(s, qd, s->ld, nq, k, oid, od, nullptr);
  if (!rc) {
    std::vector<uint32_t> h((size_t)nq * k);
    e = hipMemcpy(h.data(), oid, h.size() * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_d, od, h.size() * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
      rc = ph_hip_fail(e, "bruteforce readback", __FILE__, __LINE__);
    else
      for (size_t i = 0; i < h.size(); i++) out_ids[i] = h[i] == PH_EMPTY32 ? PHNSW_EMPTY : h[i];
  }
  if (qd) hipFree(qd);
  if (oid) hipFree(oid);
  if (od) hipFree(od);
  return rc;
} catch (...) { return ph_caught(); }
