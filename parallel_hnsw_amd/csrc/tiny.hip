// Dense top layers: the LDS-staged queries x rows tile pass of the search path.
//
// Every query of Hnsw::search walks ALL layers (/root/reference/src/search.rs:113-137), and in
// the small ones at the top -- 4 / 48 / 578 nodes of a 1M-vector index with the default order 12
// (lib.rs:1883-1899) -- a queue of number_of_candidates >= 100 makes closest_nodes
// (lib.rs:175-248) evaluate nearly every node.  Reading those few hundred rows once per query
// is the one place on the traversal where a batch of queries meets the SAME candidate rows, so
// it is done as a tile pass instead of a gather:
//
//   ph_tiny_table_kernel   D[position][node] = compare_vec(query, Stored(node's vector)) for every
//                          query of the launch and every node of the largest small layer.  A
//                          256-thread block stages 8 rows in LDS (ds_write_b128) and its 4 waves
//                          hold 8 queries each in registers: one LDS read of a row feeds 8 fma
//                          chains.  The chain is chain_partial (phnsw_device.h) and the 64 partial
//                          sums are combined in the butterfly's order, so D holds exactly the bits
//                          the per-hop evaluation would produce.  64 results (8 queries x 8 rows)
//                          are reduced together: two registers are folded into one per step
//                          (lanes pick their half), 63 shuffles instead of 384.
//   ph_tiny_prep_kernel    neighbour rows of the small layers rewritten in the id space of the
//                          largest one ("table ids": layers are nested and their node lists are
//                          sorted, lib.rs:685 / search.rs:150-157, so NodeId order == table-id
//                          order and queue ties break identically), plus a membership mask.
//
// search.hip then walks those layers with its visited set and its table row in LDS and never
// touches a vector row there.  VALU bound, not HBM: 2*dim flop per entry at the f32 vector rate.
// MFMA would change the summation order (k-ordered chain), i.e. the result bits: not used.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "phnsw_device.h"

struct PhTinyPrepArgs {
  PhLayerDev layers[PH_TINY_MAX_LAYERS];
  uint32_t T, tiny_n;
  uint32_t off[PH_TINY_MAX_LAYERS];
  uint32_t *nbr;
  uint32_t *member;  // [tiny_n + 1]
};

__global__ void ph_tiny_prep_kernel(PhTinyPrepArgs p) {
  const uint32_t l = blockIdx.y;
  const PhLayerDev L = p.layers[l], TL = p.layers[p.T - 1];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < L.n_nodes; i += gridDim.x * blockDim.x) {
    const uint32_t vid = L.nodes[i];
    const uint32_t t = TL.vec2node ? TL.vec2node[vid] : vid;
    if (t >= p.tiny_n) {  // a node of layer l that the table layer lacks: not nested, table unusable
      p.member[p.tiny_n] = 1u;
      continue;
    }
    atomicOr(&p.member[t], 1u << l);
    for (uint32_t k = 0; k < L.W; k++) {
      const uint32_t nb = L.neighbors[(uint64_t)i * L.W + k];
      uint32_t tn = PH_EMPTY32;
      if (nb < L.n_nodes) {
        const uint32_t v2 = L.nodes[nb];
        tn = TL.vec2node ? TL.vec2node[v2] : v2;
        if (tn >= p.tiny_n) {
          p.member[p.tiny_n] = 1u;
          tn = PH_EMPTY32;
        }
      }
      p.nbr[p.off[l] + (uint64_t)t * L.W + k] = tn;
    }
  }
}

struct PhTinyTableArgs {
  PhDistArgs dist;
  const float *queries;
  uint32_t ldq;
  const uint32_t *qids;
  const uint32_t *order;
  uint32_t npos;
  const uint32_t *tnodes;  // table id -> VectorId
  uint32_t tiny_n, stride;
  float *D;
  uint32_t rows_per_slice;  // multiple of 8
  uint32_t dbg;             // PHNSW_TINY_DBG (experiments): 1 = no table stores, 2 = stage only the first tile
};

// two registers folded into one: lanes whose bit `m` is clear keep A's butterfly step, the others
// B's -- the same two operands as v += shfl_xor(v, m) on each (f32 add is commutative, so which
// of the two comes first does not change the bits).  m = 32 / 16 are one v_permlane{32,16}_swap
// (gfx950) + one add; 8, 2, 1 one DPP move (row_ror:8, quad_perm); 4 two DPP moves with bank masks.
template <int M>
__device__ __forceinline__ float fold2(float A, float B, uint32_t lane) {
#ifdef PH_FOLD_GENERIC
  const bool hi = (lane & M) != 0;
  const float own = hi ? B : A, other = hi ? A : B;
  return own + __shfl_xor(other, M);
#else
  if constexpr (M == 32) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(A), __float_as_uint(B), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else if constexpr (M == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(A), __float_as_uint(B), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else {
    const bool hi = (lane & M) != 0;
    const float own = hi ? B : A, other = hi ? A : B;
    const int o = (int)__float_as_uint(other);
    int t;
    if constexpr (M == 8) {
      t = __builtin_amdgcn_update_dpp(0, o, 0x128, 0xF, 0xF, true);  // row_ror:8
    } else if constexpr (M == 4) {
      t = __builtin_amdgcn_update_dpp(o, o, 0x114, 0xF, 0xA, false);  // row_shr:4 into lanes 4-7, 12-15
      t = __builtin_amdgcn_update_dpp(t, o, 0x104, 0xF, 0x5, false);  // row_shl:4 into lanes 0-3, 8-11
    } else if constexpr (M == 2) {
      t = __builtin_amdgcn_update_dpp(0, o, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    } else {
      t = __builtin_amdgcn_update_dpp(0, o, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    }
    return own + __uint_as_float((uint32_t)t);
  }
#endif
}

template <int NV, int QT, bool EXACT, bool L2>
__device__ __forceinline__ void tiny_tile(const float4 *lds, const ph_f2 (&qv)[QT / 2][NV][4], uint32_t nv4, uint32_t lane,
                                          float &V) {
  float R[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    float4 x[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) x[k] = lds[(i * NV + k) * 64 + lane];
    float p[QT];
    ph_f2 pp[QT / 2];
    chain_partial2<NV, QT / 2, EXACT, L2>(x, qv, nv4, lane, pp);
#pragma unroll
    for (int j = 0; j < QT / 2; j++) {
      p[2 * j] = pp[j].x;
      p[2 * j + 1] = pp[j].y;
    }
    if constexpr (QT == 8) {
      float f0 = fold2<32>(p[0], p[1], lane), f1 = fold2<32>(p[2], p[3], lane);
      float f2 = fold2<32>(p[4], p[5], lane), f3 = fold2<32>(p[6], p[7], lane);
      float g0 = fold2<16>(f0, f1, lane), g1 = fold2<16>(f2, f3, lane);
      R[i] = fold2<8>(g0, g1, lane);
    } else {
      float f0 = fold2<32>(p[0], p[1], lane), f1 = fold2<32>(p[2], p[3], lane);
      R[i] = fold2<16>(f0, f1, lane);
    }
  }
  constexpr int M0 = QT == 8 ? 4 : 8;
  float h0 = fold2<M0>(R[0], R[1], lane), h1 = fold2<M0>(R[2], R[3], lane);
  float h2 = fold2<M0>(R[4], R[5], lane), h3 = fold2<M0>(R[6], R[7], lane);
  float e0 = fold2<M0 / 2>(h0, h1, lane), e1 = fold2<M0 / 2>(h2, h3, lane);
  V = fold2<M0 / 4>(e0, e1, lane);
  if constexpr (QT == 4) V += __shfl_xor(V, 1);
}

template <int NV, int QT>
__global__ __launch_bounds__(256) void ph_tiny_table_kernel(PhTinyTableArgs a) {
  __shared__ float4 rows_lds[8 * NV * 64];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const uint32_t nv4 = a.dist.nv4;
  const uint32_t p0 = (blockIdx.x * 4u + w) * QT;
  // queries in pairs: qv[j][k][e] = component e of chunk k of (query 2j, query 2j + 1)
  ph_f2 qv[QT / 2][NV][4];
#pragma unroll
  for (int j = 0; j < QT; j++) {
    const uint32_t p = p0 + j;
    const bool valid = p < a.npos;
    const uint32_t q = valid ? (a.order ? a.order[p] : p) : 0u;
    const float4 *src = nullptr;
    if (valid)
      src = a.queries ? (const float4 *)(a.queries + (uint64_t)q * a.ldq)
                      : (const float4 *)(a.dist.vecs + (uint64_t)a.qids[q] * a.dist.ld);
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const uint32_t c = lane + 64u * k;
      const float4 v = (valid && c < nv4) ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      qv[j / 2][k][0][j & 1] = v.x;
      qv[j / 2][k][1][j & 1] = v.y;
      qv[j / 2][k][2][j & 1] = v.z;
      qv[j / 2][k][3][j & 1] = v.w;
    }
  }
  const bool exact = nv4 == 64u * NV, l2 = a.dist.metric == PHNSW_METRIC_L2;
  const uint32_t r_begin = blockIdx.y * a.rows_per_slice;
  const uint32_t r_end = min(a.tiny_n, r_begin + a.rows_per_slice);
  for (uint32_t r0 = r_begin; r0 < r_end; r0 += 8) {
    __syncthreads();  // the previous tile has been consumed
    if (!(a.dbg & 2u) || r0 == r_begin)
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const uint32_t slot = 2u * w + u;
      const uint32_t rr = min(r0 + slot, a.tiny_n - 1u);
      const float4 *src = (const float4 *)(a.dist.vecs + (uint64_t)a.tnodes[rr] * a.dist.ld);
#pragma unroll
      for (int k = 0; k < NV; k++) {
        uint32_t c = lane + 64u * k;
        c = c < nv4 ? c : nv4 - 1u;
        rows_lds[(slot * NV + k) * 64 + lane] = src[c];
      }
    }
    __syncthreads();
    float V;
    if (exact) {
      if (l2)
        tiny_tile<NV, QT, true, true>(rows_lds, qv, nv4, lane, V);
      else
        tiny_tile<NV, QT, true, false>(rows_lds, qv, nv4, lane, V);
    } else {
      if (l2)
        tiny_tile<NV, QT, false, true>(rows_lds, qv, nv4, lane, V);
      else
        tiny_tile<NV, QT, false, false>(rows_lds, qv, nv4, lane, V);
    }
    // lane -> (query j, row i) after the folds: the first log2(QT) levels chose the query, the next three the row
    uint32_t j, i;
    bool writer = true;
    if (QT == 8) {
      j = ((lane >> 5) & 1u) | (((lane >> 4) & 1u) << 1) | (((lane >> 3) & 1u) << 2);
      i = ((lane >> 2) & 1u) | (((lane >> 1) & 1u) << 1) | ((lane & 1u) << 2);
    } else {
      j = ((lane >> 5) & 1u) | (((lane >> 4) & 1u) << 1);
      i = ((lane >> 3) & 1u) | (((lane >> 2) & 1u) << 1) | (((lane >> 1) & 1u) << 2);
      writer = (lane & 1u) == 0u;
    }
    const uint32_t p = p0 + j, r = r0 + i;
    if (writer && p < a.npos && r < r_end && !(a.dbg & 1u)) a.D[(uint64_t)p * a.stride + r] = finalize_metric(V, a.dist.metric);
  }
}

// ------------------------------------------------------------------ host side

void ph_tiny_free(PhWorkspace &ws) {
  if (ws.tiny_d) hipFree(ws.tiny_d);
  if (ws.tiny_nbr) hipFree(ws.tiny_nbr);
  if (ws.tiny_member) hipFree(ws.tiny_member);
  ws.tiny_d = nullptr;
  ws.tiny_nbr = ws.tiny_member = nullptr;
  ws.tiny_d_bytes = ws.tiny_nbr_bytes = ws.tiny_member_bytes = 0;
}

// Which leading layers run densely.  Measured at 1M x 768 (100 000 queries): the tile pass costs 0.030 us per
// (query, node), a table lookup on the walk 0.08 us, a gathered evaluation 0.28-0.35 us; closest_nodes evaluates
// about 7 x number_of_candidates nodes of a layer it cannot exhaust.  A layer of n nodes is therefore worth a
// table when n * 0.030 < 7 * ef * 0.2, i.e. n <= 48 * ef (and <= PH_TINY_MAX_NODES): at ef 104 the 7 000-node layer
// of a 1M index stays on the per-hop path, at ef >= 150 (and in every build round, ef 300) it is tabulated.
// PHNSW_TINY_MAX overrides.
uint32_t ph_tiny_layer_count(const phnsw_index *ix, uint32_t n_layers, uint32_t ef) {
  const bool off = getenv("PHNSW_NO_TINY") != nullptr;  // tests compare both paths
  if (off || !ix->store->rows || ix->store->ld / 4 > 384) return 0;
  uint64_t cap = std::min<uint64_t>(PH_TINY_MAX_NODES, 48ull * ef);
  if (const char *e = getenv("PHNSW_TINY_MAX"))
    if (atoi(e) > 0) cap = std::min<uint64_t>(PH_TINY_MAX_NODES, (uint64_t)atoi(e));
  uint32_t T = 0;
  while (T < n_layers && T < PH_TINY_MAX_LAYERS && ix->layers[T].n_nodes <= cap) T++;
  return T;
}

static uint32_t tiny_stride_of(uint32_t n) { return (n + 63u) / 64u * 64u; }

// the table of one launch is kept below 4 GiB; longer query lists run in chunks (api.hip)
uint64_t ph_tiny_max_positions(const phnsw_index *ix, uint32_t n_layers, uint32_t ef) {
  uint32_t T = ph_tiny_layer_count(ix, n_layers, ef);
  if (!T) return 0;
  return (4ull << 30) / ((uint64_t)tiny_stride_of(ix->layers[T - 1].n_nodes) * 4u);
}

size_t ph_tiny_lds_bytes(const PhSearchArgs &a) {
  if (!a.tiny_layers) return 0;
  return (a.tiny_n <= PH_TINY_LDS_NODES ? (size_t)a.tiny_stride * 4u : 0u) + (size_t)((a.tiny_n + 31u) / 32u + 1u) * 4u;
}

template <class T>
static hipError_t grow(T **p, size_t *have, size_t need) {
  if (*have >= need) return hipSuccess;
  if (*p) hipFree(*p);
  *p = nullptr;
  *have = 0;
  hipError_t e = hipMalloc((void **)p, need);
  if (e == hipSuccess) *have = need;
  return e;
}

int ph_tiny_prepare(const phnsw_index *ix, PhWorkspace &ws, PhSearchArgs &a, uint32_t max_layers, hipStream_t stream) {
  a.tiny_layers = 0;
  if (a.knn_mode || a.layer_lo) return 0;
  uint32_t T = std::min(ph_tiny_layer_count(ix, a.n_layers, a.ef), max_layers);
  if (!T) return 0;
  const uint32_t tn = ix->layers[T - 1].n_nodes, stride = tiny_stride_of(tn), npos = a.nq;
  if ((uint64_t)npos > ph_tiny_max_positions(ix, a.n_layers, a.ef)) return 0;  // the caller chunks; never reached through api.hip
  PhTinyPrepArgs p;
  memset(&p, 0, sizeof(p));
  p.T = T;
  p.tiny_n = tn;
  size_t nbr_words = 0;
  for (uint32_t l = 0; l < T; l++) {
    p.layers[l] = a.layers[l];
    p.off[l] = (uint32_t)nbr_words;
    a.tiny_off[l] = (uint32_t)nbr_words;
    nbr_words += (size_t)tn * a.layers[l].W;
  }
  PH_HIP(grow(&ws.tiny_d, &ws.tiny_d_bytes, std::max<size_t>((size_t)npos * stride * 4u, 1u << 20)));
  PH_HIP(grow(&ws.tiny_nbr, &ws.tiny_nbr_bytes, nbr_words * 4u));
  PH_HIP(grow(&ws.tiny_member, &ws.tiny_member_bytes, (size_t)(PH_TINY_MAX_NODES + 1u) * 4u));
  p.nbr = ws.tiny_nbr;
  p.member = ws.tiny_member;
  PH_HIP(hipMemsetAsync(ws.tiny_nbr, 0xFF, nbr_words * 4u, stream));
  PH_HIP(hipMemsetAsync(ws.tiny_member, 0, (size_t)(tn + 1u) * 4u, stream));
  hipLaunchKernelGGL(ph_tiny_prep_kernel, dim3((tn + 255u) / 256u, T), dim3(256), 0, stream, p);
  PH_HIP(hipGetLastError());

  PhTinyTableArgs t;
  memset(&t, 0, sizeof(t));
  t.dist = a.dist;
  t.queries = a.queries;
  t.ldq = a.ldq;
  t.qids = a.qids;
  t.order = a.order;
  t.npos = npos;
  t.tnodes = a.layers[T - 1].nodes;
  t.tiny_n = tn;
  t.stride = stride;
  t.D = ws.tiny_d;
  if (const char *e = getenv("PHNSW_TINY_DBG")) t.dbg = (uint32_t)atoi(e);
  const uint32_t nv4 = a.dist.nv4;
  const int nv = nv4 <= 64 ? 1 : (nv4 <= 192 ? 3 : 6);
  const uint32_t qt = nv == 6 ? 4u : 8u;
  const uint32_t gx = (npos + 4u * qt - 1u) / (4u * qt);
  const uint32_t tiles = (tn + 7u) / 8u;
  uint32_t slices = std::min<uint32_t>(tiles, std::max<uint32_t>(1u, (2048u + gx - 1u) / gx));
  t.rows_per_slice = (tiles + slices - 1u) / slices * 8u;
  slices = (tn + t.rows_per_slice - 1u) / t.rows_per_slice;
  dim3 grid(gx, slices);
  if (nv == 1)
    hipLaunchKernelGGL((ph_tiny_table_kernel<1, 8>), grid, dim3(256), 0, stream, t);
  else if (nv == 3)
    hipLaunchKernelGGL((ph_tiny_table_kernel<3, 8>), grid, dim3(256), 0, stream, t);
  else
    hipLaunchKernelGGL((ph_tiny_table_kernel<6, 4>), grid, dim3(256), 0, stream, t);
  PH_HIP(hipGetLastError());
  a.tiny_layers = T;
  a.tiny_n = tn;
  a.tiny_stride = stride;
  a.tiny_d = ws.tiny_d;
  a.tiny_nbr = ws.tiny_nbr;
  a.tiny_member = ws.tiny_member;
  return 0;
}
