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
//   ph_tiny_table_mfma_kernel  the same table on the matrix cores for dot-product metrics over rows of whole
//                          64-chunk groups (256 / 768 / 1536 floats): chains of v_mfma_f32_32x32x1_2b_f32, one
//                          fused multiply-add per K step, visited in the butterfly's own tree order -- the same
//                          bits again (see the comment above the kernel).
//
// search.hip then walks those layers with its visited set and its table row in LDS and never
// touches a vector row there.  Compute bound, not HBM: 2*dim flop per entry at the f32 vector or matrix rate.
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

// ------------------------------------------------------------------ the same table on the matrix cores
//
// D[position][node] is a GEMM, but its bits are fixed by the per-hop evaluation: lane l of a wave chains
// fma(x, q, acc) over the 4 components of float4 chunks l, l + 64, ... (chain_partial) and the 64 partial sums
// meet in the xor-butterfly's tree (32, 16, 8, 4, 2, 1).  v_mfma_f32_32x32x1_2b_f32 performs exactly one fused
// multiply-add per output element and K step (measured bit-identical to v_fma_f32 including denormals:
// scripts/micro/mfma_fma_exact.hip), so a chain of NV*4 such instructions over the columns of lane l builds
// p_l for 32 x 32 (position, node) pairs at once -- its two blocks take lanes l and l + 32, whose sum is the
// butterfly's first step.  The 32 sums s1[l] are visited in bit-reversed order, which is a depth-first walk of
// the butterfly's tree: finished subtrees are added as soon as both halves exist, five pending 32 x 32 tiles at
// most.  Operands are packed once per launch (ph_tiny_pack_kernel) so that what a leaf needs of 32 rows is one
// contiguous, lane-ordered run: [row tile][leaf t][chunk k][block b][row x] float4.
typedef float ph_f32x32 __attribute__((ext_vector_type(32)));

__host__ __device__ constexpr uint32_t ph_rev5(uint32_t t) {
  return ((t & 1u) << 4) | ((t & 2u) << 2) | (t & 4u) | ((t & 8u) >> 2) | ((t & 16u) >> 4);
}

struct PhTinyPackArgs {
  const float *vecs;     // stored rows
  uint32_t ld;
  const float *queries;  // raw query rows, or nullptr: rows are stored vectors
  uint32_t ldq;
  const uint32_t *ids;    // row -> VectorId of a stored vector (when queries == nullptr)
  const uint32_t *order;  // nullable: row r reads entry order[r]
  uint32_t n, n_pad;      // rows, rows rounded up to the block tile (the rest is zero)
  uint32_t nv;            // float4 chunks per lane
  float4 *out;
};

// blockDim (32 rows, 8 chunks): a row's 8 chunks are one 128-byte read, a chunk's 32 rows one 512-byte write
__global__ void ph_tiny_pack_kernel(PhTinyPackArgs p) {
  const uint32_t r = blockIdx.x * 32u + threadIdx.x;
  const uint32_t c = blockIdx.y * 8u + threadIdx.y;  // < 64 * nv
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r < p.n) {
    const uint32_t e = p.order ? p.order[r] : r;
    const float4 *src = p.queries ? (const float4 *)(p.queries + (uint64_t)e * p.ldq)
                                  : (const float4 *)(p.vecs + (uint64_t)p.ids[e] * p.ld);
    v = src[c];
  }
  const uint32_t t = ph_rev5(c & 31u), b = (c >> 5) & 1u, k = c >> 6;
  p.out[(((((uint64_t)(r >> 5) * 32u + t) * p.nv + k) * 2u + b) << 5) + (r & 31u)] = v;
}

struct PhTinyMfmaArgs {
  const float4 *pq, *pn;  // packed positions / nodes
  uint32_t npos, tiny_n, stride;
  uint32_t qtiles, ntiles;  // 64-row block tiles along each side
  int metric;
  float *D;
};

// 256 threads = 4 waves, block tile 64 positions x 64 nodes, wave w: positions half w & 1, nodes half w >> 1.
// G leaves of all four 32-row tiles are staged per step (double buffered: the next step's global loads are in
// flight while the matrix cores work on this one); wave w stages row tile w (0, 1: positions; 2, 3: nodes), G * NV
// wave-wide loads of 1 KiB per step.  The butterfly's tree is written as a compile-time recursion, so that the
// pending partial tiles are plain locals (five at most) and every index is a constant.
template <int NV, int G>
struct PhMfmaTile {
  static constexpr int TILE4 = G * NV * 64;  // float4 of one row tile in one step
  static constexpr int BUF4 = 4 * TILE4;
  static constexpr int NG = 32 / G;
  float4 *sm4;
  const float4 *mysrc;   // this wave's row tile in the packed operand, + lane
  const float4 *A0, *B0;  // this wave's operand tiles in LDS buffer 0, + lane
  float4 *mydst;          // where this wave parks its row tile in LDS buffer 0, + lane
  float4 stage[G * NV];

  __device__ __forceinline__ void fetch(int g) {
#pragma unroll
    for (int i = 0; i < G * NV; i++) stage[i] = mysrc[g * TILE4 + i * 64];
  }
  __device__ __forceinline__ void park(int buf) {
#pragma unroll
    for (int i = 0; i < G * NV; i++) mydst[buf * BUF4 + i * 64] = stage[i];
  }
  // s1 of leaf T: p_l + p_(l + 32) for l = ph_rev5(T), 32 x 32 pairs (16 registers per lane)
  template <int T>
  __device__ __forceinline__ void leaf(float (&out)[16]) {
    constexpr int g = T / G, tt = T % G;
    if (tt == 0 && g + 1 < NG) fetch(g + 1);
    ph_f32x32 acc;
#pragma unroll
    for (int v = 0; v < 32; v++) acc[v] = 0.f;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const float4 qa = A0[(g & 1) * BUF4 + (tt * NV + k) * 64];
      const float4 xb = B0[(g & 1) * BUF4 + (tt * NV + k) * 64];
      acc = __builtin_amdgcn_mfma_f32_32x32x1f32(qa.x, xb.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x1f32(qa.y, xb.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x1f32(qa.z, xb.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x1f32(qa.w, xb.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 16; v++) {
      out[v] = acc[v] + acc[v + 16];
      // pins the add here: instruction selection otherwise keeps all 32 leaves' accumulators (spilled) and adds at the end
      asm volatile("" : "+v"(out[v]));
    }
    if (tt == G - 1) {
      if (g + 1 < NG) park((g + 1) & 1);
      __syncthreads();
    }
  }
  // the subtree of 2^L leaves starting at leaf T0 (bit-reversed leaf order = the butterfly's pairing, deepest first)
  template <int L, int T0>
  __device__ __forceinline__ void subtree(float (&out)[16]) {
    if constexpr (L == 0) {
      leaf<T0>(out);
    } else {
      float lo[16], hi[16];
      subtree<L - 1, T0>(lo);
      subtree<L - 1, T0 + (1 << (L - 1))>(hi);
#pragma unroll
      for (int v = 0; v < 16; v++) {
        out[v] = lo[v] + hi[v];
        asm volatile("" : "+v"(out[v]));
      }
    }
  }
};

template <int NV, int G>
__global__ __launch_bounds__(256) void ph_tiny_table_mfma_kernel(PhTinyMfmaArgs a) {
  extern __shared__ float4 sm4[];
  using Tile = PhMfmaTile<NV, G>;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  // XCD-aware tile order: blocks are dealt round-robin to the 8 XCDs, so the blocks one L2 sees are B % 8 == xcd;
  // each XCD walks its own 8 x 8 super-tiles (8 position tiles x 8 node tiles = 3 MiB of operands, inside its L2)
  const uint32_t B = blockIdx.x, xcd = B & 7u, slot = B >> 3;
  const uint32_t sq = (a.qtiles + 7u) / 8u;  // super-tiles along positions
  const uint32_t super = (slot >> 6) * 8u + xcd, within = slot & 63u;
  const uint32_t qt = (super % sq) * 8u + (within & 7u), nt = (super / sq) * 8u + (within >> 3);
  if (qt >= a.qtiles || nt >= a.ntiles) return;  // the whole block
  Tile t;
  t.sm4 = sm4;
  t.mysrc = (w < 2u ? a.pq + (uint64_t)(2u * qt + w) * (32u * NV * 64u)
                    : a.pn + (uint64_t)(2u * nt + (w - 2u)) * (32u * NV * 64u)) + lane;
  t.mydst = sm4 + w * Tile::TILE4 + lane;
  t.A0 = sm4 + (w & 1u) * Tile::TILE4 + lane;
  t.B0 = sm4 + (2u + (w >> 1)) * Tile::TILE4 + lane;
  t.fetch(0);
  t.park(0);
  __syncthreads();
  float V[16];
  t.template subtree<5, 0>(V);
  // 32x32 accumulator layout: register v of lane L holds row 8 * (v / 4) + 4 * (L / 32) + v % 4, column L % 32;
  // rows are the A operand's (positions), columns the B operand's (nodes): a register is two 128-byte row pieces
  const uint32_t r = (2u * nt + (w >> 1)) * 32u + (lane & 31u);
#pragma unroll
  for (int v = 0; v < 16; v++) {
    const uint32_t p = (2u * qt + (w & 1u)) * 32u + 8u * (v / 4) + 4u * (lane >> 5) + (v & 3);
    if (p < a.npos && r < a.tiny_n) a.D[(uint64_t)p * a.stride + r] = finalize_metric(V[v], a.metric);
  }
}

// ------------------------------------------------------------------ host side

void ph_tiny_free(PhWorkspace &ws) {
  if (ws.tiny_d) hipFree(ws.tiny_d);
  if (ws.tiny_nbr) hipFree(ws.tiny_nbr);
  if (ws.tiny_member) hipFree(ws.tiny_member);
  if (ws.tiny_pq) hipFree(ws.tiny_pq);
  if (ws.tiny_pn) hipFree(ws.tiny_pn);
  ws.tiny_pq = ws.tiny_pn = nullptr;
  ws.tiny_pq_bytes = ws.tiny_pn_bytes = 0;
  ws.tiny_d = nullptr;
  ws.tiny_nbr = ws.tiny_member = nullptr;
  ws.tiny_d_bytes = ws.tiny_nbr_bytes = ws.tiny_member_bytes = 0;
}

// rows the matrix-core table kernel takes: a dot-product metric over whole 64-chunk rows (256 / 768 / 1536 floats)
static bool tiny_mfma_shape(int metric, uint32_t ld) {
  return metric != PHNSW_METRIC_L2 && (ld == 256u || ld == 768u || ld == 1536u) && !getenv("PHNSW_TINY_VALU");
}

bool ph_tiny_matrix_cores(const phnsw_index *ix) { return tiny_mfma_shape(ix->store->metric, ix->store->ld); }

// Which leading layers run densely.  Measured at 1M x 768 (100 000 queries): the vector-unit tile pass costs
// 0.030 us per (query, node), the matrix-core one 0.016 us, a table lookup on the walk 0.08 us, a gathered
// evaluation 0.28-0.35 us; closest_nodes evaluates about 7 x number_of_candidates nodes of a layer it cannot
// exhaust.  A layer of n nodes is therefore worth a table when n * cost < 7 * ef * 0.2, i.e. n <= 48 * ef (80 * ef
// on the matrix cores) and <= PH_TINY_MAX_NODES: at ef 104 the 7 000-node layer of a 1M x 768 cosine index is
// tabulated on the matrix cores and stays on the per-hop path otherwise; at ef >= 150 (and in every build round,
// ef 300) it is tabulated either way.  PHNSW_TINY_MAX overrides.
uint32_t ph_tiny_layer_count(const phnsw_index *ix, uint32_t n_layers, uint32_t ef) {
  const bool off = getenv("PHNSW_NO_TINY") != nullptr;  // tests compare both paths
  if (off || !ix->store->rows || ix->store->ld / 4 > 384) return 0;
  const uint64_t per_ef = tiny_mfma_shape(ix->store->metric, ix->store->ld) ? 80ull : 48ull;
  uint64_t cap = std::min<uint64_t>(PH_TINY_MAX_NODES, per_ef * ef);
  if (const char *e = getenv("PHNSW_TINY_MAX"))
    if (atoi(e) > 0) cap = std::min<uint64_t>(PH_TINY_MAX_NODES, (uint64_t)atoi(e));
  uint32_t T = 0;
  while (T < n_layers && T < PH_TINY_MAX_LAYERS && ix->layers[T].n_nodes <= cap) T++;
  return T;
}

static uint32_t tiny_stride_of(uint32_t n) { return (n + 63u) / 64u * 64u; }

// the table of one launch is kept below 4 GiB (PHNSW_TINY_TABLE_BYTES overrides: tests, small devices); longer
// query lists run in chunks (api.hip)
uint64_t ph_tiny_max_positions(const phnsw_index *ix, uint32_t n_layers, uint32_t ef) {
  uint32_t T = ph_tiny_layer_count(ix, n_layers, ef);
  if (!T) return 0;
  uint64_t budget = 4ull << 30;
  if (const char *e = getenv("PHNSW_TINY_TABLE_BYTES"))
    if (atoll(e) > 0) budget = (uint64_t)atoll(e);
  return std::max<uint64_t>(64, budget / ((uint64_t)tiny_stride_of(ix->layers[T - 1].n_nodes) * 4u));
}

size_t ph_tiny_lds_bytes(const PhSearchArgs &a) {
  if (!a.tiny_layers) return 0;
  return (a.tiny_n <= a.tiny_lds_nodes ? (size_t)a.tiny_stride * 4u : 0u) + (size_t)((a.tiny_n + 31u) / 32u + 1u) * 4u;
}

template <class T>
static hipError_t grow(T **p, size_t *have, size_t need) {
  if (*have >= need) return hipSuccess;
  if (*p) ph_timed_free(*p);
  *p = nullptr;
  *have = 0;
  hipError_t e = ph_timed_malloc((void **)p, need);
  if (e == hipSuccess) *have = need;
  return e;
}

// the neighbour rows of the dense layers rewritten in table ids + the membership mask (per launch: the rows change
// between build rounds)
static int tiny_prep_graph(PhWorkspace &ws, PhSearchArgs &a, uint32_t T, uint32_t tn, hipStream_t stream) {
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
  if (grow(&ws.tiny_nbr, &ws.tiny_nbr_bytes, nbr_words * 4u) != hipSuccess ||
      grow(&ws.tiny_member, &ws.tiny_member_bytes, (size_t)(PH_TINY_MAX_NODES + 1u) * 4u) != hipSuccess) {
    (void)hipGetLastError();
    return 1;  // not an error: the launch walks every layer on the per-hop path
  }
  p.nbr = ws.tiny_nbr;
  p.member = ws.tiny_member;
  PH_HIP(hipMemsetAsync(ws.tiny_nbr, 0xFF, nbr_words * 4u, stream));
  PH_HIP(hipMemsetAsync(ws.tiny_member, 0, (size_t)(tn + 1u) * 4u, stream));
  hipLaunchKernelGGL(ph_tiny_prep_kernel, dim3((tn + 255u) / 256u, T), dim3(256), 0, stream, p);
  PH_HIP(hipGetLastError());
  return 0;
}

// D[p][t] = compare_vec(query of position p, Stored(tnodes[t])) for npos positions (position p = query order[p], or p
// itself; raw queries or Stored ids), the per-hop path's bits: the matrix-core kernel where it applies, else the
// vector-unit tile pass.  returns 1 (no error set) when its operand buffers cannot be allocated.
static int tiny_table(PhWorkspace &ws, const PhDistArgs &dist, const float *queries, uint32_t ldq, const uint32_t *qids,
                      const uint32_t *order, uint32_t npos, const uint32_t *tnodes, uint32_t tn, uint32_t stride, float *D,
                      hipStream_t stream) {
  PhTinyTableArgs t;
  memset(&t, 0, sizeof(t));
  t.dist = dist;
  t.queries = queries;
  t.ldq = ldq;
  t.qids = qids;
  t.order = order;
  t.npos = npos;
  t.tnodes = tnodes;
  t.tiny_n = tn;
  t.stride = stride;
  t.D = D;
  if (const char *e = getenv("PHNSW_TINY_DBG")) t.dbg = (uint32_t)atoi(e);
  const uint32_t nv4 = dist.nv4;
  const int nv = nv4 <= 64 ? 1 : (nv4 <= 192 ? 3 : 6);
  // dot-product metrics over whole 64-chunk rows go to the matrix cores (same bits, see above); the Euclidean
  // chain (fma(d, d, acc) of a difference) is not a product of the two operands and stays on the vector units, as
  // do ragged rows and launches of a handful of queries.  PHNSW_TINY_VALU=1 forces the vector kernel (tests).
  const bool mfma = tiny_mfma_shape(dist.metric, nv4 * 4u) && nv4 == 64u * (uint32_t)nv && npos >= 32u;
  if (mfma) {
    const uint32_t qtiles = (npos + 63u) / 64u, ntiles = (tn + 63u) / 64u;
    const size_t row_bytes = (size_t)nv * 64u * sizeof(float4);
    if (grow(&ws.tiny_pq, &ws.tiny_pq_bytes, (size_t)qtiles * 64u * row_bytes) != hipSuccess ||
        grow(&ws.tiny_pn, &ws.tiny_pn_bytes, (size_t)ntiles * 64u * row_bytes) != hipSuccess) {
      (void)hipGetLastError();
      return 1;
    }
    PhTinyPackArgs k;
    memset(&k, 0, sizeof(k));
    k.vecs = dist.vecs;
    k.ld = dist.ld;
    k.nv = (uint32_t)nv;
    k.queries = queries;
    k.ldq = ldq;
    k.ids = qids;
    k.order = order;
    k.n = npos;
    k.n_pad = qtiles * 64u;
    k.out = ws.tiny_pq;
    hipLaunchKernelGGL(ph_tiny_pack_kernel, dim3(k.n_pad / 32u, 8u * (uint32_t)nv), dim3(32, 8), 0, stream, k);
    k.queries = nullptr;
    k.ids = tnodes;
    k.order = nullptr;
    k.n = tn;
    k.n_pad = ntiles * 64u;
    k.out = ws.tiny_pn;
    hipLaunchKernelGGL(ph_tiny_pack_kernel, dim3(k.n_pad / 32u, 8u * (uint32_t)nv), dim3(32, 8), 0, stream, k);
    PH_HIP(hipGetLastError());
    PhTinyMfmaArgs m;
    memset(&m, 0, sizeof(m));
    m.pq = ws.tiny_pq;
    m.pn = ws.tiny_pn;
    m.npos = npos;
    m.tiny_n = tn;
    m.stride = stride;
    m.qtiles = qtiles;
    m.ntiles = ntiles;
    m.metric = dist.metric;
    m.D = D;
    const uint32_t supers = ((qtiles + 7u) / 8u) * ((ntiles + 7u) / 8u);
    const uint32_t blocks = (supers + 7u) / 8u * 8u * 64u;
    constexpr int G = 2;
    const size_t lds = (size_t)2 * 4 * G * nv * 64 * sizeof(float4);
    if (nv == 1) {
      hipLaunchKernelGGL((ph_tiny_table_mfma_kernel<1, G>), dim3(blocks), dim3(256), lds, stream, m);
    } else if (nv == 3) {
      hipLaunchKernelGGL((ph_tiny_table_mfma_kernel<3, G>), dim3(blocks), dim3(256), lds, stream, m);
    } else {
      PH_HIP(hipFuncSetAttribute((const void *)ph_tiny_table_mfma_kernel<6, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((ph_tiny_table_mfma_kernel<6, G>), dim3(blocks), dim3(256), lds, stream, m);
    }
  } else {
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
  }
  PH_HIP(hipGetLastError());
  return 0;
}

int ph_tiny_prepare(const phnsw_index *ix, PhWorkspace &ws, PhSearchArgs &a, uint32_t max_layers, hipStream_t stream) {
  a.tiny_layers = 0;
  a.tiny_rows = 0;
  if (a.knn_mode || a.layer_lo) return 0;
  uint32_t T = std::min(ph_tiny_layer_count(ix, a.n_layers, a.ef), max_layers);
  if (!T) return 0;
  const uint32_t tn = ix->layers[T - 1].n_nodes, stride = tiny_stride_of(tn), npos = a.nq;
  if ((uint64_t)npos > ph_tiny_max_positions(ix, a.n_layers, a.ef)) return 0;  // the caller chunks; never reached through api.hip
  // the table is an accelerator, never a requirement: when the device cannot spare it the launch simply walks
  // every layer on the per-hop path (same results)
  if (grow(&ws.tiny_d, &ws.tiny_d_bytes, std::max<size_t>((size_t)npos * stride * 4u, 1u << 20)) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int rc = tiny_prep_graph(ws, a, T, tn, stream);
  if (rc) return rc < 0 ? rc : 0;
  rc = tiny_table(ws, a.dist, a.queries, a.ldq, a.qids, a.order, npos, a.layers[T - 1].nodes, tn, stride, ws.tiny_d, stream);
  if (rc) return rc < 0 ? rc : 0;
  a.tiny_layers = T;
  a.tiny_n = tn;
  a.tiny_lds_nodes = PH_TINY_LDS_NODES;
  a.tiny_stride = stride;
  a.tiny_d = ws.tiny_d;
  a.tiny_nbr = ws.tiny_nbr;
  a.tiny_member = ws.tiny_member;
  return 0;
}

// ------------------------------------------------------------------ the build's kept table

void ph_build_table_free(phnsw_index *ix) {
  for (PhBuildTable &b : ix->bt)
    if (b.D) ph_timed_free(b.D);
  ix->bt.clear();
  for (auto &sp : ix->bt_spare) ph_timed_free(sp.first);
  ix->bt_spare.clear();
}
// the rows go, the slot (which layer it belongs to) stays; the buffer waits for the next table that fits
static void build_table_drop(phnsw_index *ix, PhBuildTable &b) {
  if (b.D) ix->bt_spare.emplace_back(b.D, b.cap);
  b.D = nullptr;
  b.bytes = b.cap = 0;
  b.lo = b.hi = b.lo_alloc = b.hi_alloc = 0;
}
// every table is void (a node list changed): the slots go, the buffers are kept for their successors
static void build_table_recycle(phnsw_index *ix) {
  for (PhBuildTable &b : ix->bt) build_table_drop(ix, b);
  ix->bt.clear();
}
// a buffer of at least `bytes`: the smallest spare that holds them without wasting more than half of itself, or a new
// allocation with a sixteenth of headroom (a promotion grows a layer by a few per cent)
static float *build_table_buffer(phnsw_index *ix, size_t bytes, size_t *cap) {
  int best = -1;
  for (size_t i = 0; i < ix->bt_spare.size(); i++)
    if (ix->bt_spare[i].second >= bytes && ix->bt_spare[i].second <= 2 * bytes &&
        (best < 0 || ix->bt_spare[i].second < ix->bt_spare[(size_t)best].second))
      best = (int)i;
  if (best >= 0) {
    float *p = ix->bt_spare[(size_t)best].first;
    *cap = ix->bt_spare[(size_t)best].second;
    ix->bt_spare.erase(ix->bt_spare.begin() + best);
    return p;
  }
  size_t mfree = 0, mtotal = 0;
  if (hipMemGetInfo(&mfree, &mtotal) != hipSuccess) return nullptr;
  size_t want = bytes + bytes / 16;
  if (want + (8ull << 30) > mfree) {  // the searches' own workspaces come first: give the spares back, then try the bare size
    for (auto &sp : ix->bt_spare) ph_timed_free(sp.first);
    ix->bt_spare.clear();
    if (hipMemGetInfo(&mfree, &mtotal) != hipSuccess) return nullptr;
    want = bytes;
  }
  if (want + (8ull << 30) > mfree || bytes > mtotal / 2) return nullptr;
  float *p = nullptr;
  if (ph_timed_malloc((void **)&p, want) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  *cap = want;
  return p;
}

// rows kept only for table layers of at least this many nodes (smaller tables cost less than the bookkeeping);
// PHNSW_BUILD_TABLE_MIN overrides (the tests keep every table), PHNSW_NO_BUILD_TABLE=1 switches the cache off
static uint32_t build_table_min() {
  if (const char *e = getenv("PHNSW_BUILD_TABLE_MIN"))
    if (atoi(e) > 0) return (uint32_t)atoi(e);
  return 1024u;
}

int ph_build_table_prepare(phnsw_index *ix, PhWorkspace &ws, PhSearchArgs &a, const PhRowHint &h, uint32_t T,
                           hipStream_t stream, bool *used) {
  *used = false;
  a.tiny_rows = 0;
  if (!ix->bt_enabled || !T || !a.qids || a.queries || a.knn_mode || a.layer_lo || getenv("PHNSW_NO_BUILD_TABLE")) return 0;
  const uint32_t tn = ix->layers[T - 1].n_nodes, stride = tiny_stride_of(tn);
  const uint32_t *tnodes = a.layers[T - 1].nodes;
  if (tn < build_table_min() || h.qn == 0) return 0;
  // one table per querying layer (an outer round of improve_neighbors_upto links every layer in turn); a changed node
  // list anywhere (epoch) voids them all
  if (!ix->bt.empty() && ix->bt[0].epoch != ix->nodes_epoch) build_table_recycle(ix);
  PhBuildTable *found = nullptr;
  for (PhBuildTable &b : ix->bt)
    if (b.qnodes == h.qnodes && b.qn == h.qn) found = &b;
  if (!found) {
    if (ix->bt.size() >= PH_MAX_LAYERS) return 0;
    ix->bt.emplace_back();
    found = &ix->bt.back();
    found->epoch = ix->nodes_epoch;
    found->qnodes = h.qnodes;
    found->qn = h.qn;
  }
  PhBuildTable &B = *found;
  const uint32_t need_lo = h.contiguous ? h.first : 0u, need_hi = h.contiguous ? h.first + h.count : h.qn;
  if (need_hi > h.qn || need_lo >= need_hi) return 0;
  const bool same = B.D && B.tnodes == tnodes && B.tn == tn && B.T == T;
  if (B.D && !same) build_table_drop(ix, B);
  if (!h.contiguous && !(B.D && B.lo <= need_lo && B.hi >= need_hi)) return 0;  // a sample: only from rows already there
  if (!B.D || need_lo < B.lo_alloc || need_hi > B.hi_alloc) {
    // (re)allocate for the requested range: the whole layer on one GPU, a rank's node range in a sharded build
    if (B.D && (need_lo > B.lo_alloc || need_hi < B.hi_alloc)) return 0;  // a second, different range: not worth juggling
    build_table_drop(ix, B);
    const size_t bytes = (size_t)(need_hi - need_lo) * stride * 4u;
    B.D = build_table_buffer(ix, bytes, &B.cap);
    if (!B.D) return 0;
    B.bytes = bytes;
    B.qnodes = h.qnodes;
    B.qn = h.qn;
    B.tnodes = tnodes;
    B.tn = tn;
    B.T = T;
    B.stride = stride;
    B.lo_alloc = need_lo;
    B.hi_alloc = need_hi;
    B.lo = B.hi = need_lo;
    B.epoch = ix->nodes_epoch;
  }
  // rows missing from [need_lo, need_hi): the valid rows are one interval; fill what lies outside it
  auto fill = [&](uint32_t lo, uint32_t hi) -> int {
    const uint32_t PIECE = 131072u;  // bounds the packed-operand buffer (400 MB at 768 floats)
    for (uint32_t at = lo; at < hi; at += PIECE) {
      const uint32_t cnt = std::min(PIECE, hi - at);
      int rc = tiny_table(ws, a.dist, nullptr, 0, h.qnodes + at, nullptr, cnt, tnodes, tn, stride,
                          B.D + (size_t)(at - B.lo_alloc) * stride, stream);
      if (rc) return rc;
    }
    return 0;
  };
  int rc = 0;
  if (B.lo == B.hi) {
    rc = fill(need_lo, need_hi);
    if (!rc) B.lo = need_lo, B.hi = need_hi;
  } else {
    if (need_lo < B.lo) {
      rc = fill(need_lo, B.lo);
      if (!rc) B.lo = need_lo;
    }
    if (!rc && need_hi > B.hi) {
      rc = fill(B.hi, need_hi);
      if (!rc) B.hi = need_hi;
    }
  }
  if (rc) return rc < 0 ? rc : 0;
  rc = tiny_prep_graph(ws, a, T, tn, stream);
  if (rc) return rc < 0 ? rc : 0;
  a.tiny_layers = T;
  a.tiny_n = tn;
  a.tiny_lds_nodes = PH_TINY_LDS_NODES;
  a.tiny_stride = stride;
  a.tiny_d = B.D;
  a.tiny_rows = 1;
  a.tiny_row_first = B.lo_alloc;
  a.tiny_row_map = h.vec2node;
  a.tiny_nbr = ws.tiny_nbr;
  a.tiny_member = ws.tiny_member;
  *used = true;
  return 0;
}
