// Index construction on gfx950: Hnsw::generate / generate_layer / link rounds / recall
// (/root/reference/src/lib.rs:675-893, 1070-1154, 1463-1544, 1546-1686; promotion
// lib.rs:1273-1427 is not performed -- SURVEY section 8 row f2).
//
// Every expensive step of the reference build is "run search_layers for every node of a
// layer" (K2, search.hip) or "a distance batch per node" (K3 below); what remains is
// integer bookkeeping.  The reference mutates neighbour rows under per-row RwLocks in
// thread-schedule order (lib.rs:789-815, 1102-1147); here each round is evaluated against
// a snapshot and resolved per target row as
//        row' = best-W by (distance, id) of  row U proposals          (K5)
// which is what the sequential reference code yields for (d,id)-sorted rows, and is
// deterministic.  The CPU oracle (oracle/orc_build.c) implements the same definition and
// the two are compared bit for bit.
//
//   K3 ph_seed_rows_kernel     generate_layer step 3      lib.rs:719-787 (+choose_n_1 1830-1852)
//   K5 ph_merge_rows_kernel    the row insertions         lib.rs:797-815, 1118-1147
//      ph_row_dist_kernel      occupant distances         lib.rs:1128-1133
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "phnsw_device.h"

// group.hip (rocPRIM sort / scan)
int ph_sort_u32_host(uint32_t *keys, uint32_t n);
int ph_exclusive_scan_u32(const uint32_t *cnt, uint32_t n_plus_1, uint32_t *start, hipStream_t st);

// ------------------------------------------------------------------ small helpers

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  int alloc(size_t count) {
    n = count;
    hipError_t e = ph_pool_alloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) {
      p = nullptr;
      return ph_hip_fail(e, "hipMalloc (build)", __FILE__, __LINE__);
    }
    return 0;
  }
  ~DevBuf() {
    if (p) ph_pool_free(p);
  }
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
};

#define PH_TRY(x)          \
  do {                     \
    int rc__ = (x);        \
    if (rc__) return rc__; \
  } while (0)

#include <chrono>
static bool ph_verbose() {
  static int v = -1;
  if (v < 0) v = getenv("PHNSW_VERBOSE") ? 1 : 0;
  return v == 1;
}
struct PhTimer {
  const char *what;
  uint64_t n;
  std::chrono::steady_clock::time_point t0;
  PhTimer(const char *w, uint64_t n_) : what(w), n(n_), t0(std::chrono::steady_clock::now()) {}
  ~PhTimer() {
    if (!ph_verbose()) return;
    hipDeviceSynchronize();
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "[phnsw] %-28s n=%-9llu %.3f s\n", what, (unsigned long long)n, s);
  }
};

static inline uint64_t stream_next(uint64_t *state) {
  *state += 0x9E3779B97F4A7C15ULL;
  return ph_mix64(*state);
}

// slice.shuffle(rng) shape (Fisher-Yates from the top); stream shared by definition with
// the oracle.  Replaces thread_rng()/StdRng (lib.rs:832, 1468-1481): "parity unpinned".
void ph_shuffle_u64(uint64_t *v, uint64_t n, uint64_t seed) {
  uint64_t st = ph_mix64(seed ^ 0x5851F42D4C957F2DULL);
  for (uint64_t i = n; i-- > 1;) {
    uint64_t j = ph_mulhi64(stream_next(&st), i + 1);
    std::swap(v[i], v[j]);
  }
}

// calculate_partitions  lib.rs:1883-1899 (f32 arithmetic as written there)
static std::vector<uint64_t> calculate_partitions(uint64_t total, uint64_t order) {
  float lc = ceilf(logf((float)total) / logf((float)order));
  uint64_t layer_count = (lc != lc || lc < 0.0f) ? 0 : (uint64_t)lc;
  layer_count = std::max<uint64_t>(1, std::min<uint64_t>(layer_count, PH_MAX_LAYERS));
  std::vector<uint64_t> p;
  uint64_t size = total;
  for (uint64_t i = 0; i < layer_count; i++) {
    p.push_back(size);
    size /= order;
  }
  std::reverse(p.begin(), p.end());
  return p;
}

__device__ __forceinline__ float key_dist(uint64_t key) {
  uint32_t fk = (uint32_t)(key >> 32);
  uint32_t u = (fk & 0x80000000u) ? (fk ^ 0x80000000u) : ~fk;
  return __uint_as_float(u);
}

// ------------------------------------------------------------------ kernels

// search results (VectorIds) -> this layer's NodeIds, self dropped
// initial_vector_distances + the binary_search map  search.rs:54-62, 73-82
__global__ void ph_init_from_search_kernel(const uint32_t *nodes, uint32_t count, uint32_t n,
                                           const uint32_t *vec2node, const uint32_t *res_ids, const float *res_d,
                                           const uint32_t *res_len, uint32_t K, uint32_t *init_ids, float *init_d,
                                           uint32_t *init_len, uint32_t *bad) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint32_t self = nodes[i];
  uint32_t m = 0, len = res_len[i];
  for (uint32_t k = 0; k < len && k < K; k++) {
    uint32_t vid = res_ids[(uint64_t)i * K + k];
    if (vid == self) continue;
    uint32_t nid = vec2node ? vec2node[vid] : vid;
    if (nid >= n) {
      atomicAdd(bad, 1u);
      continue;
    }
    init_ids[(uint64_t)i * K + m] = nid;
    init_d[(uint64_t)i * K + m] = res_d[(uint64_t)i * K + k];
    m++;
  }
  init_len[i] = m;
}

struct PhSeedArgs {
  PhDistArgs dist;
  const uint32_t *nodes;  // NodeId -> VectorId of the layer being built
  uint32_t n, W, K;
  const uint32_t *init_ids;
  const float *init_d;
  const uint32_t *init_len;
  const uint32_t *gm;      // group members (node ids) in group order
  const uint32_t *gstart;  // [n+1] per key node; slot n = the None group
  const uint32_t *gsize;
  uint64_t layer_count;  // self.layer_count() in the seed expression  lib.rs:729-731
  uint64_t seed;
  uint32_t first, count;  // node range handled by this launch
  const uint32_t *order;  // nullable: processing order within the range (locality schedule)
  uint32_t *rows;  // [count][W], row of node i at (i - first)
  float *rows_d;
};

#define SEED_CMAX 448  // >= 5*64 + 64 + slack

// K3: one wave per node: candidates = supers U picks from the supers' partitions, a
// distance batch, then sort (d,id) / dedup / drop self / take W  (lib.rs:719-787)
template <class Dist>
__global__ __launch_bounds__(64) void ph_seed_rows_kernel(PhSeedArgs a) {
  __shared__ uint64_t keys[SEED_CMAX];
  __shared__ uint64_t sorted[SEED_CMAX];
  extern __shared__ float dist_lds[];  // DistPQ table
  const uint32_t lane = threadIdx.x;
  const uint64_t lt = lanemask_lt(lane);
  for (uint32_t j = blockIdx.x; j < a.count; j += gridDim.x) {
    const uint32_t i = a.first + (a.order ? a.order[j] : j);  // cell order: neighbouring nodes pick from the same partitions
    const uint32_t self_vec = a.nodes[i];
    Dist dist;
    dist.prepare_stored(a.dist, self_vec, dist_lds, lane);
    const uint32_t len = a.init_len[i];
    uint32_t sid = 0;
    float sd = 0.f;
    uint32_t psz = 0, pst = 0;
    if (lane < len) {
      sid = a.init_ids[(uint64_t)i * a.K + lane];
      sd = a.init_d[(uint64_t)i * a.K + lane];
      psz = a.gsize[sid];  // filter_map(|n| partition_groups.get(&Some(n)))  lib.rs:735-738
      pst = a.gstart[sid];
      keys[lane] = mkkey(sd, sid);  // distances.clone(): the supers are candidates too
    }
    uint64_t pm = __ballot(psz > 0);
    uint32_t size0;
    uint64_t total = 0;
    if (pm == 0) {
      // "probably we're in the top layer. best add ourselves."  lib.rs:739-742 (the None group)
      pst = a.gstart[a.n];
      psz = a.gsize[a.n];
      pm = 1ull;
      if (lane != 0) psz = 0;
    }
    size0 = rl32(psz, __builtin_ctzll(pm));
    {
      uint64_t rem = pm;
      while (rem) {
        int j = __builtin_ctzll(rem);
        rem &= rem - 1;
        total += rl32(psz, j);
      }
    }
    const uint64_t choice_count = std::min<uint64_t>((uint64_t)a.W * 5, total);  // lib.rs:745-746
    const uint32_t excl = i < size0 ? 1u : 0u;  // choose_n_1 skips (partition 0, index == node_id.0)
    const uint64_t domain = total - excl;
    const uint32_t picks = (uint32_t)std::min<uint64_t>(choice_count, domain);
    const uint64_t rkey =
        ph_mix64(a.layer_count + (uint64_t)self_vec + (uint64_t)a.n) ^ ph_mix64(a.seed + 0x632BE59BD9B4E019ULL);
    const uint32_t ncand = len + picks;

    for (uint32_t base = 0; base < picks; base += 64) {
      uint32_t k = base + lane;
      bool act = k < picks;
      uint32_t member = 0, vm = 0;
      if (act) {
        uint64_t f = ph_feistel_perm(k, domain, rkey);
        if (excl && f >= i) f += 1;
        uint32_t midx = 0;
        bool found = false;
        uint64_t rem = pm;
        while (rem) {
          int j = __builtin_ctzll(rem);
          rem &= rem - 1;
          uint32_t s = rl32(psz, j), st = rl32(pst, j);
          if (!found) {
            if (f < s) {
              midx = st + (uint32_t)f;
              found = true;
            } else
              f -= s;
          }
        }
        member = a.gm[midx];
        vm = a.nodes[member];
      } else {
        // keep the readlane loop wave-uniform: inactive lanes still run it (nothing to do)
      }
      // compare_vec(Stored(vector_id), Stored(choice.1))  lib.rs:750-754
      const float myd = dist.batch(a.dist, __ballot(act), vm, lane);
      if (act) keys[len + k] = mkkey(myd, member);
    }
    __syncthreads();
    // distances.sort_by_key(|d| (OrderedFloat(d.1), d.0))  lib.rs:757 -- rank by counting
    for (uint32_t c = lane; c < ncand; c += 64) {
      uint64_t kc = keys[c];
      uint32_t rank = 0;
      for (uint32_t e = 0; e < ncand; e++) {
        uint64_t ke = keys[e];
        rank += (ke < kc || (ke == kc && e < c)) ? 1u : 0u;
      }
      sorted[rank] = kc;
    }
    __syncthreads();
    // dedup(); filter(node_id != n); take(W); pad  lib.rs:758-764
    uint32_t outn = 0;
    for (uint32_t base = 0; base < ncand; base += 64) {
      uint32_t j = base + lane;
      bool keep = false;
      uint64_t kj = KEY_NONE;
      if (j < ncand) {
        kj = sorted[j];
        keep = (j == 0 || sorted[j - 1] != kj) && ((uint32_t)kj & IDM) != i;
      }
      uint64_t km = __ballot(keep);
      uint32_t at = outn + __popcll(km & lt);
      if (keep && at < a.W) {
        a.rows[(uint64_t)(i - a.first) * a.W + at] = (uint32_t)kj & IDM;
        a.rows_d[(uint64_t)(i - a.first) * a.W + at] = key_dist(kj);
      }
      outn += __popcll(km);
    }
    for (uint32_t j = std::min(outn, a.W) + lane; j < a.W; j += 64) {
      a.rows[(uint64_t)(i - a.first) * a.W + j] = PH_EMPTY32;
      a.rows_d[(uint64_t)(i - a.first) * a.W + j] = PH_FMAX;
    }
    __syncthreads();
  }
}

// proposals live in fixed-stride slot arrays: slot (s, k) proposes "insert node s into row
// tgt[s*S+k] at distance d[s*S+k]" (PH_EMPTY32 = no proposal)
__global__ void ph_count_targets_kernel(const uint32_t *tgt, uint64_t nslots, uint32_t n, uint32_t *cnt) {
  for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < nslots; x += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t t = tgt[x];
    if (t < n) atomicAdd(&cnt[t], 1u);
  }
}

__global__ void ph_fill_targets_kernel(const uint32_t *tgt, const float *d, uint64_t nslots, uint32_t S, uint32_t n,
                                       const uint32_t *start, uint32_t *cursor, uint32_t *inc_src, float *inc_d) {
  for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < nslots; x += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t t = tgt[x];
    if (t < n) {
      uint32_t at = start[t] + atomicAdd(&cursor[t], 1u);
      inc_src[at] = (uint32_t)(x / S);
      inc_d[at] = d[x];
    }
  }
}

// K5: row'[t] = best-W by (d,id) of row[t] U incoming[t]; one wave per target row.
// The incoming list is unordered (filled with atomics); the result does not depend on it.
__global__ __launch_bounds__(64) void ph_merge_rows_kernel(uint32_t n, uint32_t W, uint32_t *rows, float *rows_d,
                                                           const uint32_t *start, const uint32_t *inc_src,
                                                           const float *inc_d, unsigned long long *added) {
  __shared__ uint64_t lk[64];
  __shared__ uint32_t lnew[64];
  const uint32_t lane = threadIdx.x;
  for (uint32_t t = blockIdx.x; t < n; t += gridDim.x) {
    uint32_t s0 = start[t], cnt = start[t + 1] - s0;
    if (cnt == 0) continue;
    uint64_t cur = KEY_NONE;
    uint32_t isnew = 0;
    if (lane < W) {
      uint32_t id = rows[(uint64_t)t * W + lane];
      if (id != PH_EMPTY32) cur = mkkey(rows_d[(uint64_t)t * W + lane], id);
    }
    {
      // rows adopted from elsewhere need not be (d,id)-sorted: order them first
      uint32_t r = 0;
      for (uint32_t j = 0; j < W; j++) r += (rl64(cur, j) < cur) ? 1u : 0u;
      lk[lane] = KEY_NONE;
      __syncthreads();
      if (cur != KEY_NONE) lk[r] = cur;
      __syncthreads();
      cur = lk[lane];
      __syncthreads();
    }
    for (uint32_t base = 0; base < cnt; base += 64) {
      uint64_t e = KEY_NONE;
      if (base + lane < cnt) e = mkkey(inc_d[s0 + base + lane], inc_src[s0 + base + lane]);
      // against the current row: duplicates (same id => same distance) and rank
      bool dup = false;
      uint32_t lt_e = 0;
      for (uint32_t j = 0; j < W; j++) {
        uint64_t cj = rl64(cur, j);
        dup |= (cj == e);
        lt_e += (cj < e) ? 1u : 0u;
      }
      bool valid = e != KEY_NONE && !dup && ((uint32_t)e & IDM) != t;
      uint64_t vm = __ballot(valid);
      uint32_t rank_e = 0, shift = 0;
      uint64_t rem = vm;
      while (rem) {
        int j = __builtin_ctzll(rem);
        rem &= rem - 1;
        uint64_t ej = rl64(e, j);
        rank_e += (ej < e) ? 1u : 0u;
        shift += (ej < cur) ? 1u : 0u;
      }
      lk[lane] = KEY_NONE;
      lnew[lane] = 0;
      __syncthreads();
      if (cur != KEY_NONE && lane + shift < W) {
        lk[lane + shift] = cur;
        lnew[lane + shift] = isnew;
      }
      if (valid && lt_e + rank_e < W) {
        lk[lt_e + rank_e] = e;
        lnew[lt_e + rank_e] = 1;
      }
      __syncthreads();
      cur = lk[lane];
      isnew = lnew[lane];
      __syncthreads();
    }
    if (lane < W) {
      bool live = cur != KEY_NONE;
      rows[(uint64_t)t * W + lane] = live ? ((uint32_t)cur & IDM) : PH_EMPTY32;
      rows_d[(uint64_t)t * W + lane] = live ? key_dist(cur) : PH_FMAX;
    }
    uint64_t nm = __ballot(lane < W && cur != KEY_NONE && isnew);
    if (lane == 0 && nm) atomicAdd(added, (unsigned long long)__popcll(nm));
  }
}

// distance of every occupant to its row owner (the values lib.rs:1128-1133 recomputes);
// one wave per row
template <class Dist>
__global__ __launch_bounds__(64) void ph_row_dist_kernel(PhDistArgs da, const uint32_t *nodes, uint32_t n, uint32_t W,
                                                         const uint32_t *rows, float *rows_d) {
  extern __shared__ float dist_lds[];
  const uint32_t lane = threadIdx.x;
  for (uint32_t t = blockIdx.x; t < n; t += gridDim.x) {
    Dist dist;
    dist.prepare_stored(da, nodes[t], dist_lds, lane);
    uint32_t o = lane < W ? rows[(uint64_t)t * W + lane] : PH_EMPTY32;
    uint32_t vo = o < n ? nodes[o] : 0;
    float myd = dist.batch(da, __ballot(o < n), vo, lane);
    if (!(o < n)) myd = PH_FMAX;
    if (lane < W) rows_d[(uint64_t)t * W + lane] = myd;
    __syncthreads();
  }
}

// promotion thinning (filter_promotion_candidates): one wave per candidate j computes compare_vec(cand[j], sel[k]) for
// every k -- 64 at a time, one per lane, through the store's distance policy -- and either reports whether some
// distance is below that vector's radius (cov) or writes the whole row (table[j][k])
#define PH_THIN_BLOCK 2048u
template <class Dist>
__global__ __launch_bounds__(64) void ph_cover_kernel(PhDistArgs da, const uint32_t *cand, uint32_t n_cand,
                                                      const uint32_t *sel, uint32_t n_sel, const float *radius,
                                                      uint32_t *cov, float *table) {
  extern __shared__ float dist_lds[];
  const uint32_t lane = threadIdx.x;
  for (uint32_t j = blockIdx.x; j < n_cand; j += gridDim.x) {
    Dist dist;
    dist.prepare_stored(da, cand[j], dist_lds, lane);
    bool any = false;
    for (uint32_t base = 0; base < n_sel; base += 64) {
      const uint32_t k = base + lane;
      const bool live = k < n_sel;
      const uint32_t vid = live ? sel[k] : 0u;
      const float d = dist.batch(da, __ballot(live), vid, lane);
      if (live && table) table[(uint64_t)j * n_sel + k] = d;
      if (live && radius) any |= d < radius[k];
    }
    const uint64_t bm = __ballot(any);
    if (cov && lane == 0) cov[j] = bm ? 1u : 0u;
    __syncthreads();
  }
}

// top-M search results (VectorIds) of node i -> proposals "insert i into row of result k"
// for (neighbor_vec, distance) in matches.take(M) { if neighbor_vec == vector { break } .. }  lib.rs:1118-1122
// (ids arrive from other ranks in the sharded build: anything outside the store is dropped, never dereferenced)
__global__ void ph_link_targets_kernel(const uint32_t *nodes, uint32_t n, const uint32_t *vec2node, uint32_t n_store,
                                       const uint32_t *res_ids, const uint32_t *res_len, uint32_t M, uint32_t *tgt) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t self = nodes[i], len = res_len[i];
  bool stop = false;
  for (uint32_t k = 0; k < M; k++) {
    uint32_t t = PH_EMPTY32;
    if (!stop && k < len) {
      uint32_t vid = res_ids[(uint64_t)i * M + k];
      if (vid == self)
        stop = true;
      else if (vid < n_store)
        t = vec2node ? vec2node[vid] : vid;
    }
    tgt[(uint64_t)i * M + k] = t;
  }
}

// ------------------------------------------------------------------ host: shared steps

static int nv_for(uint32_t nv4) { return nv4 <= 64 ? 1 : (nv4 <= 192 ? 3 : (nv4 <= 384 ? 6 : 0)); }
static bool store_supported(const phnsw_store *s) { return s->codes ? true : nv_for(s->ld / 4) != 0; }
// resident one-wave blocks a PQ kernel can have per chip: the lookup table dominates the LDS
static uint32_t pq_grid(const phnsw_store *s, uint32_t want) {
  size_t per = ph_pq_lds_bytes(s) + 8 * 1024;
  uint32_t per_cu = (uint32_t)std::max<size_t>(1, (160 * 1024) / per);
  return std::min<uint32_t>(want, 256u * std::min<uint32_t>(per_cu, 16u));
}

template <typename K1, typename K3, typename K6, typename KQ, typename... Args>
static int launch_by_policy(const phnsw_store *s, dim3 g, K1 k1, K3 k3, K6 k6, KQ kq, Args... args) {
  if (s->codes) {
    size_t lds = ph_pq_lds_bytes(s);
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void *)kq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return ph_hip_fail(e, "hipFuncSetAttribute(LDS)", __FILE__, __LINE__);
    }
    hipLaunchKernelGGL(kq, g, dim3(64), lds, 0, args...);
  } else {
    switch (nv_for(s->ld / 4)) {
      case 1:
        hipLaunchKernelGGL(k1, g, dim3(64), 0, 0, args...);
        break;
      case 3:
        hipLaunchKernelGGL(k3, g, dim3(64), 0, 0, args...);
        break;
      case 6:
        hipLaunchKernelGGL(k6, g, dim3(64), 0, 0, args...);
        break;
      default:
        ph_set_error("dim %u unsupported (max 1536)", s->dim);
        return PHNSW_E_UNSUPPORTED;
    }
  }
  PH_HIP(hipGetLastError());
  return 0;
}

static uint32_t wave_grid(uint32_t n) {
  return std::min<uint32_t>(n, 256u * 32u);
}

// apply proposal slots to the rows of `L`: count, scan, fill, merge (K5)
static int apply_proposals(PhLayerHost &L, const uint32_t *tgt, const float *d, uint32_t S, uint64_t *out_added) {
  uint32_t n = L.n_nodes;
  PhTimer tm(" apply_proposals(K5)", n);
  uint64_t nslots = (uint64_t)n * S;
  DevBuf<uint32_t> cnt, start, cursor, inc_src;
  DevBuf<float> inc_d;
  DevBuf<unsigned long long> added;
  PH_TRY(cnt.alloc(n + 1));
  PH_TRY(start.alloc(n + 1));
  PH_TRY(cursor.alloc(n + 1));
  PH_TRY(added.alloc(1));
  PH_HIP(hipMemsetAsync(cnt.p, 0, (size_t)(n + 1) * 4, 0));
  PH_HIP(hipMemsetAsync(cursor.p, 0, (size_t)(n + 1) * 4, 0));
  PH_HIP(hipMemsetAsync(added.p, 0, 8, 0));
  uint32_t blocks = (uint32_t)std::min<uint64_t>((nslots + 255) / 256, 8192);
  hipLaunchKernelGGL(ph_count_targets_kernel, dim3(blocks), dim3(256), 0, 0, tgt, nslots, n, cnt.p);
  PH_TRY(ph_exclusive_scan_u32(cnt.p, n + 1, start.p, 0));  // cnt[n] == 0, so start[n] = total
  PH_HIP(hipGetLastError());
  uint32_t total = 0;
  PH_HIP(hipMemcpy(&total, start.p + n, 4, hipMemcpyDeviceToHost));
  PH_TRY(inc_src.alloc(total));
  PH_TRY(inc_d.alloc(total));
  hipLaunchKernelGGL(ph_fill_targets_kernel, dim3(blocks), dim3(256), 0, 0, tgt, d, nslots, S, n, start.p, cursor.p,
                     inc_src.p, inc_d.p);
  hipLaunchKernelGGL(ph_merge_rows_kernel, dim3(wave_grid(n)), dim3(64), 0, 0, n, L.W, L.neighbors, L.nbr_dist,
                     start.p, inc_src.p, inc_d.p, added.p);
  PH_HIP(hipGetLastError());
  unsigned long long h = 0;
  PH_HIP(hipMemcpy(&h, added.p, 8, hipMemcpyDeviceToHost));
  if (out_added) *out_added = h;
  return 0;
}

static int ensure_row_dist(phnsw_index *ix, PhLayerHost &L) {
  if (L.nbr_dist) return 0;
  const phnsw_store *s = ix->store;
  PH_HIP(hipMalloc(&L.nbr_dist, (size_t)L.n_nodes * L.W * 4));
  // a PQ table takes most of a CU's LDS: one resident wave per CU is all that fits
  dim3 g(s->codes ? pq_grid(s, L.n_nodes) : wave_grid(L.n_nodes));
  return launch_by_policy(s, g, ph_row_dist_kernel<DistF32<1>>, ph_row_dist_kernel<DistF32<3>>,
                          ph_row_dist_kernel<DistF32<6>>, ph_row_dist_kernel<DistPQ>, ph_dist_args(s), L.nodes, L.n_nodes,
                          L.W, L.neighbors, L.nbr_dist);
}

// run the batched search for Stored queries and surface per-query failures
static PhRowHint row_hint(const PhLayerHost &L, uint32_t first, uint32_t count, bool contiguous) {
  PhRowHint h;
  h.qnodes = L.nodes;
  h.qn = L.n_nodes;
  h.vec2node = L.identity ? nullptr : L.vec2node;
  h.first = first;
  h.count = count;
  h.contiguous = contiguous;
  return h;
}

static int search_stored(const phnsw_index *ix, const uint32_t *qids_dev, uint32_t nq, const phnsw_search_params *sp,
                         uint32_t upto, const uint32_t *exclude_dev, uint32_t *out_ids, float *out_d,
                         uint32_t *out_len, uint32_t out_stride, uint32_t *out_hit, const uint32_t *order = nullptr,
                         const PhRowHint *hint = nullptr) {
  if (sp->number_of_candidates == 0 || sp->number_of_candidates > 1024 || sp->probe_depth == 0) {
    ph_set_error("build: search parameters out of range (number_of_candidates 1..1024, probe_depth >= 1)");
    return PHNSW_E_INVALID;
  }
  DevBuf<uint32_t> status;
  PH_TRY(status.alloc(nq));
  uint32_t ovf_cap = ph_default_ovf_cap((uint32_t)sp->number_of_candidates);
  for (int attempt = 0; attempt < 3; attempt++) {
    PH_TRY(ph_search_device(ix, nullptr, 0, qids_dev, nq, sp, upto, exclude_dev, out_ids, out_d, out_len, nullptr,
                            status.p, ovf_cap, 0, 0, out_stride, out_hit, 0.f, 0, 0.f, order, nullptr, hint));
    PH_HIP(hipDeviceSynchronize());
    std::vector<uint32_t> h(nq);
    PH_HIP(hipMemcpy(h.data(), status.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    bool overflow = false;
    for (uint32_t i = 0; i < nq; i++) {
      if (h[i] == 4) {
        ph_set_error("build: a candidate vector is missing from a lower layer (layers not nested, lib.rs:261)");
        return PHNSW_E_MISSING_NODE;
      }
      overflow |= (h[i] == 5);
    }
    if (!overflow) return 0;
    ovf_cap *= 8;  // rare: rerun the whole batch with more spill room (results are deterministic)
  }
  ph_set_error("build: frontier spill exceeded %u entries", ovf_cap);
  return PHNSW_E_OVERFLOW;
}

// ------------------------------------------------------------------ generate_layer

struct HostPair {
  float d;
  uint32_t id;
};
static inline bool pair_less(const HostPair &a, const HostPair &b) { return a.d < b.d || (a.d == b.d && a.id < b.id); }

// host form of "sort (d,id), dedup, drop self, take W, pad" for the tiny top layer
static void finish_row_host(std::vector<HostPair> &list, uint32_t self, uint32_t W, uint32_t *ids, float *d) {
  std::sort(list.begin(), list.end(), pair_less);
  uint32_t out = 0;
  for (size_t k = 0; k < list.size() && out < W; k++) {
    if (k > 0 && list[k].id == list[k - 1].id && list[k].d == list[k - 1].d) continue;
    if (list[k].id == self) continue;
    ids[out] = list[k].id;
    d[out] = list[k].d;
    out++;
  }
  for (; out < W; out++) {
    ids[out] = PH_EMPTY32;
    d[out] = PH_FMAX;
  }
}

// groups keyed by the nearest super node (lib.rs:711-713); member order = (first distance,
// node id), the deterministic form of the unstable sort in search.rs:67-69
static void build_groups(uint32_t n, const std::vector<uint32_t> &key, const std::vector<float> &keyd,
                         std::vector<uint32_t> &gm, std::vector<uint32_t> &gstart, std::vector<uint32_t> &gsize) {
  gm.resize(n);
  std::iota(gm.begin(), gm.end(), 0u);
  std::sort(gm.begin(), gm.end(), [&](uint32_t x, uint32_t y) {
    if (key[x] != key[y]) return key[x] < key[y];  // PH_EMPTY32 (None) sorts last
    if (keyd[x] != keyd[y]) return keyd[x] < keyd[y];
    return x < y;
  });
  gstart.assign(n + 1, 0);
  gsize.assign(n + 1, 0);
  for (uint32_t p = 0; p < n; p++) {
    uint32_t slot = key[gm[p]] == PH_EMPTY32 ? n : key[gm[p]];
    if (gsize[slot] == 0) gstart[slot] = p;
    gsize[slot]++;
  }
}

// the first layer of a stack has no layers above: compare_all (search.rs:13-30) gives every
// node all other nodes; n < order here, so the distances come from n K1 launches and the
// (integer) selection runs on the host
static int generate_first_layer(phnsw_index *ix, const std::vector<uint32_t> &nodes, uint32_t W,
                                const phnsw_build_params *bp, std::vector<uint32_t> &rows, std::vector<float> &rows_d) {
  const phnsw_store *s = ix->store;
  uint32_t n = (uint32_t)nodes.size();
  if (n > 16384) {
    ph_set_error("first layer has %u nodes; the all-pairs seeding (search.rs:46-48) is limited to 16384", n);
    return PHNSW_E_UNSUPPORTED;
  }
  DevBuf<uint32_t> ids;
  DevBuf<float> out;
  PH_TRY(ids.alloc(n));
  PH_TRY(out.alloc((size_t)n * n));
  PH_HIP(hipMemcpy(ids.p, nodes.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  for (uint32_t i = 0; i < n; i++)
    PH_TRY(ph_distance_batch(s, nullptr, nodes[i], ids.p, n, out.p + (size_t)i * n, 0));
  std::vector<float> D((size_t)n * n);
  PH_HIP(hipMemcpy(D.data(), out.p, D.size() * 4, hipMemcpyDeviceToHost));
  std::vector<std::vector<HostPair>> init(n);
  std::vector<uint32_t> key(n, PH_EMPTY32);
  std::vector<float> keyd(n, 0.f);
  for (uint32_t i = 0; i < n; i++) {
    for (uint32_t j = 0; j < n; j++)
      if (j != i) init[i].push_back({D[(size_t)i * n + j], j});
    std::sort(init[i].begin(), init[i].end(), pair_less);
    if (!init[i].empty()) {
      key[i] = init[i][0].id;
      keyd[i] = init[i][0].d;
    }
  }
  std::vector<uint32_t> gm, gstart, gsize;
  build_groups(n, key, keyd, gm, gstart, gsize);
  rows.assign((size_t)n * W, PH_EMPTY32);
  rows_d.assign((size_t)n * W, PH_FMAX);
  uint64_t layer_count = ix->layers.size();
  for (uint32_t i = 0; i < n; i++) {
    std::vector<HostPair> list = init[i];
    std::vector<std::pair<uint32_t, uint32_t>> parts;
    uint64_t total = 0;
    for (auto &p : init[i])
      if (gsize[p.id]) {
        parts.push_back({gstart[p.id], gsize[p.id]});
        total += gsize[p.id];
      }
    if (parts.empty()) {
      uint32_t slot = init[i].empty() ? n : init[i][0].id;
      parts.push_back({gstart[slot], gsize[slot]});
      total = gsize[slot];
    }
    uint64_t choice_count = std::min<uint64_t>((uint64_t)W * 5, total);
    uint64_t excl = i < parts[0].second ? 1 : 0;
    uint64_t domain = total - excl;
    uint64_t picks = std::min(choice_count, domain);
    uint64_t rkey = ph_mix64(layer_count + (uint64_t)nodes[i] + (uint64_t)n) ^ ph_mix64(bp->seed + 0x632BE59BD9B4E019ULL);
    for (uint64_t k = 0; k < picks; k++) {
      uint64_t f = ph_feistel_perm(k, domain, rkey);
      if (excl && f >= i) f += 1;
      size_t p = 0;
      while (f >= parts[p].second) {
        f -= parts[p].second;
        p++;
      }
      uint32_t member = gm[parts[p].first + f];
      list.push_back({D[(size_t)i * n + member], member});
    }
    finish_row_host(list, i, W, &rows[(size_t)i * W], &rows_d[(size_t)i * W]);
  }
  return 0;
}

// ---- generate_layer in phases.  A driver that shards the node range over several GPUs
// calls begin / init_search(range) / seed(range) / finish with an all-gather between the
// phases (parallel_hnsw_amd/sharded.py); the single-GPU entry point runs the same phases
// over the whole range.

int ph_build_groups_device(const uint32_t *init_ids, const float *init_d, const uint32_t *init_len, uint32_t K,
                           uint32_t n, uint32_t *gm, uint32_t *gstart, uint32_t *gsize);  // group.hip

struct PhPendingLayer {
  PhLayerHost L;
  uint32_t K = 0;
  bool grouped = false;
  DevBuf<uint32_t> gm, gstart, gsize;
};

static void pending_drop(phnsw_index *ix) {
  if (ix->pending) {
    ph_layer_free(ix->pending->L);
    delete ix->pending;
    ix->pending = nullptr;
  }
}
void ph_pending_free(phnsw_index *ix) { pending_drop(ix); }

// begin: validate, sort, upload nodes + id map, allocate rows.  *needs_phases = 0 when the
// layer was completed here (first layer of a stack: all-pairs seeding, n < order).
// Row maintenance keeps, per row, each occupant's distance to the row owner and merges
// proposals by (distance, id): that needs d(a, b) == d(b, a) bit for bit.  The 8-bit PQ table
// scales by the QUERY's table, so it is for searching a graph built in mode 0 or 1.
static int symmetric_store(const phnsw_store *s) {
  if (s->codes16) {
    ph_set_error("a shared-codebook PQ store is searched, not built on: build the index over "
                 "phnsw_pq_shared_reconstruct_store (identical distances) and adopt it with phnsw_index_from_layers");
    return PHNSW_E_UNSUPPORTED;
  }
  if (s->codes && s->pq_table_f16 == 2) {
    ph_set_error("8-bit PQ tables (phnsw_pq_set_table_mode 2) are asymmetric: build / link in mode 0 or 1, then switch");
    return PHNSW_E_UNSUPPORTED;
  }
  return 0;
}

static int layer_begin_impl(phnsw_index *ix, const uint64_t *vids, uint64_t n64, uint64_t W64,
                            const phnsw_build_params *bp, int *needs_phases, int *defer_cells = nullptr) {
  const phnsw_store *s = ix->store;
  PH_TRY(symmetric_store(s));
  if (!vids || n64 == 0 || W64 == 0 || W64 > 64 || n64 >= 0x7FFFFFFFull || !bp) {
    ph_set_error("generate_layer: need 1 <= n < 2^31 nodes and 1 <= neighborhood_size <= 64");
    return PHNSW_E_INVALID;
  }
  if (ix->layers.size() >= PH_MAX_LAYERS) {
    ph_set_error("generate_layer: more than %d layers", PH_MAX_LAYERS);
    return PHNSW_E_UNSUPPORTED;
  }
  if (!store_supported(s)) {
    ph_set_error("dim %u unsupported (max 1536)", s->dim);
    return PHNSW_E_UNSUPPORTED;
  }
  const uint32_t K = (uint32_t)bp->initial_partition_search.number_of_candidates;
  if (!ix->layers.empty() && (K == 0 || K > 64)) {
    ph_set_error("initial_partition_search.number_of_candidates must be 1..64 (got %u)", K);
    return PHNSW_E_UNSUPPORTED;
  }
  pending_drop(ix);
  uint32_t n = (uint32_t)n64, W = (uint32_t)W64;
  PhTimer tb(" layer_begin", n);
  std::vector<uint32_t> nodes(n);
  for (uint32_t i = 0; i < n; i++) {
    if (vids[i] >= s->n) {
      ph_set_error("generate_layer: VectorId %llu outside the store", (unsigned long long)vids[i]);
      return PHNSW_E_INVALID;
    }
    nodes[i] = (uint32_t)vids[i];
  }
  if (!std::is_sorted(nodes.begin(), nodes.end())) {  // vs.sort()  lib.rs:685
    if (n >= 65536)
      PH_TRY(ph_sort_u32_host(nodes.data(), n));
    else
      std::sort(nodes.begin(), nodes.end());
  }
  for (uint32_t i = 1; i < n; i++)
    if (nodes[i] == nodes[i - 1]) {
      ph_set_error("generate_layer: duplicate VectorId %u", nodes[i]);
      return PHNSW_E_INVALID;
    }
  PhPendingLayer *P = new PhPendingLayer();
  int rc;
  {
    PhTimer tu("  layer arrays (alloc + fill)", n);
    rc = ph_layer_upload(ix, nodes.data(), nullptr, n, W, &P->L);  // rows start empty
  }
  if (!rc) {
    PhTimer tu("  nbr_dist alloc", n);
    hipError_t e = hipMalloc(&P->L.nbr_dist, (size_t)n * W * 4);
    if (e != hipSuccess) rc = ph_hip_fail(e, "nbr_dist alloc", __FILE__, __LINE__);
  }
  if (rc) {
    ph_layer_free(P->L);
    delete P;
    return rc;
  }
  if (defer_cells) *defer_cells = 0;
  if (!rc) {
    PhTimer ta("  layer cells (anchor GEMM)", n);
    // locality schedule of the build rounds (bruteforce.hip); a sharded driver computes it range by range instead
    if (defer_cells && !ix->layers.empty() && ph_layer_wants_cells(ix->store, n))
      *defer_cells = 1;
    else
      rc = ph_layer_anchor_pos(ix->store, P->L);
  }
  if (rc) {
    ph_layer_free(P->L);
    delete P;
    return rc;
  }
  P->K = K;
  ix->pending = P;
  if (ix->layers.empty()) {
    std::vector<uint32_t> rows;
    std::vector<float> rows_d;
    rc = generate_first_layer(ix, nodes, W, bp, rows, rows_d);
    DevBuf<uint32_t> r;
    DevBuf<float> rd;
    if (!rc) rc = r.alloc(rows.size());
    if (!rc) rc = rd.alloc(rows_d.size());
    if (!rc) {
      hipError_t e = hipMemcpy(r.p, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(rd.p, rows_d.data(), rows_d.size() * 4, hipMemcpyHostToDevice);
      if (e != hipSuccess) rc = ph_hip_fail(e, "first layer upload", __FILE__, __LINE__);
    }
    int layer_finish_impl(phnsw_index *, const uint32_t *, const float *);
    if (!rc) rc = layer_finish_impl(ix, r.p, rd.p);
    if (rc) pending_drop(ix);
    *needs_phases = 0;
    return rc;
  }
  *needs_phases = 1;
  return 0;
}

// 1. generate_initial_partitions for nodes [first, first+count)  search.rs:32-71:
// out_* are [count][K] (NodeIds of the new layer, self dropped) + [count] lengths
static int layer_init_search_impl(phnsw_index *ix, const phnsw_build_params *bp, uint32_t first, uint32_t count,
                                  uint32_t *out_ids, float *out_d, uint32_t *out_len) {
  PhPendingLayer *P = ix->pending;
  if (!P || first + (uint64_t)count > P->L.n_nodes) {
    ph_set_error("layer_init_search: no pending layer or range out of bounds");
    return PHNSW_E_INVALID;
  }
  if (count == 0) return 0;
  PhTimer tm(" layer_init_search", count);
  const uint32_t K = P->K;
  DevBuf<uint32_t> res_ids, res_len, bad;
  DevBuf<float> res_d;
  PH_TRY(res_ids.alloc((size_t)count * K));
  PH_TRY(res_d.alloc((size_t)count * K));
  PH_TRY(res_len.alloc(count));
  PH_TRY(bad.alloc(1));
  const uint32_t *order = nullptr;
  PH_TRY(ph_layer_range_order(P->L, first, count, &order));
  const PhRowHint hint = row_hint(P->L, first, count, true);
  PH_TRY(search_stored(ix, P->L.nodes + first, count, &bp->initial_partition_search, 0, nullptr, res_ids.p, res_d.p,
                       res_len.p, 0, nullptr, order, &hint));
  PH_HIP(hipMemsetAsync(bad.p, 0, 4, 0));
  hipLaunchKernelGGL(ph_init_from_search_kernel, dim3((count + 255) / 256), dim3(256), 0, 0, P->L.nodes + first, count,
                     P->L.n_nodes, P->L.identity ? nullptr : P->L.vec2node, res_ids.p, res_d.p, res_len.p, K, out_ids,
                     out_d, out_len, bad.p);
  PH_HIP(hipGetLastError());
  uint32_t hbad = 0;
  PH_HIP(hipMemcpy(&hbad, bad.p, 4, hipMemcpyDeviceToHost));
  if (hbad) {
    ph_set_error("generate_layer: %u search results are not nodes of the new layer (layers must be nested)", hbad);
    return PHNSW_E_MISSING_NODE;
  }
  return 0;
}

// 2.+3. partition groups from the FULL init lists (host, replicated on every rank), then the
// seeding kernel K3 for nodes [first, first+count); out_rows [count][W]
static int layer_seed_impl(phnsw_index *ix, const phnsw_build_params *bp, const uint32_t *init_ids, const float *init_d,
                           const uint32_t *init_len, uint32_t first, uint32_t count, uint32_t *out_rows,
                           float *out_rows_d) {
  PhPendingLayer *P = ix->pending;
  if (!P || first + (uint64_t)count > P->L.n_nodes) {
    ph_set_error("layer_seed: no pending layer or range out of bounds");
    return PHNSW_E_INVALID;
  }
  const phnsw_store *s = ix->store;
  const uint32_t n = P->L.n_nodes, K = P->K, W = P->L.W;
  if (!P->grouped) {
    PhTimer tg(" layer_group", n);
    PH_TRY(P->gm.alloc(n));
    PH_TRY(P->gstart.alloc(n + 1));
    PH_TRY(P->gsize.alloc(n + 1));
    PH_TRY(ph_build_groups_device(init_ids, init_d, init_len, K, n, P->gm.p, P->gstart.p, P->gsize.p));  // lib.rs:711-713
    P->grouped = true;
  }
  if (count == 0) return 0;
  PhTimer ts(" layer_seed(K3)", count);
  PhSeedArgs a;
  a.dist = ph_dist_args(s);
  a.nodes = P->L.nodes;
  a.n = n;
  a.W = W;
  a.K = K;
  a.init_ids = init_ids;
  a.init_d = init_d;
  a.init_len = init_len;
  a.gm = P->gm.p;
  a.gstart = P->gstart.p;
  a.gsize = P->gsize.p;
  a.layer_count = ix->layers.size();
  a.seed = bp->seed;
  a.first = first;
  a.count = count;
  a.order = nullptr;
  PH_TRY(ph_layer_range_order(P->L, first, count, &a.order));
  a.rows = out_rows;
  a.rows_d = out_rows_d;
  dim3 g(s->codes ? pq_grid(s, count) : wave_grid(count));
  PH_TRY(launch_by_policy(s, g, ph_seed_rows_kernel<DistF32<1>>, ph_seed_rows_kernel<DistF32<3>>,
                          ph_seed_rows_kernel<DistF32<6>>, ph_seed_rows_kernel<DistPQ>, a));
  PH_HIP(hipDeviceSynchronize());
  return 0;
}

// 4. make neighbourhoods bidirectional  lib.rs:789-815: every row entry (t, d) of node i
// proposes (i, d) to row t, evaluated against the seeded rows (the snapshot); then the layer
// joins the stack
int layer_finish_impl(phnsw_index *ix, const uint32_t *rows, const float *rows_d) {
  PhPendingLayer *P = ix->pending;
  if (!P) {
    ph_set_error("layer_finish: no pending layer");
    return PHNSW_E_INVALID;
  }
  size_t cnt = (size_t)P->L.n_nodes * P->L.W;
  PH_HIP(hipMemcpy(P->L.neighbors, rows, cnt * 4, hipMemcpyDeviceToDevice));
  PH_HIP(hipMemcpy(P->L.nbr_dist, rows_d, cnt * 4, hipMemcpyDeviceToDevice));
  PH_TRY(apply_proposals(P->L, rows, rows_d, P->L.W, nullptr));
  ix->layers.push_back(P->L);
  P->L = PhLayerHost();
  pending_drop(ix);
  return 0;
}

static int generate_layer_impl(phnsw_index *ix, const uint64_t *vids, uint64_t n64, uint64_t W64,
                               const phnsw_build_params *bp) {
  PhTimer tm("generate_layer", n64);
  int phases = 0;
  PH_TRY(layer_begin_impl(ix, vids, n64, W64, bp, &phases));
  if (!phases) return 0;
  uint32_t n = (uint32_t)n64, W = (uint32_t)W64, K = ix->pending->K;
  DevBuf<uint32_t> init_ids, init_len, rows;
  DevBuf<float> init_d, rows_d;
  int rc = init_ids.alloc((size_t)n * K);
  if (!rc) rc = init_d.alloc((size_t)n * K);
  if (!rc) rc = init_len.alloc(n);
  if (!rc) rc = rows.alloc((size_t)n * W);
  if (!rc) rc = rows_d.alloc((size_t)n * W);
  if (!rc) rc = layer_init_search_impl(ix, bp, 0, n, init_ids.p, init_d.p, init_len.p);
  if (!rc) rc = layer_seed_impl(ix, bp, init_ids.p, init_d.p, init_len.p, 0, n, rows.p, rows_d.p);
  if (!rc) rc = layer_finish_impl(ix, rows.p, rows_d.p);
  if (rc) pending_drop(ix);
  return rc;
}

// ------------------------------------------------------------------ link / recall / improve

// link round, phase 1: searches of nodes [first, first+count) against the unmodified layer
// search_layers(Stored(vector), sp, &pseudo_stack, Some(vector))  lib.rs:1112-1117;
// out_ids [count][M] VectorIds of the best M results
static int link_search_impl(phnsw_index *ix, uint32_t lft, const phnsw_search_params *sp, uint64_t link_count,
                            uint32_t first, uint32_t count, uint32_t *out_ids, float *out_d, uint32_t *out_len) {
  if (lft >= ix->layers.size() || link_count == 0 || link_count > sp->number_of_candidates ||
      first + (uint64_t)count > ix->layers[lft].n_nodes) {
    ph_set_error("link_search: layer %u / range out of bounds or link_count %llu not in 1..number_of_candidates", lft,
                 (unsigned long long)link_count);
    return PHNSW_E_INVALID;
  }
  if (count == 0) return 0;
  PhLayerHost &L = ix->layers[lft];
  const uint32_t *order = nullptr;
  PH_TRY(ph_layer_range_order(L, first, count, &order));
  const PhRowHint hint = row_hint(L, first, count, true);
  return search_stored(ix, L.nodes + first, count, sp, lft + 1, L.nodes + first, out_ids, out_d, out_len,
                       (uint32_t)link_count, nullptr, order, &hint);
}

// link round, phase 2: all proposals -> rows (K5)  lib.rs:1118-1147
static int link_apply_impl(phnsw_index *ix, uint32_t lft, uint64_t link_count, const uint32_t *res_ids,
                           const float *res_d, const uint32_t *res_len, uint64_t *out_added) {
  if (lft >= ix->layers.size() || link_count == 0) {
    ph_set_error("link_apply: layer %u out of range", lft);
    return PHNSW_E_INVALID;
  }
  PhLayerHost &L = ix->layers[lft];
  PH_TRY(symmetric_store(ix->store));
  PH_TRY(ensure_row_dist(ix, L));
  uint32_t n = L.n_nodes, M = (uint32_t)link_count;
  DevBuf<uint32_t> tgt;
  PH_TRY(tgt.alloc((size_t)n * M));
  hipLaunchKernelGGL(ph_link_targets_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, L.nodes, n,
                     L.identity ? nullptr : L.vec2node, (uint32_t)ix->store->n, res_ids, res_len, M, tgt.p);
  PH_HIP(hipGetLastError());
  return apply_proposals(L, tgt.p, res_d, M, out_added);
}

// link_nodes_in_layer_to_better_neighbors over all nodes  lib.rs:1070-1154
static int link_layer_impl(phnsw_index *ix, uint32_t lft, const phnsw_search_params *sp, uint64_t link_count,
                           uint64_t *out_added) {
  if (lft >= ix->layers.size()) {
    ph_set_error("link_layer: layer %u out of range", lft);
    return PHNSW_E_INVALID;
  }
  uint32_t n = ix->layers[lft].n_nodes, M = (uint32_t)link_count;
  PhTimer tm("link_layer", n);
  DevBuf<uint32_t> res_ids, res_len;
  DevBuf<float> res_d;
  PH_TRY(res_ids.alloc((size_t)n * M));
  PH_TRY(res_d.alloc((size_t)n * M));
  PH_TRY(res_len.alloc(n));
  PH_TRY(link_search_impl(ix, lft, sp, link_count, 0, n, res_ids.p, res_d.p, res_len.p));
  return link_apply_impl(ix, lft, link_count, res_ids.p, res_d.p, res_len.p, out_added);
}

// position (locality schedule) of stored queries that are nodes of the layer
__global__ void ph_query_pos_kernel(const uint32_t *qvec, uint32_t nq, const uint32_t *vec2node, const uint32_t *pos,
                                    uint32_t n_nodes, uint32_t *out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  uint32_t v = qvec[i];
  uint32_t nid = vec2node ? vec2node[v] : v;
  out[i] = nid < n_nodes ? pos[nid] : PH_EMPTY32;
}

// the sample of stochastic_recall_at  lib.rs:1468-1483 (StdRng::seed_from_u64(42): the same
// sample every call, so it is kept with the layer until its node list changes)
static int recall_sample(phnsw_index *ix, uint32_t at, const phnsw_optimization_params *op, const uint32_t **out_dev,
                         uint64_t *out_selection) {
  PhLayerHost &L = ix->layers[at];
  uint64_t total = L.n_nodes;
  uint64_t selection = (uint64_t)((float)total * op->recall_proportion);
  selection = std::min<uint64_t>(std::max<uint64_t>(selection, 1), total);
  *out_selection = selection;
  if (selection == total) {
    *out_dev = L.nodes;
    return 0;
  }
  if (!L.recall_q || L.recall_n != selection) {
    std::vector<uint32_t> h_nodes(total);
    PH_HIP(hipMemcpy(h_nodes.data(), L.nodes, total * 4, hipMemcpyDeviceToHost));
    std::vector<uint64_t> vecs(h_nodes.begin(), h_nodes.end());
    ph_shuffle_u64(vecs.data(), total, 42);
    std::vector<uint32_t> q(selection);
    for (uint64_t i = 0; i < selection; i++) q[i] = (uint32_t)vecs[i];
    if (L.recall_q) hipFree(L.recall_q);
    L.recall_q = nullptr;
    PH_HIP(hipMalloc(&L.recall_q, selection * 4));
    PH_HIP(hipMemcpy(L.recall_q, q.data(), selection * 4, hipMemcpyHostToDevice));
    L.recall_n = (uint32_t)selection;
  }
  *out_dev = L.recall_q;
  return 0;
}

// hits among sample[first, first+count): self.search(Stored(vid), op.search) over the whole
// stack, "any result == vid"  lib.rs:1485-1494
static int recall_hits_impl(phnsw_index *ix, uint32_t at, const phnsw_optimization_params *op, uint64_t first,
                            uint64_t count, uint64_t *out_hits, uint64_t *out_selection) {
  if (at >= ix->layers.size()) {
    ph_set_error("stochastic_recall_at: layer %u out of range", at);
    return PHNSW_E_INVALID;
  }
  const uint32_t *q = nullptr;
  uint64_t selection = 0;
  PH_TRY(recall_sample(ix, at, op, &q, &selection));
  if (out_selection) *out_selection = selection;
  if (first > selection) first = selection;
  if (first + count > selection) count = selection - first;
  *out_hits = 0;
  if (count == 0) return 0;
  uint32_t nq = (uint32_t)count;
  DevBuf<uint32_t> ids, len, hit;
  DevBuf<float> d;
  PH_TRY(ids.alloc(nq));
  PH_TRY(d.alloc(nq));
  PH_TRY(len.alloc(nq));
  PH_TRY(hit.alloc(nq));
  DevBuf<uint32_t> okeys, order;
  const PhLayerHost &L = ix->layers[at];
  if (L.pos && nq >= PH_ORDER_MIN && !getenv("PHNSW_NO_LOCALITY")) {
    PH_TRY(okeys.alloc(nq));
    PH_TRY(order.alloc(nq));
    hipLaunchKernelGGL(ph_query_pos_kernel, dim3((nq + 255) / 256), dim3(256), 0, 0, q + first, nq,
                       L.identity ? nullptr : L.vec2node, L.pos, L.n_nodes, okeys.p);
    PH_HIP(hipGetLastError());
    PH_TRY(ph_order_by_keys_device(okeys.p, nq, order.p, 0));
  }
  // the sample is the layer's node list itself when every node is sampled, else any of its nodes
  const PhRowHint hint = row_hint(L, (uint32_t)first, nq, q == L.nodes);
  PH_TRY(search_stored(ix, q + first, nq, &op->search, 0, nullptr, ids.p, d.p, len.p, 1, hit.p, order.p, &hint));
  std::vector<uint32_t> h(nq);
  PH_HIP(hipMemcpy(h.data(), hit.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
  uint64_t relevant = 0;
  for (uint32_t x : h) relevant += x;
  *out_hits = relevant;
  return 0;
}

// stochastic_recall_at  lib.rs:1463-1499
static int recall_impl(phnsw_index *ix, uint32_t at, const phnsw_optimization_params *op, float *out) {
  if (at >= ix->layers.size()) {
    ph_set_error("stochastic_recall_at: layer %u out of range", at);
    return PHNSW_E_INVALID;
  }
  PhTimer tm("stochastic_recall_at", ix->layers[at].n_nodes);
  uint64_t hits = 0, selection = 0;
  PH_TRY(recall_hits_impl(ix, at, op, 0, UINT64_MAX / 2, &hits, &selection));
  *out = (float)hits / (float)selection;
  return 0;
}

// improve_neighbors_upto  lib.rs:1515-1544
static int improve_neighbors_upto_impl(phnsw_index *ix, uint32_t upto, const phnsw_build_params *bp, float last_recall,
                                       float *out) {
  if (upto < 1 || upto > ix->layers.size()) {
    ph_set_error("improve_neighbors_upto: upto %u out of range", upto);
    return PHNSW_E_INVALID;
  }
  const phnsw_optimization_params *op = &bp->optimization;
  float last = (last_recall != last_recall) ? 0.0f : last_recall;
  float improvement = 1.0f;
  uint64_t rounds = 0;
  while (improvement >= op->neighborhood_threshold && last < 1.0f) {
    for (uint32_t lft = 0; lft < upto; lft++) PH_TRY(link_layer_impl(ix, lft, &op->search, bp->neighborhood_size, nullptr));
    float recall = 0.f;
    PH_TRY(recall_impl(ix, upto - 1, op, &recall));
    improvement = recall - last;
    last = recall;
    rounds++;
    if (ph_verbose()) fprintf(stderr, "[phnsw] improve_neighbors_upto(%u) round %llu recall %.4f\n", upto, (unsigned long long)rounds, recall);
    if (bp->max_link_rounds && rounds >= bp->max_link_rounds) break;
  }
  *out = last;
  return 0;
}

// ------------------------------------------------------------------ promotion (lib.rs:1002-1068, 1167-1427)

static int build_impl(phnsw_store *s, const uint64_t *vids, uint64_t n, const phnsw_build_params *bp,
                      phnsw_progress_cb cb, void *user, phnsw_index **out);

// discover_unreachable_vectors  lib.rs:1002-1037: nodes of layer `lft` that a search over
// layers[0..=lft] does not return among its leading |d| < 1e-5 results
// (match_within_epsilon search.rs:173-187) and that are not in the layer above
// phase 1: self-hit flags (epsilon form) of nodes [first, first+count) of layer lft
static int discover_hits_impl(phnsw_index *ix, uint32_t lft, const phnsw_search_params *sp, uint32_t first,
                              uint32_t count, uint32_t *hit_dev) {
  if (lft >= ix->layers.size() || first + (uint64_t)count > ix->layers[lft].n_nodes) {
    ph_set_error("discover_unreachable: layer %u / range out of bounds", lft);
    return PHNSW_E_INVALID;
  }
  if (count == 0) return 0;
  PhLayerHost &L = ix->layers[lft];
  PhTimer tm("discover_unreachable", count);
  DevBuf<uint32_t> ids, len, status;
  DevBuf<float> d;
  PH_TRY(ids.alloc(count));
  PH_TRY(d.alloc(count));
  PH_TRY(len.alloc(count));
  PH_TRY(status.alloc(count));
  uint32_t ovf_cap = ph_default_ovf_cap((uint32_t)sp->number_of_candidates);
  const uint32_t *order = nullptr;
  PH_TRY(ph_layer_range_order(L, first, count, &order));
  for (int attempt = 0; attempt < 3; attempt++) {
    const PhRowHint hint = row_hint(L, first, count, true);
    PH_TRY(ph_search_device(ix, nullptr, 0, L.nodes + first, count, sp, lft + 1, nullptr, ids.p, d.p, len.p, nullptr,
                            status.p, ovf_cap, 0, 0, 1, hit_dev, 0.f, 0, 1e-5f, order, nullptr, &hint));
    PH_HIP(hipDeviceSynchronize());
    std::vector<uint32_t> hs(count);
    PH_HIP(hipMemcpy(hs.data(), status.p, (size_t)count * 4, hipMemcpyDeviceToHost));
    bool overflow = false;
    for (uint32_t x : hs) {
      if (x == 4) {
        ph_set_error("discover_unreachable: layers are not nested");
        return PHNSW_E_MISSING_NODE;
      }
      overflow |= x == 5;
    }
    if (!overflow) return 0;
    ovf_cap *= 8;
  }
  ph_set_error("discover_unreachable: frontier spill overflow");
  return PHNSW_E_OVERFLOW;
}

// phase 2: the vectors that did not find themselves and are not in the layer above
static int discover_filter_impl(phnsw_index *ix, uint32_t lft, const uint32_t *hit_dev, std::vector<uint32_t> &out) {
  PhLayerHost &L = ix->layers[lft];
  uint32_t n = L.n_nodes;
  PhTimer tm(" discover_filter", n);
  std::vector<uint32_t> h(n), nodes(n), above;
  PH_HIP(hipMemcpy(h.data(), hit_dev, (size_t)n * 4, hipMemcpyDeviceToHost));
  PH_HIP(hipMemcpy(nodes.data(), L.nodes, (size_t)n * 4, hipMemcpyDeviceToHost));
  if (lft > 0) {
    above.resize(ix->layers[lft - 1].n_nodes);
    PH_HIP(hipMemcpy(above.data(), ix->layers[lft - 1].nodes, above.size() * 4, hipMemcpyDeviceToHost));
  }
  out.clear();
  for (uint32_t i = 0; i < n; i++)
    if (!h[i] && (lft == 0 || !std::binary_search(above.begin(), above.end(), nodes[i]))) out.push_back(nodes[i]);
  return 0;
}

static int discover_unreachable_impl(phnsw_index *ix, uint32_t lft, const phnsw_search_params *sp,
                                     std::vector<uint32_t> &out) {
  if (lft >= ix->layers.size()) {
    ph_set_error("discover_unreachable: layer %u out of range", lft);
    return PHNSW_E_INVALID;
  }
  DevBuf<uint32_t> hit;
  PH_TRY(hit.alloc(ix->layers[lft].n_nodes));
  PH_TRY(discover_hits_impl(ix, lft, sp, 0, ix->layers[lft].n_nodes, hit.p));
  return discover_filter_impl(ix, lft, hit.p, out);
}

// extend_layer  lib.rs:1039-1068 (generate_node_maps :1767-1812, copy_old_neighborhoods :1737-1765,
// initialize_new_neighborhoods :1727-1735): integer renumbering on the host
static int extend_layer_impl(phnsw_index *ix, uint32_t lft, std::vector<uint32_t> vecs) {
  PhLayerHost &L = ix->layers[lft];
  const uint32_t W = L.W, n_old = L.n_nodes;
  PhTimer tm(" extend_layer", vecs.size());
  std::sort(vecs.begin(), vecs.end());
  std::vector<uint32_t> old_nodes(n_old), old_nb((size_t)n_old * W);
  std::vector<float> old_d;
  PH_HIP(hipMemcpy(old_nodes.data(), L.nodes, (size_t)n_old * 4, hipMemcpyDeviceToHost));
  PH_HIP(hipMemcpy(old_nb.data(), L.neighbors, old_nb.size() * 4, hipMemcpyDeviceToHost));
  if (L.nbr_dist) {
    old_d.resize((size_t)n_old * W);
    PH_HIP(hipMemcpy(old_d.data(), L.nbr_dist, old_d.size() * 4, hipMemcpyDeviceToHost));
  }
  const uint32_t n_new = n_old + (uint32_t)vecs.size();
  std::vector<uint32_t> nodes(n_new), old_map(n_old);
  size_t a = 0, b = 0, o = 0;
  while (a < n_old || b < vecs.size()) {
    if (b >= vecs.size() || (a < n_old && old_nodes[a] < vecs[b])) {
      old_map[a] = (uint32_t)o;
      nodes[o++] = old_nodes[a++];
    } else {
      if (a < n_old && old_nodes[a] == vecs[b]) {
        ph_set_error("extend_layer: tried to insert vector that already exists in this layer");  // lib.rs:1797
        return PHNSW_E_INVALID;
      }
      nodes[o++] = vecs[b++];
    }
  }
  std::vector<uint32_t> nb((size_t)n_new * W, PH_EMPTY32);
  std::vector<float> nd((size_t)n_new * W, PH_FMAX);
  for (uint32_t i = 0; i < n_old; i++)
    for (uint32_t k = 0; k < W; k++) {
      uint32_t x = old_nb[(size_t)i * W + k];
      nb[(size_t)old_map[i] * W + k] = x == PH_EMPTY32 ? PH_EMPTY32 : old_map[x];
      if (!old_d.empty()) nd[(size_t)old_map[i] * W + k] = old_d[(size_t)i * W + k];
    }
  PhLayerHost NL;
  PH_TRY(ph_layer_upload(ix, nodes.data(), nb.data(), n_new, W, &NL));
  if (!old_d.empty()) {
    hipError_t e = hipMalloc(&NL.nbr_dist, nd.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(NL.nbr_dist, nd.data(), nd.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      ph_layer_free(NL);
      return ph_hip_fail(e, "extend_layer upload", __FILE__, __LINE__);
    }
  }
  if (L.pos) {  // keep the locality order: old nodes keep their rank, new nodes go last
    std::vector<uint32_t> op(n_old), np(n_new, PH_EMPTY32);
    PH_HIP(hipMemcpy(op.data(), L.pos, (size_t)n_old * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n_old; i++) np[old_map[i]] = op[i];
    uint32_t next = n_old;
    for (uint32_t i = 0; i < n_new; i++)
      if (np[i] == PH_EMPTY32) np[i] = next++;
    hipError_t e = hipMalloc(&NL.pos, (size_t)n_new * 4);
    if (e == hipSuccess) e = hipMemcpy(NL.pos, np.data(), (size_t)n_new * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      ph_layer_free(NL);
      return ph_hip_fail(e, "extend_layer positions", __FILE__, __LINE__);
    }
  }
  ph_layer_free(L);
  ix->layers[lft] = NL;
  return 0;
}

__global__ void ph_gather_rows_kernel(const uint32_t *neighbors, uint32_t W, const uint32_t *which, uint32_t cnt,
                                      uint32_t *out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)cnt * W) return;
  out[i] = neighbors[(size_t)which[i / W] * W + (i % W)];
}

// filter_promotion_candidates  lib.rs:1176-1271.  Every unreachable vector of layer `lft`
// is absent from the layer above, so (nesting invariant) its discover_order_from_top is
// `lft`: one histogram.  Ties of the count sort (HashMap order in the reference) are
// broken by node id.
static int filter_promotion_candidates_impl(phnsw_index *ix, uint32_t lft, const std::vector<uint32_t> &vecs,
                                            const phnsw_search_params *sp, std::vector<uint32_t> &sel) {
  sel.clear();
  if (lft == 0) return 0;
  const phnsw_store *s = ix->store;
  PhLayerHost &L = ix->layers[lft];
  const uint32_t n = L.n_nodes, W = L.W, nv = (uint32_t)vecs.size();
  PhTimer tm(" filter_promotion_candidates", vecs.size());
  // only the rows of the unreachable nodes are needed on the host
  std::vector<uint32_t> nodes(n), vnode(nv), nb((size_t)nv * W);
  PH_HIP(hipMemcpy(nodes.data(), L.nodes, (size_t)n * 4, hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < nv; i++)
    vnode[i] = (uint32_t)(std::lower_bound(nodes.begin(), nodes.end(), vecs[i]) - nodes.begin());
  {
    DevBuf<uint32_t> which, rows;
    PH_TRY(which.alloc(nv));
    PH_TRY(rows.alloc((size_t)nv * W));
    PH_HIP(hipMemcpy(which.p, vnode.data(), (size_t)nv * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ph_gather_rows_kernel, dim3((uint32_t)(((uint64_t)nv * W + 255) / 256)), dim3(256), 0, 0,
                       L.neighbors, W, which.p, nv, rows.p);
    PH_HIP(hipGetLastError());
    PH_HIP(hipMemcpy(nb.data(), rows.p, nb.size() * 4, hipMemcpyDeviceToHost));
  }
  std::vector<uint32_t> hit_nodes;
  for (uint32_t i = 0; i < nv; i++)  // histogramming  :1190-1223
    for (uint32_t k = 0; k < W; k++) {
      uint32_t x = nb[(size_t)i * W + k];
      if (x == PH_EMPTY32) break;
      if (std::binary_search(vecs.begin(), vecs.end(), nodes[x])) hit_nodes.push_back(x);
    }
  std::sort(hit_nodes.begin(), hit_nodes.end());
  std::vector<std::pair<uint32_t, uint32_t>> histo;  // (count, node), ascending; popped from the back
  for (size_t i = 0; i < hit_nodes.size();) {
    size_t j = i;
    while (j < hit_nodes.size() && hit_nodes[j] == hit_nodes[i]) j++;
    histo.push_back({(uint32_t)(j - i), hit_nodes[i]});
    i = j;
  }
  std::sort(histo.begin(), histo.end());
  const uint32_t H = (uint32_t)histo.size();
  if (H == 0) return 0;
  std::vector<uint32_t> cand(H);  // in the order the reference pops them  :1243
  for (uint32_t j = 0; j < H; j++) cand[j] = nodes[histo[H - 1 - j].second];
  DevBuf<uint32_t> cand_d;
  PH_TRY(cand_d.alloc(H));
  PH_HIP(hipMemcpy(cand_d.p, cand.data(), (size_t)H * 4, hipMemcpyHostToDevice));
  // The reference thins the candidates one at a time (:1243-1267): a candidate is kept unless an earlier kept one k
  // has compare_vec(candidate, k) < radius_k, radius_k = search_upto(Stored(k), sp, lft)[0].1.  The index does not
  // change meanwhile, so radii and distances can be computed ahead in blocks of PH_THIN_BLOCK candidates -- one
  // batched search for the block's radii, one kernel for "covered by a vector kept in an earlier block", one for
  // the block x block table -- and only the (integer, order-dependent) selection inside a block runs on the host.
  // Same comparisons on the same values as the sequential form, hence the same selection (one launch + two
  // synchronisations per CANDIDATE before: 4.4 s of a 10M build on every rank).
  const uint32_t B = PH_THIN_BLOCK;
  std::vector<float> radius;
  DevBuf<uint32_t> sel_d, r_id, r_len, cov_d;
  DevBuf<float> rad_d, r_d, table;
  PH_TRY(sel_d.alloc(H));
  PH_TRY(rad_d.alloc(H));
  PH_TRY(r_id.alloc(std::min(B, H)));
  PH_TRY(r_d.alloc(std::min(B, H)));
  PH_TRY(r_len.alloc(std::min(B, H)));
  PH_TRY(cov_d.alloc(std::min(B, H)));
  PH_TRY(table.alloc((size_t)std::min(B, H) * std::min(B, H)));
  std::vector<float> hr, ht;
  std::vector<uint32_t> hl, hc;
  for (uint32_t c0 = 0; c0 < H; c0 += B) {
    const uint32_t bc = std::min(B, H - c0);
    const size_t prev = sel.size();
    PH_TRY(search_stored(ix, cand_d.p + c0, bc, sp, lft, nullptr, r_id.p, r_d.p, r_len.p, 1, nullptr));
    dim3 g(s->codes ? pq_grid(s, bc) : wave_grid(bc));
    if (prev)
      PH_TRY(launch_by_policy(s, g, ph_cover_kernel<DistF32<1>>, ph_cover_kernel<DistF32<3>>, ph_cover_kernel<DistF32<6>>,
                              ph_cover_kernel<DistPQ>, ph_dist_args(s), cand_d.p + c0, bc, (const uint32_t *)sel_d.p,
                              (uint32_t)prev, (const float *)rad_d.p, cov_d.p, (float *)nullptr));
    PH_TRY(launch_by_policy(s, g, ph_cover_kernel<DistF32<1>>, ph_cover_kernel<DistF32<3>>, ph_cover_kernel<DistF32<6>>,
                            ph_cover_kernel<DistPQ>, ph_dist_args(s), cand_d.p + c0, bc, (const uint32_t *)(cand_d.p + c0), bc,
                            (const float *)nullptr, (uint32_t *)nullptr, table.p));
    hr.resize(bc);
    hl.resize(bc);
    hc.assign(bc, 0);
    ht.resize((size_t)bc * bc);
    PH_HIP(hipMemcpy(hr.data(), r_d.p, (size_t)bc * 4, hipMemcpyDeviceToHost));
    PH_HIP(hipMemcpy(hl.data(), r_len.p, (size_t)bc * 4, hipMemcpyDeviceToHost));
    if (prev) PH_HIP(hipMemcpy(hc.data(), cov_d.p, (size_t)bc * 4, hipMemcpyDeviceToHost));
    PH_HIP(hipMemcpy(ht.data(), table.p, ht.size() * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> kept;  // block-local indices kept so far
    for (uint32_t j = 0; j < bc; j++) {
      bool covered = hc[j] != 0;
      for (size_t k = 0; k < kept.size() && !covered; k++) covered = ht[(size_t)j * bc + kept[k]] < radius[prev + k];
      if (covered) continue;
      kept.push_back(j);
      sel.push_back(cand[c0 + j]);
      radius.push_back(hl[j] ? hr[j] : 0.f);
    }
    if (sel.size() > prev) {
      PH_HIP(hipMemcpy(sel_d.p + prev, sel.data() + prev, (sel.size() - prev) * 4, hipMemcpyHostToDevice));
      PH_HIP(hipMemcpy(rad_d.p + prev, radius.data() + prev, (sel.size() - prev) * 4, hipMemcpyHostToDevice));
    }
  }
  return 0;
}

// promote_at_layer  lib.rs:1273-1427
static int promote_at_layer_impl(phnsw_index *ix, uint32_t lft, const phnsw_build_params *bp, int *promoted,
                                 const uint32_t *hit_dev = nullptr) {
  *promoted = 0;
  std::vector<uint32_t> vecs;
  if (hit_dev) {
    if (lft >= ix->layers.size()) return PHNSW_E_INVALID;
    PH_TRY(discover_filter_impl(ix, lft, hit_dev, vecs));
  } else
    PH_TRY(discover_unreachable_impl(ix, lft, &bp->optimization.search, vecs));
  if (vecs.empty()) return 0;
  const float max_proportion = bp->optimization.promotion_proportion;
  if (max_proportion < 1.0f) {  // :1288-1294
    vecs.resize((size_t)((float)vecs.size() * max_proportion));
    if (vecs.empty()) return 0;
  }
  std::vector<uint32_t> sel;
  PH_TRY(filter_promotion_candidates_impl(ix, lft, vecs, &bp->optimization.search, sel));
  *promoted = 1;
  if (lft == 0 || sel.empty()) return 0;  // nothing to extend; the reference still returns true
  // the else branch of :1332-1420 (layer_from_top = lft >= 1)
  const uint32_t nsz = lft;
  std::vector<uint64_t> sizes(nsz);
  for (uint32_t i = 0; i < nsz; i++) sizes[i] = ix->layers[lft - 1 - i].n_nodes;  // reversed: [0] = just above
  std::vector<uint64_t> new_sizes = calculate_partitions(sizes[0] + sel.size(), bp->order);
  std::reverse(new_sizes.begin(), new_sizes.end());  // from the bottom
  if (new_sizes.size() < nsz) new_sizes.resize(nsz, 0);  // :1345-1349
  const uint32_t retop_upto = (uint32_t)new_sizes.size() - nsz;
  std::vector<uint64_t> promo(nsz);
  for (uint32_t i = 0; i < nsz; i++) promo[i] = new_sizes[i] > sizes[i] ? new_sizes[i] - sizes[i] : 0;
  uint32_t offset = 0;
  if (retop_upto != 0) {  // :1361-1397
    const uint32_t retop_index = nsz - retop_upto;
    uint64_t into_top = std::min<uint64_t>(promo[retop_index], sel.size());
    promo.resize(retop_index);
    const PhLayerHost &T = ix->layers[retop_upto - 1];
    std::vector<uint32_t> tn(T.n_nodes);
    PH_HIP(hipMemcpy(tn.data(), T.nodes, (size_t)T.n_nodes * 4, hipMemcpyDeviceToHost));
    std::vector<uint64_t> top(tn.begin(), tn.end());
    for (uint64_t k = 0; k < into_top; k++) top.push_back(sel[k]);
    std::sort(top.begin(), top.end());
    top.erase(std::unique(top.begin(), top.end()), top.end());
    phnsw_build_params nbp = *bp;
    nbp.zero_layer_neighborhood_size = bp->neighborhood_size;  // "a fake zero layer"  :1377-1379
    nbp.seed = bp->seed + 0x51ED270B9F3ULL + ix->layers.size();  // thread_rng in the reference
    phnsw_index *nt = nullptr;
    PhTimer tr(" promote: rebuild top", top.size());
    PH_TRY(build_impl(ix->store, top.data(), top.size(), &nbp, nullptr, nullptr, &nt));
    std::vector<PhLayerHost> nl = nt->layers;
    nt->layers.clear();
    const uint32_t new_top_len = (uint32_t)nl.size();
    for (uint32_t i = 0; i < retop_upto; i++) ph_layer_free(ix->layers[i]);
    for (size_t i = retop_upto; i < ix->layers.size(); i++) nl.push_back(ix->layers[i]);
    ix->layers = nl;
    ix->nodes_epoch++;
    phnsw_index_destroy(nt);
    offset = new_top_len;
  }
  // promotion_sizes.reverse(); extend each remaining layer above  :1398-1412
  for (uint32_t i = 0; i < promo.size(); i++) {
    uint64_t size = promo[promo.size() - 1 - i];
    uint32_t cur = offset + i;
    const PhLayerHost &L = ix->layers[cur];
    std::vector<uint32_t> ln(L.n_nodes), tp;
    PH_HIP(hipMemcpy(ln.data(), L.nodes, (size_t)L.n_nodes * 4, hipMemcpyDeviceToHost));
    for (size_t k = 0; k < sel.size() && tp.size() < size; k++)
      if (!std::binary_search(ln.begin(), ln.end(), sel[k])) tp.push_back(sel[k]);
    PH_TRY(extend_layer_impl(ix, cur, tp));
  }
  return 0;
}

// improve_index_at  lib.rs:1546-1603; *lft_io may grow when promotion adds layers
static int improve_index_at_impl(phnsw_index *ix, uint32_t *lft_io, const phnsw_build_params *bp, float *out) {
  const phnsw_optimization_params *op = &bp->optimization;
  uint32_t lft = *lft_io;
  float recall = 0.f;
  PH_TRY(recall_impl(ix, lft, op, &recall));
  float improvement = 1.0f;
  int bailout = 1;
  while (improvement >= op->promotion_threshold && recall < 1.0f && bailout != 0) {
    float last = recall;
    uint32_t cur = 0;
    while (cur <= lft && bailout != 0) {
      size_t layer_count = ix->layers.size();
      PH_TRY(improve_neighbors_upto_impl(ix, cur + 1, bp, NAN, &recall));
      if (recall == 1.0f) {  // :1569-1572
        cur++;
        continue;
      }
      if (bp->promote) {
        int promoted = 0;
        PH_TRY(promote_at_layer_impl(ix, cur, bp, &promoted));  // :1575
        if (promoted) {
          uint32_t delta = (uint32_t)(ix->layers.size() - layer_count);
          cur += delta;
          lft += delta;
          PH_TRY(improve_neighbors_upto_impl(ix, cur + 1, bp, recall, &recall));  // :1586-1587
        }
      }
      cur++;
    }
    bailout--;
    improvement = recall - last;
  }
  *lft_io = lft;
  *out = recall;
  return 0;
}

static int improve_index_impl(phnsw_index *ix, const phnsw_build_params *bp, float last_recall, phnsw_progress_cb cb,
                              void *user, float *out) {
  if (ix->layers.empty()) {
    ph_set_error("improve_index: index has no layers");
    return PHNSW_E_INVALID;
  }
  // last_recall.unwrap_or_else(|| self.stochastic_recall(bp.optimization))  lib.rs:1671
  float recall = last_recall;
  if (last_recall != last_recall) PH_TRY(recall_impl(ix, (uint32_t)ix->layers.size() - 1, &bp->optimization, &recall));
  uint32_t lft = 0;
  while (lft < ix->layers.size()) {  // lib.rs:1673-1683
    PH_TRY(improve_index_at_impl(ix, &lft, bp, &recall));
    lft++;
    if (cb && cb(user, "improve_index", lft, ix->layers.size())) {
      ph_set_error("interrupted by the progress callback");
      return PHNSW_E_INVALID;
    }
  }
  if (out) *out = recall;
  return 0;
}

// ------------------------------------------------------------------ C ABI

// the table kept across the rounds of a build lives for the duration of the call that builds (tiny.hip)
struct BuildTableScope {
  phnsw_index *ix;
  explicit BuildTableScope(phnsw_index *i) : ix(i) { ix->bt_enabled = true; }
  ~BuildTableScope() {
    ix->bt_enabled = false;
    ph_build_table_free(ix);
  }
};

static int enter(const phnsw_index *ix) {
  if (!ix) {
    ph_set_error("null index");
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(ix->store->device));
  return 0;
}

extern "C" int phnsw_generate_layer(phnsw_index *ix, const uint64_t *vids, uint64_t n, uint64_t neighborhood_size,
                                    const phnsw_build_params *bp) try {
  PH_TRY(enter(ix));
  return generate_layer_impl(ix, vids, n, neighborhood_size, bp);
} catch (...) { return ph_caught(); }

extern "C" int phnsw_link_layer(phnsw_index *ix, uint32_t layer_from_top, const phnsw_search_params *sp,
                                uint64_t link_count, uint64_t *out_added) try {
  PH_TRY(enter(ix));
  if (!sp) return PHNSW_E_INVALID;
  return link_layer_impl(ix, layer_from_top, sp, link_count, out_added);
} catch (...) { return ph_caught(); }

extern "C" int phnsw_stochastic_recall_at(phnsw_index *ix, uint32_t layer_from_top,
                                          const phnsw_optimization_params *op, float *out_recall) try {
  PH_TRY(enter(ix));
  if (!op || !out_recall) return PHNSW_E_INVALID;
  return recall_impl(ix, layer_from_top, op, out_recall);
} catch (...) { return ph_caught(); }

extern "C" int phnsw_improve_neighbors_upto(phnsw_index *ix, uint32_t upto, const phnsw_build_params *bp,
                                            float last_recall, float *out_recall) try {
  PH_TRY(enter(ix));
  if (!bp || !out_recall) return PHNSW_E_INVALID;
  BuildTableScope scope(ix);
  return improve_neighbors_upto_impl(ix, upto, bp, last_recall, out_recall);
} catch (...) { return ph_caught(); }

extern "C" int phnsw_improve_index(phnsw_index *ix, const phnsw_build_params *bp, float last_recall,
                                   phnsw_progress_cb cb, void *user, float *out_recall) try {
  PH_TRY(enter(ix));
  if (!bp) return PHNSW_E_INVALID;
  BuildTableScope scope(ix);
  return improve_index_impl(ix, bp, last_recall, cb, user, out_recall);
} catch (...) { return ph_caught(); }

// ---- phase entry points for multi-GPU drivers (device buffers, u32 ids) ----
extern "C" int phnsw_layer_begin(phnsw_index *ix, const uint64_t *vids, uint64_t n, uint64_t neighborhood_size,
                                 const phnsw_build_params *bp, int *needs_phases) try {
  PH_TRY(enter(ix));
  if (!needs_phases) return PHNSW_E_INVALID;
  return layer_begin_impl(ix, vids, n, neighborhood_size, bp, needs_phases);
} catch (...) { return ph_caught(); }
// the sharded form of layer_begin: the cells of the new layer (the anchor GEMM, the one costly step) are left to
// phnsw_layer_cells_device (a node range per rank) + phnsw_layer_set_cells_device (the all-gathered array)
extern "C" int phnsw_layer_begin_sharded(phnsw_index *ix, const uint64_t *vids, uint64_t n, uint64_t neighborhood_size,
                                         const phnsw_build_params *bp, int *needs_phases, int *needs_cells) try {
  PH_TRY(enter(ix));
  if (!needs_phases || !needs_cells) return PHNSW_E_INVALID;
  return layer_begin_impl(ix, vids, n, neighborhood_size, bp, needs_phases, needs_cells);
} catch (...) { return ph_caught(); }
extern "C" int phnsw_layer_cells_device(phnsw_index *ix, uint64_t first, uint64_t count, uint32_t *out_pos) try {
  PH_TRY(enter(ix));
  PhPendingLayer *P = ix->pending;
  if (!P || !out_pos || first + count > P->L.n_nodes) {
    ph_set_error("layer_cells: no pending layer or range out of bounds");
    return PHNSW_E_INVALID;
  }
  return ph_layer_cells_range(ix->store, P->L, (uint32_t)first, (uint32_t)count, out_pos);
} catch (...) { return ph_caught(); }
extern "C" int phnsw_layer_set_cells_device(phnsw_index *ix, const uint32_t *pos) try {
  PH_TRY(enter(ix));
  PhPendingLayer *P = ix->pending;
  if (!P || !pos) {
    ph_set_error("layer_set_cells: no pending layer");
    return PHNSW_E_INVALID;
  }
  if (!P->L.pos) PH_HIP(hipMalloc(&P->L.pos, (size_t)P->L.n_nodes * 4));
  PH_HIP(hipMemcpy(P->L.pos, pos, (size_t)P->L.n_nodes * 4, hipMemcpyDeviceToDevice));
  return 0;
} catch (...) { return ph_caught(); }
extern "C" int phnsw_layer_init_search_device(phnsw_index *ix, const phnsw_build_params *bp, uint64_t first,
                                              uint64_t count, uint32_t *out_ids, float *out_d, uint32_t *out_len) try {
  PH_TRY(enter(ix));
  if (!bp) return PHNSW_E_INVALID;
  return layer_init_search_impl(ix, bp, (uint32_t)first, (uint32_t)count, out_ids, out_d, out_len);
} catch (...) { return ph_caught(); }
extern "C" int phnsw_layer_seed_device(phnsw_index *ix, const phnsw_build_params *bp, const uint32_t *init_ids,
                                       const float *init_d, const uint32_t *init_len, uint64_t first, uint64_t count,
                                       uint32_t *out_rows, float *out_rows_d) try {
  PH_TRY(enter(ix));
  if (!bp) return PHNSW_E_INVALID;
  return layer_seed_impl(ix, bp, init_ids, init_d, init_len, (uint32_t)first, (uint32_t)count, out_rows, out_rows_d);
} catch (...) { return ph_caught(); }
extern "C" int phnsw_layer_finish_device(phnsw_index *ix, const uint32_t *rows, const float *rows_d) try {
  PH_TRY(enter(ix));
  return layer_finish_impl(ix, rows, rows_d);
} catch (...) { return ph_caught(); }
extern "C" int phnsw_link_search_device(phnsw_index *ix, uint32_t layer_from_top, const phnsw_search_params *sp,
                                        uint64_t link_count, uint64_t first, uint64_t count, uint32_t *out_ids,
                                        float *out_d, uint32_t *out_len) try {
  PH_TRY(enter(ix));
  if (!sp) return PHNSW_E_INVALID;
  return link_search_impl(ix, layer_from_top, sp, link_count, (uint32_t)first, (uint32_t)count, out_ids, out_d, out_len);
} catch (...) { return ph_caught(); }
extern "C" int phnsw_link_apply_device(phnsw_index *ix, uint32_t layer_from_top, uint64_t link_count,
                                       const uint32_t *ids, const float *d, const uint32_t *len, uint64_t *out_added) try {
  PH_TRY(enter(ix));
  return link_apply_impl(ix, layer_from_top, link_count, ids, d, len, out_added);
} catch (...) { return ph_caught(); }
extern "C" int phnsw_recall_hits(phnsw_index *ix, uint32_t layer_from_top, const phnsw_optimization_params *op,
                                 uint64_t first, uint64_t count, uint64_t *out_hits, uint64_t *out_selection) try {
  PH_TRY(enter(ix));
  if (!op || !out_hits) return PHNSW_E_INVALID;
  return recall_hits_impl(ix, layer_from_top, op, first, count, out_hits, out_selection);
} catch (...) { return ph_caught(); }
extern "C" int phnsw_index_create(phnsw_store *s, const phnsw_build_params *bp, phnsw_index **out) try {
  if (!s || !out) return PHNSW_E_INVALID;
  PH_HIP(hipSetDevice(s->device));
  phnsw_index *ix = new phnsw_index();
  ix->store = s;
  s->refcount++;
  if (bp)
    ix->bp = *bp;
  else
    phnsw_default_build_params(&ix->bp);
  *out = ix;
  return 0;
} catch (...) { return ph_caught(); }
// the deterministic id shuffle of phnsw_build (lib.rs:832-833) and its layer sizes
// (calculate_partitions lib.rs:1883-1899) for drivers that run the phases themselves
extern "C" int phnsw_build_plan(const uint64_t *vids, uint64_t n, const phnsw_build_params *bp, uint64_t *shuffled,
                                uint64_t *layer_sizes, uint32_t max_layers, uint32_t *layer_count) try {
  if (!vids || !bp || !shuffled || !layer_sizes || !layer_count || n == 0 || bp->order < 2) {
    ph_set_error("phnsw_build_plan: invalid argument");
    return PHNSW_E_INVALID;
  }
  memcpy(shuffled, vids, n * 8);
  ph_shuffle_u64(shuffled, n, bp->seed);
  std::vector<uint64_t> parts = calculate_partitions(n, bp->order);
  if (parts.size() > max_layers) return PHNSW_E_INVALID;
  for (size_t i = 0; i < parts.size(); i++) layer_sizes[i] = parts[i];
  *layer_count = (uint32_t)parts.size();
  return 0;
} catch (...) { return ph_caught(); }

// Hnsw::generate  lib.rs:825-893
static int build_impl(phnsw_store *s, const uint64_t *vids, uint64_t n, const phnsw_build_params *bp,
                      phnsw_progress_cb cb, void *user, phnsw_index **out) {
  if (!s || !vids || !bp || !out || n == 0 || bp->order < 2) {
    ph_set_error("phnsw_build: invalid argument (need n > 0, order >= 2)");  // assert!(total_size > 0) lib.rs:837
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(s->device));
  phnsw_index *ix = new phnsw_index();
  ix->store = s;
  s->refcount++;
  ix->bp = *bp;
  ix->bt_enabled = true;  // until the build is done (below): the dense table is kept across the rounds of a layer
  std::vector<uint64_t> vs(vids, vids + n);
  ph_shuffle_u64(vs.data(), n, bp->seed);  // vs.shuffle(&mut thread_rng())  lib.rs:832-833
  std::vector<uint64_t> parts = calculate_partitions(n, bp->order);
  size_t i = 0;
  while (i != parts.size()) {  // lib.rs:854-890
    uint64_t length = std::min<uint64_t>(parts[i], n);  // lib.rs:858-860
    size_t level = parts.size() - i - 1;
    uint64_t W = level == 0 ? bp->zero_layer_neighborhood_size : bp->neighborhood_size;
    int rc = generate_layer_impl(ix, vs.data(), length, W, bp);
    size_t old_count = ix->layers.size();
    if (!rc) rc = improve_index_impl(ix, bp, NAN, nullptr, nullptr, nullptr);  // improve_index(bp, None, progress)  lib.rs:877
    if (!rc && cb && cb(user, "generate", i + 1, parts.size())) {
      ph_set_error("interrupted by the progress callback");
      rc = PHNSW_E_INVALID;
    }
    if (rc) {
      phnsw_index_destroy(ix);
      return rc;
    }
    size_t delta = ix->layers.size() - old_count;
    if (delta > 0) {  // promotion added layers: fix the partitions  lib.rs:880-887
      std::vector<uint64_t> np;
      for (auto &l : ix->layers) np.push_back(l.n_nodes);
      np.insert(np.end(), parts.begin() + i + 1, parts.end());
      parts = np;
      i += delta;
    }
    i++;
  }
  ix->bt_enabled = false;
  ph_build_table_free(ix);
  *out = ix;
  return 0;
}

extern "C" int phnsw_build(phnsw_store *s, const uint64_t *vids, uint64_t n, const phnsw_build_params *bp,
                           phnsw_progress_cb cb, void *user, phnsw_index **out) try {
  int rc = build_impl(s, vids, n, bp, cb, user, out);
  ph_pool_trim();  // scratch of the rounds goes back to the driver
  return rc;
} catch (...) { return ph_caught(); }

// Hnsw::extend_layer  src/lib.rs:1039-1068: the vectors join the layer with empty rows, old rows
// are renumbered; a vector that is already in the layer is an error (the reference panics, :1797)
extern "C" int phnsw_extend_layer(phnsw_index *ix, uint32_t layer_from_top, const uint64_t *vids, uint64_t n) try {
  PH_TRY(enter(ix));
  if (layer_from_top >= ix->layers.size() || (!vids && n)) {
    ph_set_error("extend_layer: layer %u out of range", layer_from_top);
    return PHNSW_E_INVALID;
  }
  std::vector<uint32_t> v(n);
  for (uint64_t i = 0; i < n; i++) {
    if (vids[i] >= ix->store->n) {
      ph_set_error("extend_layer: VectorId %llu outside the store", (unsigned long long)vids[i]);
      return PHNSW_E_INVALID;
    }
    v[i] = (uint32_t)vids[i];
  }
  std::sort(v.begin(), v.end());
  if (std::adjacent_find(v.begin(), v.end()) != v.end()) {
    ph_set_error("extend_layer: duplicate VectorId in the new vectors");
    return PHNSW_E_INVALID;
  }
  return extend_layer_impl(ix, layer_from_top, v);
} catch (...) { return ph_caught(); }

extern "C" int phnsw_promote_at_layer(phnsw_index *ix, uint32_t layer_from_top, const phnsw_build_params *bp,
                                      int *out_promoted) try {
  PH_TRY(enter(ix));
  if (!bp || !out_promoted) return PHNSW_E_INVALID;
  return promote_at_layer_impl(ix, layer_from_top, bp, out_promoted);
} catch (...) { return ph_caught(); }

// phase API: the searches of promote_at_layer for a node range, then the promotion itself from
// the all-gathered flags
extern "C" int phnsw_discover_hits_device(phnsw_index *ix, uint32_t layer_from_top, const phnsw_search_params *sp,
                                          uint64_t first, uint64_t count, uint32_t *out_hit) try {
  PH_TRY(enter(ix));
  if (!sp || !out_hit) return PHNSW_E_INVALID;
  return discover_hits_impl(ix, layer_from_top, sp, (uint32_t)first, (uint32_t)count, out_hit);
} catch (...) { return ph_caught(); }
extern "C" int phnsw_promote_at_layer_hits_device(phnsw_index *ix, uint32_t layer_from_top,
                                                  const phnsw_build_params *bp, const uint32_t *hit,
                                                  int *out_promoted) try {
  PH_TRY(enter(ix));
  if (!bp || !hit || !out_promoted) return PHNSW_E_INVALID;
  return promote_at_layer_impl(ix, layer_from_top, bp, out_promoted, hit);
} catch (...) { return ph_caught(); }

extern "C" int phnsw_discover_unreachable(phnsw_index *ix, uint32_t layer_from_top, const phnsw_search_params *sp,
                                          uint64_t *out_vecs, uint64_t *out_count) try {
  PH_TRY(enter(ix));
  if (!sp || !out_vecs || !out_count) return PHNSW_E_INVALID;
  std::vector<uint32_t> v;
  PH_TRY(discover_unreachable_impl(ix, layer_from_top, sp, v));
  for (size_t i = 0; i < v.size(); i++) out_vecs[i] = v[i];
  *out_count = v.size();
  return 0;
} catch (...) { return ph_caught(); }
