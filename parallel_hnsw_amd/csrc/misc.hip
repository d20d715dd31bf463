// Small gfx950 kernels around the search: K1 distance batch (Comparator::compare_vec
// batched, /root/reference/src/lib.rs:69-73), synthetic data (bigvec.rs:59-65
// distribution), id-map scatter, NaN scan.
#define PH_RAW_ALLOC  // this file holds the accounted wrappers themselves
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>

#include "phnsw_device.h"

// ---- K1: out[i] = compare_vec(query, Stored(ids[i])); a wave takes 64 candidates at a time ----
template <class Dist>
__global__ __launch_bounds__(64) void ph_distance_batch_kernel(PhDistArgs da, const float *__restrict__ query,
                                                               uint32_t query_id, const uint32_t *__restrict__ ids,
                                                               uint32_t k, uint64_t n_store, float *__restrict__ out) {
  extern __shared__ float dist_lds[];
  const uint32_t lane = threadIdx.x;
  Dist dist;
  if (query)
    dist.prepare_raw(da, query, dist_lds, lane);
  else
    dist.prepare_stored(da, query_id, dist_lds, lane);
  for (uint32_t base = blockIdx.x * 64u; base < k; base += gridDim.x * 64u) {
    uint32_t i = base + lane;
    uint32_t id = i < k ? ids[i] : PH_EMPTY32;
    bool ok = id < n_store;
    float d = dist.batch(da, __ballot(ok), ok ? id : 0u, lane);
    if (i < k) out[i] = ok ? d : PH_FMAX;
  }
}

int ph_distance_batch(const phnsw_store *st, const float *q_dev, uint32_t query_id, const uint32_t *ids_dev, uint32_t k,
                      float *out_dev, hipStream_t s) {
  if (k == 0) return 0;
  if (st->codes16) {
    ph_set_error("distance batches over a shared-codebook PQ store are not supported; use its reconstruction store");
    return PHNSW_E_UNSUPPORTED;
  }
  PhDistArgs da = ph_dist_args(st);
  if (st->codes) {
    size_t lds = ph_pq_lds_bytes(st);
    if (lds > 48 * 1024)
      PH_HIP(hipFuncSetAttribute((const void *)ph_distance_batch_kernel<DistPQ>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    uint32_t blocks = std::min<uint32_t>((k + 63) / 64, 256);
    hipLaunchKernelGGL(ph_distance_batch_kernel<DistPQ>, dim3(blocks), dim3(64), lds, s, da, q_dev, query_id, ids_dev,
                       k, st->n, out_dev);
  } else {
    uint32_t nv4 = st->ld / 4;
    uint32_t blocks = std::min<uint32_t>((k + 63) / 64, 4096);
#define PH_LAUNCH(NV)                                                                                             \
  hipLaunchKernelGGL(ph_distance_batch_kernel<DistF32<NV>>, dim3(blocks), dim3(64), 0, s, da, q_dev, query_id,   \
                     ids_dev, k, st->n, out_dev)
    if (nv4 <= 64)
      PH_LAUNCH(1);
    else if (nv4 <= 192)
      PH_LAUNCH(3);
    else if (nv4 <= 384)
      PH_LAUNCH(6);
    else {
      ph_set_error("dim %u unsupported (max 1536)", st->dim);
      return PHNSW_E_UNSUPPORTED;
    }
#undef PH_LAUNCH
  }
  PH_HIP(hipGetLastError());
  return 0;
}

// ---- synthetic rows: one thread per vector, sequential f32 norm like bigvec.rs:59-65 ----
__device__ __forceinline__ float ph_synth_component(uint64_t key, uint32_t j) {
  uint64_t x = ph_mix64(key * 0x9E3779B97F4A7C15ULL + ((uint64_t)j + 1) * 0xD1B54A32D192ED03ULL);
  uint32_t m = (uint32_t)(x >> 40);
  return __fsub_rn(__fmul_rn((float)m, 1.0f / 8388608.0f), 1.0f);
}

__global__ void ph_synth_rows_kernel(float *rows, uint64_t first, uint64_t count, uint32_t dim, uint32_t ld,
                                     uint64_t seed, int normalize) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= count) return;
  float *row = rows + r * (uint64_t)ld;
  uint64_t key = seed + first + r;
  float ss = 0.0f;
  for (uint32_t j = 0; j < dim; j++) {
    float f = ph_synth_component(key, j);
    ss = __fadd_rn(ss, __fmul_rn(f, f));  // no contraction: matches the sequential host sum
  }
  float norm = normalize ? sqrtf(ss) : 1.0f;
  for (uint32_t j = 0; j < dim; j++) {
    float f = ph_synth_component(key, j);
    row[j] = normalize ? __fdiv_rn(f, norm) : f;
  }
  for (uint32_t j = dim; j < ld; j++) row[j] = 0.0f;
}

int ph_synth_rows(float *rows_dev, uint64_t first, uint64_t count, uint32_t dim, uint32_t ld, uint64_t seed,
                  int normalize, hipStream_t s) {
  if (count == 0) return 0;
  uint32_t blocks = (uint32_t)((count + 63) / 64);
  hipLaunchKernelGGL(ph_synth_rows_kernel, dim3(blocks), dim3(64), 0, s, rows_dev, first, count, dim, ld, seed,
                     normalize);
  PH_HIP(hipGetLastError());
  return 0;
}

// clustered variant (SURVEY section 8d second dataset; definition shared with the oracle)
__global__ void ph_synth_clustered_kernel(float *rows, const float *cent, uint64_t first, uint64_t count, uint32_t dim,
                                          uint32_t ld, uint64_t seed, uint32_t n_clusters, float a) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= count) return;
  float *row = rows + r * (uint64_t)ld;
  uint64_t key = seed + first + r;
  uint64_t k = ph_mulhi64(ph_mix64(key * 0xA24BAED4963EE407ULL + 0x9FB21C651E98DF25ULL), n_clusters);
  const float *c = cent + k * (uint64_t)ld;
  float ss = 0.0f;
  for (uint32_t j = 0; j < dim; j++) {
    float x = __fadd_rn(c[j], __fmul_rn(a, ph_synth_component(key, j)));
    ss = __fadd_rn(ss, __fmul_rn(x, x));
  }
  float norm = sqrtf(ss);
  for (uint32_t j = 0; j < dim; j++) {
    float x = __fadd_rn(c[j], __fmul_rn(a, ph_synth_component(key, j)));
    row[j] = __fdiv_rn(x, norm);
  }
  for (uint32_t j = dim; j < ld; j++) row[j] = 0.0f;
}

int ph_synth_clustered_rows(float *rows_dev, uint64_t first, uint64_t count, uint32_t dim, uint32_t ld, uint64_t seed,
                            uint32_t n_clusters, float noise, hipStream_t s) {
  if (count == 0) return 0;
  float *cent = nullptr;
  PH_HIP(hipMalloc(&cent, (size_t)n_clusters * ld * 4));
  int rc = ph_synth_rows(cent, 0, n_clusters, dim, ld, seed ^ 0xC1A55E5EEDULL, 1, s);
  if (!rc) {
    float a = noise * sqrtf(3.0f / (float)dim);
    uint32_t blocks = (uint32_t)((count + 63) / 64);
    hipLaunchKernelGGL(ph_synth_clustered_kernel, dim3(blocks), dim3(64), 0, s, rows_dev, cent, first, count, dim, ld,
                       seed, n_clusters, a);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) rc = ph_hip_fail(e, "clustered synth", __FILE__, __LINE__);
  }
  hipFree(cent);
  return rc;
}

__global__ void ph_fill_u32_kernel(uint32_t *p, uint32_t v, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    p[i] = v;
}
int ph_fill_u32(uint32_t *p, uint32_t v, uint64_t n, hipStream_t s) {
  if (n == 0) return 0;
  uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(ph_fill_u32_kernel, dim3(blocks), dim3(256), 0, s, p, v, n);
  PH_HIP(hipGetLastError());
  return 0;
}

__global__ void ph_scatter_vec2node_kernel(const uint32_t *nodes, uint32_t n, uint32_t *vec2node) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    vec2node[nodes[i]] = i;
}
int ph_scatter_vec2node(const uint32_t *nodes, uint32_t n, uint32_t *vec2node, hipStream_t s) {
  if (n == 0) return 0;
  uint32_t blocks = std::min<uint32_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(ph_scatter_vec2node_kernel, dim3(blocks), dim3(256), 0, s, nodes, n, vec2node);
  PH_HIP(hipGetLastError());
  return 0;
}

__global__ void ph_count_nan_kernel(const float *rows, uint64_t n, uint32_t *out) {
  uint32_t c = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    float f = rows[i];
    c += (f != f) ? 1u : 0u;
  }
  if (c) atomicAdd(out, c);
}
int ph_count_nan(const float *rows, uint64_t n_floats, uint32_t *out_count_dev, hipStream_t s) {
  if (n_floats == 0) return 0;
  uint32_t blocks = (uint32_t)std::min<uint64_t>((n_floats + 255) / 256, 8192);
  hipLaunchKernelGGL(ph_count_nan_kernel, dim3(blocks), dim3(256), 0, s, rows, n_floats, out_count_dev);
  PH_HIP(hipGetLastError());
  return 0;
}

// ---- scratch pool (see phnsw_internal.h) ----
#include <map>
#include <mutex>
#include <unordered_map>
namespace {
std::mutex g_pool_mutex;
std::multimap<size_t, void *> g_pool_free;          // size class -> cached block
std::unordered_map<void *, size_t> g_pool_size;     // every live or cached block -> its size class
size_t g_pool_cached = 0;
const size_t POOL_CACHE_LIMIT = 24ull << 30;
size_t cb0(size_t c) { return c & ((1ull << 56) - 1); }
size_t pool_class(size_t bytes) {
  size_t c = 4096;
  while (c < bytes) c += c < (1ull << 20) ? c : std::max<size_t>(c / 4, 1);  // x2 up to 1 MiB, then x1.25
  return c;
}
}  // namespace

// Device allocation is host work (page tables): its wall time is accounted so that a slow build can be told from a
// slow allocator (phnsw_debug_alloc_stats; bench.py prints it next to the build time).
static std::atomic<uint64_t> g_alloc_ns{0}, g_alloc_calls{0}, g_alloc_bytes{0};
hipError_t ph_timed_malloc(void **p, size_t bytes) {
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(p, bytes);
  g_alloc_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  g_alloc_calls++;
  if (e == hipSuccess) g_alloc_bytes += bytes;
  static const bool log_it = getenv("PHNSW_ALLOC_LOG") != nullptr;
  if (log_it && bytes >= (64u << 20)) fprintf(stderr, "[phnsw] hipMalloc %.2f GB\n", bytes / 1e9);
  return e;
}
hipError_t ph_timed_free(void *p) {
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipFree(p);
  g_alloc_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  g_alloc_calls++;
  return e;
}
// out[0] = nanoseconds inside hipMalloc / hipFree, out[1] = calls, out[2] = bytes allocated -- since the last call
extern "C" void phnsw_debug_alloc_stats(uint64_t *out) {
  out[0] = g_alloc_ns.exchange(0);
  out[1] = g_alloc_calls.exchange(0);
  out[2] = g_alloc_bytes.exchange(0);
}

hipError_t ph_pool_alloc(void **p, size_t bytes) {
  int dev = 0;
  hipGetDevice(&dev);
  // blocks are cached per device: the size class carries the device in its top byte
  const size_t c = pool_class(std::max<size_t>(bytes, 1)) | ((size_t)dev << 56);
  {
    std::lock_guard<std::mutex> g(g_pool_mutex);
    auto it = g_pool_free.find(c);
    if (it != g_pool_free.end()) {
      *p = it->second;
      g_pool_free.erase(it);
      g_pool_cached -= cb0(c);
      return hipSuccess;
    }
  }
  const size_t cb = c & ((1ull << 56) - 1);
  hipError_t e = ph_timed_malloc(p, cb);
  if (e != hipSuccess) {  // memory held by the cache may be what is missing
    ph_pool_trim();
    e = ph_timed_malloc(p, cb);
  }
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> g(g_pool_mutex);
    g_pool_size[*p] = c;
  }
  return e;
}

void ph_pool_free(void *p) {
  if (!p) return;
  std::lock_guard<std::mutex> g(g_pool_mutex);
  auto it = g_pool_size.find(p);
  if (it == g_pool_size.end()) {
    ph_timed_free(p);
    return;
  }
  if (g_pool_cached + cb0(it->second) > POOL_CACHE_LIMIT) {
    g_pool_size.erase(it);
    ph_timed_free(p);
    return;
  }
  g_pool_free.emplace(it->second, p);
  g_pool_cached += cb0(it->second);
}

void ph_pool_trim(void) {
  std::lock_guard<std::mutex> g(g_pool_mutex);
  for (auto &kv : g_pool_free) {
    g_pool_size.erase(kv.second);
    ph_timed_free(kv.second);
  }
  g_pool_free.clear();
  g_pool_cached = 0;
}

// ---- a stream that runs beside another one ----
// HIP maps a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and two streams that
// share one run their work in issue order: a pipeline over such a pair does not overlap at all (measured on the host
// path: 15.2 instead of 12.4 ms per 10 000 queries when the process had made exactly two streams before -- or six with
// eight queues).  A candidate is tried: a kernel that spins for a millisecond on `other`, an empty one on the
// candidate; if the empty one has to wait, the next candidate lands on another queue.  *io: nullptr, or a stream to
// keep if it passes.
__global__ void ph_spin_kernel(uint64_t ticks) {
  const uint64_t t0 = wall_clock64();  // 100 MHz
  while (wall_clock64() - t0 < ticks) {
  }
}
__global__ void ph_noop_kernel() {}

int ph_stream_beside(hipStream_t other, hipStream_t *io) {
  const bool check = getenv("PHNSW_NO_STREAM_CHECK") == nullptr;
  // a rejected candidate is destroyed only AFTER its successor exists: destroyed first, its queue slot goes straight
  // to the successor and the collision repeats
  hipStream_t rejected = nullptr;
  struct Drop {
    hipStream_t *s;
    ~Drop() {
      if (*s) hipStreamDestroy(*s);
    }
  } drop{&rejected};
  for (int attempt = 0; attempt < 6; attempt++) {
    if (!*io) PH_HIP(hipStreamCreateWithFlags(io, hipStreamNonBlocking));
    if (rejected) {
      hipStreamDestroy(rejected);
      rejected = nullptr;
    }
    if (!check) return 0;
    PH_HIP(hipStreamSynchronize(other));
    hipLaunchKernelGGL(ph_noop_kernel, dim3(1), dim3(64), 0, *io);  // first-launch costs out of the way
    PH_HIP(hipStreamSynchronize(*io));
    hipLaunchKernelGGL(ph_spin_kernel, dim3(1), dim3(64), 0, other, (uint64_t)100000);  // 1 ms
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(ph_noop_kernel, dim3(1), dim3(64), 0, *io);
    PH_HIP(hipStreamSynchronize(*io));
    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    PH_HIP(hipStreamSynchronize(other));
    PH_HIP(hipGetLastError());
    if (waited < 0.5e-3 || attempt == 5) return 0;  // side by side (or out of candidates: a shared queue still works)
    rejected = *io;
    *io = nullptr;
  }
  return 0;
}
