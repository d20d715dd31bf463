// Partition groups of generate_layer on the device (/root/reference/src/lib.rs:711-713,
// search.rs:62-69): every node of the new layer is keyed by its nearest node found in the
// layers above; the members of one key form a group, ordered by (first distance, node id) --
// the deterministic form of the reference's unstable sort.  One 64-bit radix sort of
// (key << 32 | ordered-float(distance)) with the node id as the stable tie-break, then one
// pass that marks where each key starts.  HBM-bound integer work; rocPRIM's radix sort is the
// library routine for it.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "phnsw_internal.h"

#include "phnsw_device.h"

__global__ void ph_group_keys_kernel(const uint32_t *init_ids, const float *init_d, const uint32_t *init_len,
                                     uint32_t K, uint32_t n, uint64_t *keys, uint32_t *vals) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool some = init_len[i] != 0;
  uint32_t key = some ? init_ids[(size_t)i * K] : PH_EMPTY32;  // None sorts last
  float d = some ? init_d[(size_t)i * K] : 0.f;
  keys[i] = ((uint64_t)key << 32) | fkey(d);
  vals[i] = i;
}

// gstart[slot] = first position of the key, gsize[slot] = members; slot n holds the None group
__global__ void ph_group_bounds_kernel(const uint64_t *keys, uint32_t n, uint32_t *gstart, uint32_t *gsize) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t k = (uint32_t)(keys[p] >> 32);
  uint32_t slot = k == PH_EMPTY32 ? n : k;
  if (p == 0 || (uint32_t)(keys[p - 1] >> 32) != k) gstart[slot] = p;
  atomicAdd(&gsize[slot], 1u);
}

// gm [n], gstart [n+1], gsize [n+1] are device buffers owned by the caller
int ph_build_groups_device(const uint32_t *init_ids, const float *init_d, const uint32_t *init_len, uint32_t K,
                           uint32_t n, uint32_t *gm, uint32_t *gstart, uint32_t *gsize) {
  uint64_t *keys_in = nullptr, *keys_out = nullptr;
  uint32_t *vals_in = nullptr;
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  int rc = 0;
  hipError_t e = ph_pool_alloc((void **)&keys_in, (size_t)n * 8);
  if (e == hipSuccess) e = ph_pool_alloc((void **)&keys_out, (size_t)n * 8);
  if (e == hipSuccess) e = ph_pool_alloc((void **)&vals_in, (size_t)n * 4);
  if (e == hipSuccess)
    e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, gm, (int)n, 0, 64, 0);
  if (e == hipSuccess) e = ph_pool_alloc((void **)&tmp, std::max<size_t>(tmp_bytes, 16));
  if (e == hipSuccess) {
    uint32_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(ph_group_keys_kernel, dim3(blocks), dim3(256), 0, 0, init_ids, init_d, init_len, K, n, keys_in,
                       vals_in);
    e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, gm, (int)n, 0, 64, 0);
    if (e == hipSuccess) e = hipMemsetAsync(gstart, 0, (size_t)(n + 1) * 4, 0);
    if (e == hipSuccess) e = hipMemsetAsync(gsize, 0, (size_t)(n + 1) * 4, 0);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(ph_group_bounds_kernel, dim3(blocks), dim3(256), 0, 0, keys_out, n, gstart, gsize);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  if (e != hipSuccess) rc = ph_hip_fail(e, "partition groups (radix sort)", __FILE__, __LINE__);
  if (keys_in) ph_pool_free(keys_in);
  if (keys_out) ph_pool_free(keys_out);
  if (vals_in) ph_pool_free(vals_in);
  if (tmp) ph_pool_free(tmp);
  return rc;
}

// ---- locality schedule (a scheduling hint, never part of a result) ----
// Every large layer carries pos[node] = the node's coarse cell: its exact nearest anchor, the
// anchors being a strided sample of the layer (bruteforce.hip, one MFMA GEMM pass).  The search
// kernel walks a query list sorted by cell, one contiguous eighth per XCD, so that the rows
// one L2 / the Infinity Cache serve together belong to neighbouring queries (search.hip).

__global__ void ph_iota_kernel(uint32_t *v, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}

// order_out [n] = argsort (stable) of 32-bit keys; used to turn per-query positions into the
// processing order of one launch
int ph_order_by_keys_device(const uint32_t *keys, uint32_t n, uint32_t *order_out, hipStream_t st) {
  uint32_t *keys_out = nullptr, *vals_in = nullptr;
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  int rc = 0;
  hipError_t e = ph_pool_alloc((void **)&keys_out, (size_t)n * 4);
  if (e == hipSuccess) e = ph_pool_alloc((void **)&vals_in, (size_t)n * 4);
  if (e == hipSuccess)
    e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys_out, vals_in, order_out, (int)n, 0, 32, st);
  if (e == hipSuccess) e = ph_pool_alloc((void **)&tmp, std::max<size_t>(tmp_bytes, 16));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(ph_iota_kernel, dim3((n + 255) / 256), dim3(256), 0, st, vals_in, n);
    e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys_out, vals_in, order_out, (int)n, 0, 32, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
  }
  if (e != hipSuccess) rc = ph_hip_fail(e, "query order (radix sort)", __FILE__, __LINE__);
  if (keys_out) ph_pool_free(keys_out);
  if (vals_in) ph_pool_free(vals_in);
  if (tmp) ph_pool_free(tmp);
  return rc;
}

// start[0..n] = exclusive prefix sums of cnt[0..n] (cnt[n] must be 0, so start[n] = total); async on `st`
int ph_exclusive_scan_u32(const uint32_t *cnt, uint32_t n_plus_1, uint32_t *start, hipStream_t st) {
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cnt, start, (int)n_plus_1, st);
  if (e == hipSuccess) e = ph_pool_alloc(&tmp, std::max<size_t>(tmp_bytes, 16));
  if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, cnt, start, (int)n_plus_1, st);
  if (tmp) ph_pool_free(tmp);  // same stream: the next user is ordered behind the scan
  if (e != hipSuccess) return ph_hip_fail(e, "prefix sum", __FILE__, __LINE__);
  return 0;
}

// ascending sort of host u32 keys through the device (the node list of a big layer arrives in the
// shuffled order of Hnsw::generate, lib.rs:832-833: 47 ms with std::sort at 1M, 2 ms this way)
int ph_sort_u32_host(uint32_t *keys, uint32_t n) {
  uint32_t *in = nullptr, *out = nullptr;
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  int rc = 0;
  hipError_t e = ph_pool_alloc((void **)&in, (size_t)n * 4);
  if (e == hipSuccess) e = ph_pool_alloc((void **)&out, (size_t)n * 4);
  if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, in, out, (int)n, 0, 32, (hipStream_t)0);
  if (e == hipSuccess) e = ph_pool_alloc(&tmp, std::max<size_t>(tmp_bytes, 16));
  if (e == hipSuccess) e = hipMemcpy(in, keys, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, in, out, (int)n, 0, 32, (hipStream_t)0);
  if (e == hipSuccess) e = hipMemcpy(keys, out, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) rc = ph_hip_fail(e, "node list sort", __FILE__, __LINE__);
  if (in) ph_pool_free(in);
  if (out) ph_pool_free(out);
  if (tmp) ph_pool_free(tmp);
  return rc;
}

// ---- per-workspace argsort of the queries' locality keys, fully asynchronous on `stream`
// (buffers grow only when a larger batch arrives)
void ph_workspace_order_free(PhWorkspace &ws) {
  if (ws.okey) hipFree(ws.okey);
  if (ws.okey_sorted) hipFree(ws.okey_sorted);
  if (ws.oiota) hipFree(ws.oiota);
  if (ws.oorder) hipFree(ws.oorder);
  if (ws.sort_tmp) hipFree(ws.sort_tmp);
  ws.okey = ws.okey_sorted = ws.oiota = ws.oorder = nullptr;
  ws.sort_tmp = nullptr;
  ws.sort_tmp_bytes = 0;
  ws.order_cap = 0;
}

int ph_workspace_order_ensure(PhWorkspace &ws, uint32_t nq) {
  if (ws.order_cap >= nq) return 0;
  ph_workspace_order_free(ws);
  uint32_t cap = std::max<uint32_t>(nq, 65536u);
  PH_HIP(hipMalloc(&ws.okey, (size_t)cap * 4));
  PH_HIP(hipMalloc(&ws.okey_sorted, (size_t)cap * 4));
  PH_HIP(hipMalloc(&ws.oiota, (size_t)cap * 4));
  PH_HIP(hipMalloc(&ws.oorder, (size_t)cap * 4));
  size_t bytes = 0;
  PH_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, ws.okey, ws.okey_sorted, ws.oiota, ws.oorder, (int)cap, 0, 32,
                                            (hipStream_t)0));
  ws.sort_tmp_bytes = std::max<size_t>(bytes, 16);
  PH_HIP(hipMalloc(&ws.sort_tmp, ws.sort_tmp_bytes));
  hipLaunchKernelGGL(ph_iota_kernel, dim3((cap + 255) / 256), dim3(256), 0, 0, ws.oiota, cap);
  PH_HIP(hipDeviceSynchronize());
  ws.order_cap = cap;
  return 0;
}

// ws.oorder[0..nq) = argsort(ws.okey[0..nq))
int ph_workspace_order_sort(PhWorkspace &ws, uint32_t nq, hipStream_t stream) {
  size_t bytes = ws.sort_tmp_bytes;
  PH_HIP(hipcub::DeviceRadixSort::SortPairs(ws.sort_tmp, bytes, ws.okey, ws.okey_sorted, ws.oiota, ws.oorder, (int)nq, 0,
                                            32, stream));
  return 0;
}
