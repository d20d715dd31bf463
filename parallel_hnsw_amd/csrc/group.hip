// Partition groups of generate_layer on the device (/root/reference/src/lib.rs:711-713,
// search.rs:62-69): every node of the new layer is keyed by its nearest node found in the
// layers above; the members of one key form a group, ordered by (first distance, node id) --
// the deterministic form of the reference's unstable sort.  One 64-bit radix sort of
// (key << 32 | ordered-float(distance)) with the node id as the stable tie-break, then one
// pass that marks where each key starts.  HBM-bound integer work; rocPRIM's radix sort is the
// library routine for it.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

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
  hipError_t e = hipMalloc(&keys_in, (size_t)n * 8);
  if (e == hipSuccess) e = hipMalloc(&keys_out, (size_t)n * 8);
  if (e == hipSuccess) e = hipMalloc(&vals_in, (size_t)n * 4);
  if (e == hipSuccess)
    e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, gm, (int)n, 0, 64, 0);
  if (e == hipSuccess) e = hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16));
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
  if (keys_in) hipFree(keys_in);
  if (keys_out) hipFree(keys_out);
  if (vals_in) hipFree(vals_in);
  if (tmp) hipFree(tmp);
  return rc;
}
