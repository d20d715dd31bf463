// prints which XCC_ID values workgroups see (scheduling hint used by search.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) o[blockIdx.x] = x;
}
int main() {
  const int n = 4096;
  unsigned *d, h[n];
  hipMalloc(&d, n * 4);
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d);
  hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
  int hist[16] = {0}, match = 0;
  unsigned ormask = 0;
  for (int i = 0; i < n; i++) { hist[h[i] & 15]++; ormask |= h[i]; match += ((h[i] & 7) == (unsigned)(i & 7)); }
  printf("raw OR of all values: 0x%x; first 16 blocks:", ormask);
  for (int i = 0; i < 16; i++) printf(" %x", h[i]);
  printf("\nhistogram of (value & 15):");
  for (int i = 0; i < 16; i++) printf(" %d", hist[i]);
  printf("\nblocks with (value & 7) == blockIdx %% 8: %d of %d\n", match, n);
  return 0;
}
