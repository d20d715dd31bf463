// micro-experiment: is a chain of v_mfma_f32_32x32x1_2b_f32 (K = 1 per instruction) bit-identical to a chain of
// v_fma_f32 in the same order?  (decides whether the dense-layer distance tables can move to the matrix cores
// without changing a single bit)   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off mfma_fma_exact.hip -o mfma_fma_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x32 __attribute__((ext_vector_type(32)));

__global__ void k_mfma(const float *A, const float *B, float *D, int K) {
  // A[b][i][k] at A[(b*32+i)*K + k], B[b][k][j] at B[(b*K + k)*32 + j]; lane l: block l/32, index l%32
  const int l = threadIdx.x, b = l >> 5, x = l & 31;
  f32x32 acc;
  for (int v = 0; v < 32; v++) acc[v] = 0.f;
  for (int k = 0; k < K; k++) {
    float a = A[(b * 32 + x) * K + k];
    float bb = B[(b * K + k) * 32 + x];
    acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a, bb, acc, 0, 0, 0);
  }
  for (int v = 0; v < 32; v++) D[v * 64 + l] = acc[v];
}

__global__ void k_fma(const float *A, const float *B, float *R, int K) {
  // R[b][i][j]: one thread per output
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * 32 * 32) return;
  const int b = t >> 10, i = (t >> 5) & 31, j = t & 31;
  float acc = 0.f;
  for (int k = 0; k < K; k++) acc = __builtin_fmaf(A[(b * 32 + i) * K + k], B[(b * K + k) * 32 + j], acc);
  R[t] = acc;
}

int main() {
  const int K = 12;
  int bad_total = 0;
  for (int trial = 0; trial < 6; trial++) {
    std::vector<float> A(2 * 32 * K), B(2 * K * 32);
    srand(1234 + trial);
    for (auto *vec : {&A, &B})
      for (auto &x : *vec) {
        float u = (float)rand() / RAND_MAX * 2.f - 1.f;
        if (trial == 0) x = (float)(rand() % 17 - 8);                      // integers: exact, checks the layout
        else if (trial == 1) x = u;                                       // unit-range values (the real workload)
        else if (trial == 2) x = u * 1e-19f;                              // products are denormal
        else if (trial == 3) x = u * ((rand() & 1) ? 1e18f : 1e-18f);     // wide exponent spread, cancellation
        else if (trial == 4) x = (rand() % 5 == 0) ? 0.f : u * 1e-22f;    // denormal inputs' neighbourhood
        else { uint32_t r = ((uint32_t)rand() << 16) ^ (uint32_t)rand(); r &= 0x807FFFFFu; r |= (uint32_t)(100 + rand() % 56) << 23; memcpy(&x, &r, 4); }
      }
    float *dA, *dB, *dD, *dR;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 32 * 64 * 4); hipMalloc(&dR, 2048 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(dA, dB, dD, K);
    k_fma<<<8, 256>>>(dA, dB, dR, K);
    std::vector<float> D(32 * 64), R(2048);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(R.data(), dR, R.size() * 4, hipMemcpyDeviceToHost);
    if (hipDeviceSynchronize() != hipSuccess) { printf("hip error\n"); return 2; }
    // assumed layout: vgpr v (0..15 block 0, 16..31 block 1), lane l: j = l%32, i = 8*(v%16/4) + 4*(l/32) + v%4
    int bad = 0, sign0 = 0;
    for (int v = 0; v < 32; v++)
      for (int l = 0; l < 64; l++) {
        int b = v / 16, vv = v % 16, j = l & 31, i = 8 * (vv / 4) + 4 * (l >> 5) + (vv & 3);
        float got = D[v * 64 + l], exp = R[(b * 32 + i) * 32 + j];
        if (memcmp(&got, &exp, 4) != 0) {
          if (got == exp) sign0++;  // +0 vs -0
          else if (bad++ < 4) printf("  trial %d v%d lane %d: mfma %a fma %a\n", trial, v, l, got, exp);
        }
      }
    printf("trial %d: %d of 2048 outputs differ (%d more differ only in the sign of zero)\n", trial, bad, sign0);
    bad_total += bad;
    hipFree(dA); hipFree(dB); hipFree(dD); hipFree(dR);
  }
  printf(bad_total ? "MFMA chain != FMA chain\n" : "MFMA chain == FMA chain on every trial\n");
  return 0;
}
