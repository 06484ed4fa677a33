// fp32 product D[32x32] = A[32xK] * B[KxN] on v_mfma_f32_32x32x16_bf16 with both operands split into bf16 (hi, lo) pairs:
// hi*hi + hi*lo + lo*hi (the lo*lo term, 2^-16 relative, is dropped).  Checks the operand layout against a host reference and
// prints the error next to v_mfma_f32_32x32x2_f32, and the cycles of both.   hipcc --offload-arch=gfx950 -O3 mfma_bf16x3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define K 32
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)x[e];
    hi[e] = h;
    lo[e] = (__bf16)(x[e] - (float)h);
  }
}
// A [32][K] row-major, B [K][32] row-major, D [32][32]
__global__ void probe(const float* A, const float* B, float* D3, float* D1, long* cyc) {
  const int lane = threadIdx.x, l32 = lane & 31, half = lane >> 5;
  f32x16 d3 = {0}, d1 = {0};
  long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int s = 0; s < K / 16; ++s) {
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = A[l32 * K + 16 * s + 8 * half + e]; b[e] = B[(16 * s + 8 * half + e) * 32 + l32]; }
    bf16x8 ah, al, bh, bl;
    split8(a, ah, al); split8(b, bh, bl);
    d3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, d3, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, d3, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, d3, 0, 0, 0);
  }
  long t1 = __builtin_readcyclecounter();
#pragma unroll
  for (int kk = 0; kk < K / 2; ++kk)
    d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l32 * K + 2 * kk + half], B[(2 * kk + half) * 32 + l32], d1, 0, 0, 0);
  long t2 = __builtin_readcyclecounter();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = 8 * (r >> 2) + 4 * half + (r & 3);
    D3[m * 32 + l32] = d3[r];
    D1[m * 32 + l32] = d1[r];
  }
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
  float hA[32 * K], hB[K * 32], hD3[1024], hD1[1024];
  srand(1);
  for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  for (auto& v : hB) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  float *A, *B, *D3, *D1; long* cyc;
  (void)hipMalloc(&A, sizeof hA); (void)hipMalloc(&B, sizeof hB); (void)hipMalloc(&D3, 4096); (void)hipMalloc(&D1, 4096); (void)hipMalloc(&cyc, 16);
  (void)hipMemcpy(A, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(B, hB, sizeof hB, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, A, B, D3, D1, cyc);
  (void)hipDeviceSynchronize();
  long c[2];
  (void)hipMemcpy(hD3, D3, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(hD1, D1, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
  double e3 = 0, e1 = 0, nrm = 0, m3 = 0;
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)hA[m * K + k] * hB[k * 32 + n];
      e3 += (hD3[m * 32 + n] - ref) * (hD3[m * 32 + n] - ref);
      e1 += (hD1[m * 32 + n] - ref) * (hD1[m * 32 + n] - ref);
      nrm += ref * ref;
      m3 = fmax(m3, fabs(hD3[m * 32 + n] - ref));
    }
  printf("K = %d: bf16x3 rel. error %.3g (max abs %.3g), fp32 MFMA rel. error %.3g; cycles incl. loads: bf16x3 %ld, fp32 %ld\n", K,
         sqrt(e3 / nrm), m3, sqrt(e1 / nrm), c[0], c[1]);
  return 0;
}
