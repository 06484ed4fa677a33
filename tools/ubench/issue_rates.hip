// Issue-rate probes for gfx950 (one wave on one SIMD): plain VALU, packed fp32, DPP wave_shr / row_shr, MFMA 32x32x2 f32 alone
// and interleaved with VALU.  Prints shader cycles per instruction.   hipcc --offload-arch=gfx950 -O3 issue_rates.hip -o issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ITERS 2048
__device__ __forceinline__ float shr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float rowshr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
template <int MODE>
__global__ void probe(float* out, long* cyc) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) p[i] = f32x2{a[i], a[i] + 1.f};
  f32x16 d0 = {0}, d1 = {0};
  const float x = out[0], y = out[1];
  long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
    if (MODE == 0) {   // 16 independent-ish fma
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = a[i] * x + y;
    } else if (MODE == 1) {   // 16 pk_fma
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = p[i] * f32x2{x, x} + f32x2{y, y};
    } else if (MODE == 2) {   // 16 add with dpp wave_shr (8 chains)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = shr1(a[i]) + x;
    } else if (MODE == 3) {   // 16 add with dpp row_shr
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = rowshr1(a[i]) + x;
    } else if (MODE == 4) {   // 2 mfma alone
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, d1, 0, 0, 0);
    } else if (MODE == 5) {   // 2 mfma + 32 fma interleaved
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, d0, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = a[i] * x + y;
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, d1, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = a[i] * x + y;
    } else if (MODE == 6) {   // 32 fma alone (reference for 5)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = a[i] * x + y;
    } else if (MODE == 7) {   // 1 chain of dependent dpp adds (16)
#pragma unroll
      for (int r = 0; r < 16; ++r) a[0] = shr1(a[0]) + x;
    }
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
  }
  long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  for (int i = 0; i < 16; ++i) s += d0[i] + d1[i];
  out[2 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(const char* name, int instr, float* out, long* cyc) {
  hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(64), 0, 0, out, cyc);
  hipDeviceSynchronize();
  long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s %8.2f cycles/iter  %6.2f per instr\n", name, (double)c / ITERS, (double)c / ITERS / instr);
}
int main() {
  float* out; long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
  float h[2] = {1.0001f, 0.5f};
  hipMemcpy(out, h, 8, hipMemcpyHostToDevice);
  run<0>("16 v_fma_f32 (8 chains)", 16, out, cyc);
  run<1>("16 v_pk_fma_f32 (8 chains)", 16, out, cyc);
  run<2>("16 v_add_f32_dpp wave_shr:1 (8 chains)", 16, out, cyc);
  run<3>("16 v_add_f32_dpp row_shr:1 (8 chains)", 16, out, cyc);
  run<7>("16 v_add_f32_dpp wave_shr:1 (1 chain)", 16, out, cyc);
  run<4>("2 v_mfma_f32_32x32x2_f32", 2, out, cyc);
  run<6>("32 v_fma_f32", 32, out, cyc);
  run<5>("2 mfma + 32 v_fma interleaved", 34, out, cyc);
  return 0;
}
