// Shared device/host helpers for the rt_pose_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtp.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define RTP_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }  // RNE, NaN preserving (v_cvt_pk_bf16_f32)

__device__ __forceinline__ bf16x8 ld_bf16x8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void st_bf16x8(bf16_t* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }
// Streaming store (global_store ... nt): the line does not linger dirty in the XCD's L2.  A kernel's dirty L2 lines are written
// back at its END, before the next kernel of the stream may start -- up to 32 MB, measured 6.4 us between the last workgroup's
// exit and the next launch's first instruction for a kernel that writes an 84 MB tensor with default-policy stores.
__device__ __forceinline__ void st_bf16x8_nt(bf16_t* p, bf16x8 v) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(__builtin_bit_cast(v4i, v), reinterpret_cast<v4i*>(p));
}
__device__ __forceinline__ bf16x8 zero_bf16x8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16_t)0.0f;
  return z;
}

// wave-level sum (64 lanes) through DPP-free shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// flattened voxel index -> (z,y,x)
__device__ __forceinline__ void vox_decode(int v, int H, int W, int& z, int& y, int& x) {
  x = v % W;
  int t = v / W;
  y = t % H;
  z = t / H;
}

// boundary class of an output voxel: bit0 first-z, bit1 last-z, bit2 first-y, bit3 last-y, bit4 first-x, bit5 last-x
__device__ __forceinline__ int vox_class(int z, int y, int x, int D, int H, int W) {
  return (z == 0) | ((z == D - 1) << 1) | ((y == 0) << 2) | ((y == H - 1) << 3) | ((x == 0) << 4) | ((x == W - 1) << 5);
}

// is kernel tap k (one dim) inside the input for an output position with the given first/last flags?
//   normal conv:      p = o*s + k - pad           must satisfy 0 <= p < I
//   (flags only matter at o==0 / o==O-1; interior positions of a pad<=1,k<=3 conv are always in bounds)
__host__ __device__ __forceinline__ bool tap_inb_1d(int k, int first, int last, int O, int I, int s, int pad) {
  if (first) {
    int p = 0 * s + k - pad;
    if (p < 0 || p >= I) return false;
  }
  if (last) {
    int p = (O - 1) * s + k - pad;
    if (p < 0 || p >= I) return false;
  }
  return true;
}

#define RTP_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return RTP_ERR_LAUNCH;  \
  } while (0)

static inline int rtp_div_up(long a, long b) { return (int)((a + b - 1) / b); }

// One-time per-DEVICE set-up (hipFuncSetAttribute and cached device properties are per device: a process-wide `static bool`
// would leave a second GPU of the same process without its LDS limit).  `flags` is a static bool[RTP_MAX_DEVICES]; true exactly
// once per device.
#define RTP_MAX_DEVICES 16
static inline int rtp_device_index() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= RTP_MAX_DEVICES) d = 0;
  return d;
}
static inline bool rtp_once_per_device(bool* flags) {
  const int d = rtp_device_index();
  if (flags[d]) return false;
  flags[d] = true;
  return true;
}
