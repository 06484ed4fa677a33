// Weight-gradient correlation on MFMA (gfx950):  G[co][tap][ci] = sum_v gy[v][co] * x[v*s + tap - pad][ci].
//
// The contraction runs over VOXELS, but activations are channels-last (channel contiguous), so both MFMA
// operands need a transpose: tiles are staged voxel-major in LDS and read with ds_read_b64_tr_b16, which
// hands each lane 4 consecutive voxels of one channel (cdna_hip_programming.md T10).
//   A = gy^T  [16 co ][32 voxels]   B = x_tap [32 voxels][16 ci]   D[co][ci] (fp32, 4 regs)
// Block = 4 waves; the 27 taps are dealt round-robin to the waves (<= 7 each, 16 acc regs per tap), every
// wave gathers its own shifted x chunk for the tap it is working on.  Output: fp32 partial slabs per
// (sample, voxel split) -- reduced deterministically by rtp_wgrad_fold (no atomics).
#include <stdlib.h>

#include "rtp_common.h"
#include "rtp_multi.h"
#include "rtp_prof.h"

#define WG_VB 128  // voxels per staged chunk (41 KB of LDS per block: three blocks per CU overlap each other's gathers)

struct WgradParams {
  const bf16_t* gy; const bf16_t* x; float* gp;
  int N, Di, Hi, Wi, Do, Ho, Wo, ks, stride, pad;
  int g_cs, g_co, x_cs, x_co;
  int ci_pad, co32, ntap, nsplit, citiles;
  int Vo; int vps;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned wg_u32x4;

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile /*[vox][32]*/, int kstep, int sub, int lane) {
  const int q = lane >> 4, i = lane & 15, a = i >> 2, pp = i & 3;
  const bf16_t* p0 = tile + ((kstep * 32 + q * 8 + a) * 32 + sub * 16 + pp * 4);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * 32));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

#define WG_ITEMS (WG_VB * 4 / 64)  // 16-B items per lane and tap: 8

// A wave's x chunk for its current tap lives in its own LDS buffer, so inside a chunk the waves never meet at a
// barrier: each one stores the gathered registers, issues the NEXT tap's gather (buffer loads: per-lane offset and
// per-axis validity bits are computed once per chunk, the tap only changes the wave-uniform soffset; out-of-volume
// lanes load through an out-of-range offset that the descriptor turns into zeros) and runs this tap's MFMAs under it.
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradParams p) {
  __shared__ __attribute__((aligned(16))) bf16_t gyL[WG_VB * 32];
  __shared__ __attribute__((aligned(16))) bf16_t xL[4][WG_VB * 32];
  __shared__ int coordL[WG_VB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x, n = blockIdx.y;
  const int cot = blockIdx.z / p.citiles, cit = blockIdx.z % p.citiles;
  const int v_begin = s * p.vps;
  const int v_end = (v_begin + p.vps < p.Vo) ? v_begin + p.vps : p.Vo;

  f32x4 acc[7][2][2];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long g_n = (long)n * p.Vo;
  const int P = p.Hi * p.Wi + p.Wi + 1;  // descriptor base moved back so every tap's uniform offset is >= 0
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.x + ((long)n * p.Di * p.Hi * p.Wi - P) * p.x_cs + p.x_co + cit * 32), 0, 0x7fffffff, 0x00020000);
  const int ks2 = p.ks * p.ks;

  for (int vc = v_begin; vc < v_end; vc += WG_VB) {
    __syncthreads();  // previous chunk fully consumed
    if (tid < WG_VB) {
      const int v = vc + tid;
      int z = 0, y = 0, x = 0;
      if (v < v_end) vox_decode(v, p.Ho, p.Wo, z, y, x);
      coordL[tid] = (v < v_end) ? ((z << 20) | (y << 10) | x) : -1;
    }
#pragma unroll
    for (int r = 0; r < WG_VB * 4 / 256; ++r) {
      const int item = tid + 256 * r, vox = item >> 2, ck = item & 3;
      bf16x8 val = zero_bf16x8();
      if (vc + vox < v_end) val = ld_bf16x8(p.gy + (g_n + vc + vox) * p.g_cs + p.g_co + cot * 32 + ck * 8);
      st_bf16x8(&gyL[vox * 32 + ck * 8], val);
    }
    __syncthreads();
    // per item: byte offset of the tap-(pad,pad,pad) input voxel and the 9 per-axis validity bits
    int boff[WG_ITEMS];
    unsigned okm[WG_ITEMS];
#pragma unroll
    for (int r = 0; r < WG_ITEMS; ++r) {
      const int item = lane + 64 * r, vox = item >> 2, ck = item & 3;
      const int cd = coordL[vox];
      const int z = (cd >> 20) * p.stride, y = ((cd >> 10) & 1023) * p.stride, x = (cd & 1023) * p.stride;
      boff[r] = (((z * p.Hi + y) * p.Wi + x) * p.x_cs + ck * 8) * 2;
      unsigned m = 0;
      if (cd >= 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          m |= (unsigned)((unsigned)(z + k - p.pad) < (unsigned)p.Di) << k;
          m |= (unsigned)((unsigned)(y + k - p.pad) < (unsigned)p.Hi) << (3 + k);
          m |= (unsigned)((unsigned)(x + k - p.pad) < (unsigned)p.Wi) << (6 + k);
        }
      }
      okm[r] = m;
    }
    bf16x8 val[WG_ITEMS];
    auto gather = [&](int tap) {
      const int kz = tap / ks2, ky = (tap - kz * ks2) / p.ks, kx = tap - kz * ks2 - ky * p.ks;
      const int soff = ((((kz - p.pad) * p.Hi + (ky - p.pad)) * p.Wi + (kx - p.pad) + P) * p.x_cs) * 2;
#pragma unroll
      for (int r = 0; r < WG_ITEMS; ++r) {
        const bool ok = (okm[r] >> kz) & (okm[r] >> (3 + ky)) & (okm[r] >> (6 + kx)) & 1u;
        const wg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? boff[r] : (int)0x80000000, soff, 0);
        val[r] = __builtin_bit_cast(bf16x8, v);
      }
    };
    if (wave < p.ntap) gather(wave);
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int tap = wave + 4 * t;
      if (tap >= p.ntap) break;  // wave-uniform
#pragma unroll
      for (int r = 0; r < WG_ITEMS; ++r) {
        const int item = lane + 64 * r;
        st_bf16x8(&xL[wave][(item >> 2) * 32 + (item & 3) * 8], val[r]);
      }
      if (tap + 4 < p.ntap) gather(tap + 4);
#pragma unroll
      for (int ks = 0; ks < WG_VB / 32; ++ks) {
        bf16x8 a0 = tr_frag(gyL, ks, 0, lane), a1 = tr_frag(gyL, ks, 1, lane);
        bf16x8 b0 = tr_frag(xL[wave], ks, 0, lane), b1 = tr_frag(xL[wave], ks, 1, lane);
        acc[t][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[t][0][0], 0, 0, 0);
        acc[t][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, acc[t][0][1], 0, 0, 0);
        acc[t][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, acc[t][1][0], 0, 0, 0);
        acc[t][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[t][1][1], 0, 0, 0);
      }
    }
  }

  // D[row = co][col = ci]: lane holds rows 4q..4q+3, column lane&15
  const int q = lane >> 4, i = lane & 15;
  float* out = p.gp + ((long)n * p.nsplit + s) * p.ntap * p.co32 * p.ci_pad;
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int tap = wave + 4 * t;
    if (tap >= p.ntap) continue;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[((long)tap * p.co32 + cot * 32 + a * 16 + q * 4 + r) * p.ci_pad + cit * 32 + b * 16 + i] = acc[t][a][b][r];
  }
}

// 1x1x1 convs (the fuse rows' channel matchers and the concat-free final conv, hr3d.py:147-155, hrnet3d.py:37-41): one tap, so
// the tap-per-wave deal above leaves three of four waves idle and every (output tile, input tile) block re-reads both
// tensors.  Here a block stages a 64-voxel chunk of ALL channels of gy and x once ([tile][voxel][32] images) and its four
// waves split the (co tile, ci tile) pairs, <= 4 each: every byte of both tensors is read once per launch (HBM-bound).
#define W1_VB 64
__global__ __launch_bounds__(256) void wgrad_1x1_kernel(WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds1[];
  const int cotiles = p.co32 >> 5, tiles = cotiles * p.citiles;
  bf16_t* gyL = lds1;                               // [cotiles][W1_VB][32]
  bf16_t* xL = lds1 + cotiles * W1_VB * 32;         // [citiles][W1_VB][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x, n = blockIdx.y;
  const int v_begin = s * p.vps;
  const int v_end = (v_begin + p.vps < p.Vo) ? v_begin + p.vps : p.Vo;
  f32x4 acc[4][2][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long vox_n = (long)n * p.Vo;   // stride 1, no padding: input voxel == output voxel
  const int gck = p.co32 >> 3, xck = p.ci_pad >> 3;
  for (int vc = v_begin; vc < v_end; vc += W1_VB) {
    __syncthreads();
    for (int item = tid; item < W1_VB * gck; item += 256) {
      const int vox = item / gck, ck = item - vox * gck;
      bf16x8 val = zero_bf16x8();
      if (vc + vox < v_end) val = ld_bf16x8(p.gy + (vox_n + vc + vox) * p.g_cs + p.g_co + ck * 8);
      st_bf16x8(&gyL[((ck >> 2) * W1_VB + vox) * 32 + (ck & 3) * 8], val);
    }
    for (int item = tid; item < W1_VB * xck; item += 256) {
      const int vox = item / xck, ck = item - vox * xck;
      bf16x8 val = zero_bf16x8();
      if (vc + vox < v_end) val = ld_bf16x8(p.x + (vox_n + vc + vox) * p.x_cs + p.x_co + ck * 8);
      st_bf16x8(&xL[((ck >> 2) * W1_VB + vox) * 32 + (ck & 3) * 8], val);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int tile = wave + 4 * t;
      if (tile >= tiles) break;   // wave-uniform
      const int cot = tile / p.citiles, cit = tile - cot * p.citiles;
#pragma unroll
      for (int ks = 0; ks < W1_VB / 32; ++ks) {
        bf16x8 a0 = tr_frag(gyL + cot * W1_VB * 32, ks, 0, lane), a1 = tr_frag(gyL + cot * W1_VB * 32, ks, 1, lane);
        bf16x8 b0 = tr_frag(xL + cit * W1_VB * 32, ks, 0, lane), b1 = tr_frag(xL + cit * W1_VB * 32, ks, 1, lane);
        acc[t][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[t][0][0], 0, 0, 0);
        acc[t][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, acc[t][0][1], 0, 0, 0);
        acc[t][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, acc[t][1][0], 0, 0, 0);
        acc[t][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[t][1][1], 0, 0, 0);
      }
    }
  }
  const int q = lane >> 4, i = lane & 15;
  float* out = p.gp + ((long)n * p.nsplit + s) * p.co32 * p.ci_pad;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tile = wave + 4 * t;
    if (tile >= tiles) continue;
    const int cot = tile / p.citiles, cit = tile - cot * p.citiles;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[((long)cot * 32 + a * 16 + q * 4 + r) * p.ci_pad + cit * 32 + b * 16 + i] = acc[t][a][b][r];
  }
}

int rtp_wgrad_tiled_try(const RtpAct* gy, const RtpAct* x, const RtpConvGeom* g, int nsplit, float* gp, hipStream_t s,
                        const void* wd, float* qpart, float* tg, int slab_rows = 0, int slab_cols = 0);

int rtp_wgrad_s2_try(const RtpAct* gy, const RtpAct* x, const RtpConvGeom* g, int nsplit, float* gp, hipStream_t s, int slab_rows,
                     int slab_cols);   // wgrad_s2_tiled.hip

// rtp_wgrad on the LDS-tiled kernel that also contracts every slab with the data-gradient weights (GroupNorm backward's Q
// without a pass over dxhat).  Only the tiled kernel's geometries (rtp_wgrad_nsplit(g) > 0, nsplit equal to it).
extern "C" int rtp_wgrad_q(const RtpAct* gy, const RtpAct* x, const RtpConvGeom* g, int nsplit, float* gp, const void* wd,
                           float* qpart, float* tg, void* stream) {
  if (!gy || !x || !g || !gp || !wd || !qpart || nsplit < 1) return RTP_ERR_SHAPE;
  if ((gy->cs % 8) || (gy->co % 8) || (x->cs % 8) || (x->co % 8)) return RTP_ERR_ALIGN;
  if (gy->c < (g->co + 31) / 32 * 32 || x->c < g->ci) return RTP_ERR_SHAPE;
  if (rtp_wgrad_nsplit(g) != nsplit) return RTP_ERR_UNSUPPORTED;
  const int rc = rtp_wgrad_tiled_try(gy, x, g, nsplit, gp, (hipStream_t)stream, wd, qpart, tg);
  return rc > 0 ? RTP_ERR_UNSUPPORTED : rc;
}

// rtp_wgrad on the LDS-tiled kernel whose loader waves also sum gy over the volume and its faces / edges / corners (tg
// [n][nsplit][27][32], see wgrad_tiled.hip): the bias gradient of a conv WITHOUT GroupNorm is the sum of slot 0 over the
// partials (rtp_tail_desc_wgrad_fold_tg), so no class-sum pass reads gy a second time.  Tiled geometries with 32 channels only.
extern "C" int rtp_wgrad_tg(const RtpAct* gy, const RtpAct* x, const RtpConvGeom* g, int nsplit, float* gp, float* tg, void* stream) {
  if (!gy || !x || !g || !gp || !tg || nsplit < 1) return RTP_ERR_SHAPE;
  if ((gy->cs % 8) || (gy->co % 8) || (x->cs % 8) || (x->co % 8)) return RTP_ERR_ALIGN;
  if (gy->c < 32 || x->c < 32 || g->ci != 32 || (g->co + 31) / 32 * 32 != 32) return RTP_ERR_UNSUPPORTED;
  if (rtp_wgrad_nsplit(g) != nsplit || g->stride != 1) return RTP_ERR_UNSUPPORTED;
  const int rc = rtp_wgrad_tiled_try(gy, x, g, nsplit, gp, (hipStream_t)stream, nullptr, nullptr, tg);
  return rc > 0 ? RTP_ERR_UNSUPPORTED : rc;
}

extern "C" int rtp_wgrad(const RtpAct* gy, const RtpAct* x, const RtpConvGeom* g, int nsplit, float* gp,
                         void* stream) {
  if (!gy || !x || !g || !gp || nsplit < 1) return RTP_ERR_SHAPE;
  if (g->ks != 1 && g->ks != 3) return RTP_ERR_UNSUPPORTED;
  if ((gy->cs % 8) || (gy->co % 8) || (x->cs % 8) || (x->co % 8)) return RTP_ERR_ALIGN;
  if (gy->c < (g->co + 31) / 32 * 32 || x->c < g->ci) return RTP_ERR_SHAPE;
  {
    const int rc = rtp_wgrad_tiled_try(gy, x, g, nsplit, gp, (hipStream_t)stream, nullptr, nullptr, nullptr);
    if (rc <= 0) return rc;
  }
  if (rtp_multi_capture()) return RTP_ERR_UNSUPPORTED;   // only the stride-1 LDS-tiled kernel can be recorded for a shared launch
  {
    const int rc = rtp_wgrad_s2_try(gy, x, g, nsplit, gp, (hipStream_t)stream, 0, 0);
    if (rc <= 0) return rc;
  }
  WgradParams p;
  p.gy = (const bf16_t*)gy->ptr; p.x = (const bf16_t*)x->ptr; p.gp = gp;
  p.N = g->n; p.Di = g->di; p.Hi = g->hi; p.Wi = g->wi; p.Do = g->dov; p.Ho = g->ho; p.Wo = g->wo;
  p.ks = g->ks; p.stride = g->stride; p.pad = g->pad;
  p.g_cs = gy->cs; p.g_co = gy->co; p.x_cs = x->cs; p.x_co = x->co;
  p.ci_pad = g->ci; p.co32 = (g->co + 31) / 32 * 32; p.ntap = g->ks * g->ks * g->ks; p.nsplit = nsplit;
  if (p.ci_pad % 32) return RTP_ERR_UNSUPPORTED;
  if (gy->c < p.co32 || x->c < p.ci_pad) return RTP_ERR_SHAPE;
  if ((gy->cs % 8) || (gy->co % 8) || (x->cs % 8) || (x->co % 8)) return RTP_ERR_ALIGN;
  if (p.Ho >= 1024 || p.Wo >= 1024 || p.Do >= 2048) return RTP_ERR_UNSUPPORTED;
  p.citiles = p.ci_pad / 32;
  p.Vo = p.Do * p.Ho * p.Wo;
  p.vps = rtp_div_up(rtp_div_up(p.Vo, nsplit), WG_VB) * WG_VB;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_WGRAD, s);
  static const bool no_1x1 = false;
  if (!no_1x1 && p.ks == 1 && p.stride == 1 && p.pad == 0 && (p.co32 / 32) * p.citiles <= 16) {
    const size_t shm = sizeof(bf16_t) * (size_t)(p.co32 / 32 + p.citiles) * W1_VB * 32;
    static bool attr[RTP_MAX_DEVICES] = {};
    if (rtp_once_per_device(attr)) {
      (void)hipFuncSetAttribute((const void*)wgrad_1x1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(wgrad_1x1_kernel, dim3(nsplit, p.N), dim3(256), shm, s, p);
    RTP_CHECK_LAUNCH();
    return RTP_OK;
  }
  dim3 grid(nsplit, p.N, (p.co32 / 32) * p.citiles);
  hipLaunchKernelGGL(wgrad_kernel, grid, dim3(256), 0, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
