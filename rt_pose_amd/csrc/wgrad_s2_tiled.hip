// LDS-tiled weight-gradient correlation of the 3x3x3 STRIDE-2 convs (32 input channels x 32 output channels per launch):
//
//   G[tap][co][ci] = sum_o gy[o][co] * x[2 o + tap - 1][ci]
//
// the weight-gradient twin of conv_s2_tiled.hip (hr_util/hr3d.py:162-197, 297-305) and, like it, an HBM-bound stream of x.
// Same workgroup shape: waves 4-7 stage, two bricks ahead, the haloed input of a brick of 1 x 2 x 16 output voxels (3 planes x
// 5 rows x [17 odd-x | 16 even-x] voxels, parity separated at staging time so that a tap's 16 voxels are consecutive) plus
// the brick's 32 gy voxels into a ring of three 36-KB slots by LDS-DMA; waves 0-3 contract over the brick's 32 voxels (one
// MFMA k-step) with both operands read TRANSPOSED from the voxel-major images (ds_read_b64_tr_b16), the 27 taps dealt
// round-robin to the four waves, 7 x 4 accumulator tiles per wave kept in registers across all bricks of the workgroup.
// One fp32 slab [27][32][32] per workgroup (a window of wider slabs when the conv runs as channel slices), folded by
// rtp_wgrad_fold like every other slab -- no atomics, fixed summation order.
#include <stdlib.h>

#include "rtp_common.h"
#include "rtp_prof.h"

#define W2_OY 2
#define W2_OX 16
#define W2_ROWS (2 * W2_OY + 1)            // 5
#define W2_POS (2 * W2_OX + 1)             // 33
#define W2_PLANE_VOX (W2_ROWS * W2_POS)    // 165
#define W2_BRICK_VOX (3 * W2_PLANE_VOX)    // 495
#define W2_XITEMS (W2_BRICK_VOX * 4)       // 1980
#define W2_XPIECES 32                      // x: 32 one-KB pieces (the last one carries 60 live lanes)
#define W2_GPIECES 4                       // gy: 32 voxels = 2 pieces, padded to one per loader wave
#define W2_PPW ((W2_XPIECES + W2_GPIECES) / 4)   // 9 pieces per loader wave and brick
#define W2_SLOT ((W2_XPIECES + W2_GPIECES) * 64 * 8)   // bf16 elements per ring slot (36 KB)
#define W2_GOFF (W2_XPIECES * 64 * 8)      // element offset of the gy image inside a slot
#define W2_RING 3

__device__ __attribute__((aligned(16))) bf16_t g_zero_line_w2[8];

struct WgS2Params {
  const bf16_t* gy; int g_cs, g_co;
  const bf16_t* x; int x_cs, x_co;
  float* gp; int slab_rows, slab_cols;
  int N, D, H, W, Do, Ho, Wo;
  int tiles_y, tiles_x, bricks_per_sample, wgs_per_sample;
  int part_stride;   // slab slots per sample (>= wgs_per_sample: a width-hinted launch runs on fewer workgroups and leaves the upper slots zero)
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_w2;

__device__ __forceinline__ int w2_rot(int chunk, int xi) { return ((chunk + (xi >> 2)) & 3) << 3; }  // bf16 elements

__device__ __forceinline__ void w2_dma16(const bf16_t* src, unsigned lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(src) : "memory", "m0");
}

__device__ __forceinline__ bf16x8 w2_tr_pair(unsigned lo, unsigned hi) {
  s16x4 l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_w2*)(lo));
  s16x4 h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_w2*)(hi));
  s16x8 r = {l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
  return __builtin_bit_cast(bf16x8, r);
}

__global__ __attribute__((amdgpu_flat_work_group_size(512, 512), amdgpu_waves_per_eu(2, 2))) void wgrad_s2_kernel(WgS2Params p) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = (gridDim.x % 8 == 0) ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int n = bid / p.wgs_per_sample, wg = bid - n * p.wgs_per_sample;
  const int b_begin = (int)((long)wg * p.bricks_per_sample / p.wgs_per_sample);
  const int nb = (int)((long)(wg + 1) * p.bricks_per_sample / p.wgs_per_sample) - b_begin;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) bf16_t*)lds;

  if (wave >= 4) {
    // ================= loaders =================
    const int lw = wave - 4;
    int s_rel[W2_PPW], s_pk[W2_PPW];
#pragma unroll
    for (int k = 0; k < W2_PPW; ++k) {
      const int piece = k * 4 + lw;
      const int item = piece * 64 + lane;
      if (piece < W2_XPIECES) {   // (wave-uniform per k)
        const int cp = item & 3, hv = item >> 2;
        const int pos = hv % W2_POS, row = (hv / W2_POS) % W2_ROWS, pl = hv / W2_PLANE_VOX;
        const int xi = pos < 17 ? pos : pos - 17;
        const int ck = (cp - (xi >> 2)) & 3;
        const int xoff = pos < 17 ? 2 * pos : 2 * (pos - 17) + 1;
        s_rel[k] = ((pl * p.H + row) * p.W + xoff) * p.x_cs + ck * 8;
        s_pk[k] = (pl == 0) | ((row == 0) << 1) | ((xoff == 0) << 2) | ((item >= W2_XITEMS) << 3) | (xoff << 8);
      } else {                    // gy brick: voxel k = r * 16 + v at k * 64 B, chunk rotated by k >> 2
        const int j = item - W2_XPIECES * 64;
        const int cp = j & 3, bv = j >> 2;
        const int r = (bv >> 4) & 1, v = bv & 15;
        const int ck = (cp - (bv >> 2)) & 3;
        s_rel[k] = (r * p.Wo + v) * p.g_cs + ck * 8;
        s_pk[k] = ((bv >= 32) << 3) | (v << 8);
      }
    }
    const bf16_t* xn = p.x + (long)n * p.D * p.H * p.W * p.x_cs + p.x_co;
    const bf16_t* gn = p.gy + (long)n * p.Do * p.Ho * p.Wo * p.g_cs + p.g_co;
    int l_oz = b_begin % p.Do, l_tx = (b_begin / p.Do) % p.tiles_x, l_ty = b_begin / (p.Do * p.tiles_x);
    auto issue = [&](int slot) {
      const int z0 = 2 * l_oz - 1, y0 = 2 * l_ty * W2_OY - 1, x0 = 2 * l_tx * W2_OX - 1;
      const long org = (((long)z0 * p.H + y0) * p.W + x0) * p.x_cs;
      const long gorg = (((long)l_oz * p.Ho + l_ty * W2_OY) * p.Wo + l_tx * W2_OX) * p.g_cs;
      const int tflg = (l_oz == 0) | ((l_ty == 0) << 1) | ((l_tx == 0) << 2) | 8;
      const int xlim = (p.W - x0) << 8;
      const int glim = (p.Wo - l_tx * W2_OX) << 8;
      const unsigned dst = lds0 + 2u * (unsigned)(slot * W2_SLOT);
#pragma unroll
      for (int k = 0; k < W2_PPW; ++k) {
        const bf16_t* src;
        if (k * 4 + lw < W2_XPIECES) {
          const bool oob = (s_pk[k] & tflg & 0xff) || (s_pk[k] >> 8 << 8) >= xlim;
          src = oob ? g_zero_line_w2 : xn + org + s_rel[k];
        } else {
          const bool oob = (s_pk[k] & 8) || (s_pk[k] >> 8 << 8) >= glim;
          src = oob ? g_zero_line_w2 : gn + gorg + s_rel[k];
        }
        w2_dma16(src, dst + 1024u * (unsigned)(k * 4 + lw));
      }
      if (++l_oz == p.Do) { l_oz = 0; if (++l_tx == p.tiles_x) { l_tx = 0; ++l_ty; } }
    };
    int issued = 0;
    if (nb > 0) { issue(0); ++issued; }
    if (nb > 1) { issue(1); ++issued; }
    if (nb > 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(W2_PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = 0; i < nb; ++i) {
      if (issued < nb) {
        issue(issued % W2_RING);
        ++issued;
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(W2_PPW) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }
  // ================= consumers =================
  const int tw = wave;
  f32x4 acc[7][2][2];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per-lane fragment addresses (slot 0; the slot offset is added per brick).  k-step = the brick's 32 voxels, k = r * 16 + v:
  // 16-lane group q covers k = 8q .. 8q + 7 (row q >> 1, v = 8 (q & 1) ..), lo / hi = its first / second four voxels
  unsigned ax[7][2][2], ag[2][2];
  {
    const int q = lane >> 4, i = lane & 15, a = i >> 2, pp = i & 3;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int chunk = 2 * sub + (pp >> 1), within = (pp & 1) * 4;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kk = 8 * q + a + 4 * h;            // voxel of the k-step
        const int r = kk >> 4, v = kk & 15;
        ag[sub][h] = lds0 + 2u * (unsigned)(W2_GOFF + kk * 32 + w2_rot(chunk, kk) + within);
#pragma unroll
        for (int t = 0; t < 7; ++t) {
          int tap = tw + 4 * t;
          if (tap > 26) tap = 26;   // wave 3 has six taps; its seventh slot recomputes tap 26 into registers nobody stores
          const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
          const int pos = (kx == 1 ? 17 : (kx == 2 ? 1 : 0)) + v;
          const int xi = pos < 17 ? pos : pos - 17;
          ax[t][sub][h] = lds0 + 2u * (unsigned)(((kz * W2_ROWS + 2 * r + ky) * W2_POS + pos) * 32 + w2_rot(chunk, xi) + within);
        }
      }
    }
  }
  __builtin_amdgcn_s_barrier();   // brick 0 staged (the loaders waited for it)
  for (int i = 0; i < nb; ++i) {
    const unsigned so = 2u * (unsigned)((i % W2_RING) * W2_SLOT);
    // every fragment of the brick is requested before the first MFMA (4 + 28 transposing reads = 72 VGPRs); the fences keep
    // that order -- left alone hipcc waits for each pair of reads right in front of its MFMAs
    bf16x8 a0 = w2_tr_pair(ag[0][0] + so, ag[0][1] + so), a1 = w2_tr_pair(ag[1][0] + so, ag[1][1] + so);
    bf16x8 b0[7], b1[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      b0[t] = w2_tr_pair(ax[t][0][0] + so, ax[t][0][1] + so);
      b1[t] = w2_tr_pair(ax[t][1][0] + so, ax[t][1][1] + so);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      acc[t][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0[t], acc[t][0][0], 0, 0, 0);
      acc[t][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1[t], acc[t][0][1], 0, 0, 0);
      acc[t][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0[t], acc[t][1][0], 0, 0, 0);
      acc[t][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1[t], acc[t][1][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  // ---- one fp32 slab window per workgroup; D[row = co][col = ci]: lane holds rows 4q..4q+3, column lane & 15
  const int q = lane >> 4, ii = lane & 15;
  float* out = p.gp + ((long)n * p.part_stride + wg) * 27 * p.slab_rows * p.slab_cols;
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int tap = tw + 4 * t;
    if (tap < 27)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            out[(tap * p.slab_rows + a * 16 + q * 4 + r) * p.slab_cols + b * 16 + ii] = acc[t][a][b][r];
  }
}

static bool w2_geometry_ok(const RtpConvGeom* g) {
  static const bool disabled = getenv("RTP_DISABLE_TILED") != nullptr;
  if (disabled || !g) return false;
  if (g->ks != 3 || g->stride != 2 || g->pad != 1) return false;
  if (g->di != 2 * g->dov || g->hi != 2 * g->ho || g->wi != 2 * g->wo) return false;
  if (g->ho % W2_OY || g->dov < 1 || g->ho < 2 || g->wo < 1) return false;
  const int co32 = (g->co + 31) / 32 * 32;
  if (g->ci % 32 || g->ci > 256 || co32 > 256) return false;
  if (g->ci * co32 > 32 * 32) {   // channel slices: every launch streams x again; only for volumes that fill the chip
    static const bool no_slices = false;
    if (no_slices || g->co % 32 || (g->ci > 32 && (long)g->n * g->dov * g->ho * g->wo < 65536)) return false;
  }
  return true;
}

static int w2_wgs(const RtpConvGeom* g) {
  const int bricks = g->dov * (g->ho / W2_OY) * ((g->wo + W2_OX - 1) / W2_OX);
  int wgs = (g->wgs > 0 ? (g->wgs > 256 ? 256 : g->wgs) : 256) / g->n;   // RtpConvGeom::wgs: launch width = number of slabs
  if (wgs < 1) wgs = 1;
  if (wgs > bricks) wgs = bricks;
  return wgs;
}

// slabs per sample (0: not this kernel's geometry)
int rtp_wgrad_s2_nsplit(const RtpConvGeom* g) { return w2_geometry_ok(g) ? w2_wgs(g) : 0; }

// RTP_OK if handled, +1 if not this kernel's geometry, negative on error
int rtp_wgrad_s2_try(const RtpAct* gy, const RtpAct* x, const RtpConvGeom* g, int nsplit, float* gp, hipStream_t s, int slab_rows,
                     int slab_cols) {
  if (!w2_geometry_ok(g)) return 1;
  if (x->cs % 32 || x->co % 8 || gy->cs % 32 || gy->co % 8 || nsplit != w2_wgs(g)) return 1;
  const int K = g->ci / 32, J = (g->co + 31) / 32;
  if (K * J > 1) {
    RtpConvGeom gs = *g;
    gs.ci = 32; gs.co = 32; gs.w_ci_total = 0; gs.w_ci_off = 0;
    for (int j = 0; j < J; ++j)
      for (int k = 0; k < K; ++k) {
        RtpAct gj = *gy; gj.co = gy->co + 32 * j; gj.c = 32;
        RtpAct xk = *x; xk.co = x->co + 32 * k; xk.c = 32;
        const int rc = rtp_wgrad_s2_try(&gj, &xk, &gs, nsplit, gp + (long)(32 * j) * g->ci + 32 * k, s, 32 * J, g->ci);
        if (rc != RTP_OK) return rc > 0 ? RTP_ERR_UNSUPPORTED : rc;
      }
    return RTP_OK;
  }
  WgS2Params p;
  p.gy = (const bf16_t*)gy->ptr; p.g_cs = gy->cs; p.g_co = gy->co;
  p.x = (const bf16_t*)x->ptr; p.x_cs = x->cs; p.x_co = x->co;
  p.gp = gp; p.slab_rows = slab_rows > 0 ? slab_rows : 32; p.slab_cols = slab_cols > 0 ? slab_cols : 32;
  p.N = g->n; p.D = g->di; p.H = g->hi; p.W = g->wi; p.Do = g->dov; p.Ho = g->ho; p.Wo = g->wo;
  p.tiles_y = p.Ho / W2_OY; p.tiles_x = (p.Wo + W2_OX - 1) / W2_OX;
  p.bricks_per_sample = p.Do * p.tiles_y * p.tiles_x;
  p.wgs_per_sample = nsplit; p.part_stride = nsplit;
  const size_t shm = sizeof(bf16_t) * (size_t)W2_RING * W2_SLOT;
  RtpProfScope prof(RTP_FAM_WGRAD_TILED, s);
  static bool attr[RTP_MAX_DEVICES] = {};
  if (rtp_once_per_device(attr)) {
    (void)hipFuncSetAttribute((const void*)wgrad_s2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  }
  hipLaunchKernelGGL(wgrad_s2_kernel, dim3(p.N * p.wgs_per_sample), dim3(512), shm, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
