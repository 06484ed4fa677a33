// LDS-tiled FORWARD 3x3x3 STRIDE-2 convolution, 32 input channels -> 32 output channels (per launch): the transition convs and
// the down-sampling chains of the fuse rows (hr_util/hr3d.py:162-197, 297-305) -- convs that read a tensor of 8x the voxels
// they write, i.e. HBM-bound streams of x (96 algorithmic flop/B against a ridge of 312).
//
//   y[o][co] = sum_{tap, ci} W[co][ci][tap] * x[2 o + tap - 1][ci]          (per axis; zero padding)
//
// One persistent 8-wave workgroup per CU.  Waves 4-7 are LOADERS: they stage the haloed input of a brick of 1(z) x 2(y) x
// 16(x) output voxels -- 3 planes x 5 rows x 33 voxels x 64 B = 31.7 KB -- by LDS-DMA (global_load_lds, 16 B per lane, no
// VGPRs for the data) into a ring of three 32-KB slots, two bricks ahead of waves 0-3, which run the MFMAs
// (v_mfma_f32_16x16x32_bf16: A = weights [16 co][32 ci], B = voxels [32 ci][16]) and the epilogue.  One raw s_barrier per
// brick; the loaders wait with a COUNTED vmcnt, so the next brick's transfers stay in flight across the barrier.
//   * the sample's (folded) weights [27][32][32] bf16 stay in LDS for the workgroup's lifetime (55 KB);
//   * x parity is separated AT STAGING TIME: a staged row holds the 17 odd-x voxels (2 ox0 - 1, ..., 2 ox0 + 31), then the 16
//     even-x ones, so the 16 voxels a tap needs (x = 2 ox + kx - 1) are CONSECUTIVE in LDS -- with the interleaved order the
//     lanes of a ds_read_b128 group hit voxels 128 B apart, at best a 2-way bank conflict -- and the chunk rotation of
//     conv_tiled.hip (chunk' = (chunk + 2 (i >> 2)) & 3, applied on the DMA's SOURCE address) makes every read conflict-free;
//   * bricks are dealt z-fastest in contiguous runs per workgroup, workgroups permuted so that every XCD owns one contiguous
//     run: the plane two successive bricks share (2 oz + 1) and the halo rows / columns of y / x neighbours are L2 hits;
//   * epilogue: per-boundary-class bias (GroupNorm fold), optional fp32 partial sum in / fp32 out (input-channel slices of a
//     wider conv), ReLU, bf16 store, and the (sum y, sum y^2) statistics of the stored tensor, one partial per workgroup.
// Wave w of the four compute waves owns output row (w >> 1) of the brick and output-channel tile (w & 1): 27 MFMAs per brick
// against ~3 000 cycles of HBM time for the brick's bytes -- the matrix pipe idles by design.
#include <stdlib.h>

#include "rtp_common.h"
#include "rtp_prof.h"

#define S2F_OY 2
#define S2F_OX 16
#define S2F_ROWS (2 * S2F_OY + 1)            // 5
#define S2F_POS (2 * S2F_OX + 1)             // 33: 17 odd-x voxels, then 16 even-x ones
#define S2F_PLANE_VOX (S2F_ROWS * S2F_POS)   // 165
#define S2F_BRICK_VOX (3 * S2F_PLANE_VOX)    // 495
#define S2F_ITEMS (S2F_BRICK_VOX * 4)        // 1980 sixteen-byte items
#define S2F_PIECES 32                        // 1-KB DMA pieces per brick (the last one carries 60 live lanes)
#define S2F_SLOT (S2F_PIECES * 64 * 8)       // bf16 elements per ring slot (32 KB)
#define S2F_RING 3
#define S2F_PPW (S2F_PIECES / 4)             // pieces per loader wave and brick

__device__ __attribute__((aligned(16))) bf16_t g_zero_line_s2f[8];

struct S2FwdParams {
  const bf16_t* x; int x_cs, x_co;
  const bf16_t* w; long w_sample_stride; int w_tap_stride, w_row_stride, w_per_sample;
  const float* btab; int bt_cs;
  void* y; int y_cs, y_co, y_fp32;
  const float* acc32; int a_cs;
  const bf16_t* res; int r_cs, r_co;   // optional residual added before the ReLU
  float* stat_out; int st_cs;
  int N, D, H, W, Do, Ho, Wo;     // input dims; output dims (D == 2 Do, H == 2 Ho, W == 2 Wo)
  int tiles_y, tiles_x, bricks_per_sample, wgs_per_sample;
  int relu;
};

__device__ __forceinline__ int s2f_swz(int chunk, int xi) { return ((chunk + 2 * (xi >> 2)) & 3) << 3; }  // bf16 elements

// One LDS-DMA piece from inline assembly: the compiler does not model it, so it neither drains vmcnt before unrelated LDS reads
// nor before the barrier; the loaders count their own pieces (s_waitcnt vmcnt(N)).
__device__ __forceinline__ void s2f_dma16(const bf16_t* src, unsigned lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(src) : "memory", "m0");
}

template <bool HAS_BTAB, bool STAT>
__global__ __launch_bounds__(512, 2) void conv_s2_fwd_kernel(S2FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  bf16_t* wL = lds;                                   // [27][32][32]
  bf16_t* xL = lds + 27 * 32 * 32;                    // [RING][S2F_SLOT]
  float* bL = reinterpret_cast<float*>(xL + S2F_RING * S2F_SLOT);   // [27][32] class bias, then [4][32][2] statistics scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = (gridDim.x % 8 == 0) ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int n = bid / p.wgs_per_sample, wg = bid - n * p.wgs_per_sample;
  const int v = lane & 15, q = lane >> 4;

  // ---- weights -> LDS (rows permuted as in conv_tiled.hip: MFMA row (nt, 4q + r) <- output channel 8q + 4nt + r)
  {
    const bf16_t* wsrc = p.w + (p.w_per_sample ? (long)n * p.w_sample_stride : 0);
    for (int i = tid; i < 27 * 32 * 4; i += 512) {
      const int ck = i & 3, row = i >> 2, tap = row >> 5, co = row & 31;
      const int arow = ((co >> 2) & 1) * 16 + (co >> 3) * 4 + (co & 3);
      st_bf16x8(wL + (tap * 32 + arow) * 32 + s2f_swz(ck, arow), ld_bf16x8(wsrc + (long)tap * p.w_tap_stride + (long)co * p.w_row_stride + ck * 8));
    }
  }
  if (HAS_BTAB) {
    for (int i = tid; i < 27 * 32; i += 512) {
      const int co = i & 31, k = i >> 5;
      const int cz = k / 9, cy = (k / 3) % 3, cx = k % 3;  // 0 interior, 1 first, 2 last
      const int cls = (cz == 1) | ((cz == 2) << 1) | ((cy == 1) << 2) | ((cy == 2) << 3) | ((cx == 1) << 4) | ((cx == 2) << 5);
      bL[i] = p.btab[((long)(p.w_per_sample ? n : 0) * 64 + cls) * p.bt_cs + co];
    }
  }
  // this workgroup's contiguous run of bricks (z fastest)
  const int b_begin = (int)((long)wg * p.bricks_per_sample / p.wgs_per_sample);
  const int nb = (int)((long)(wg + 1) * p.bricks_per_sample / p.wgs_per_sample) - b_begin;
  const long xvox_n = (long)n * p.D * p.H * p.W;
  const long yvox_n = (long)n * p.Do * p.Ho * p.Wo;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) bf16_t*)lds;
  const unsigned x_base = lds0 + 2u * (unsigned)(xL - lds);

  if (wave >= 4) {
    // ================= loaders =================
    const int lw = wave - 4;
    // brick-independent descriptors of this lane's pieces: element offset relative to the brick's input origin
    // (2 oz - 1, 2 oy0 - 1, 2 ox0 - 1), the x offset (for the right-edge test) and the low-face bits
    int s_rel[S2F_PPW], s_pk[S2F_PPW];
#pragma unroll
    for (int k = 0; k < S2F_PPW; ++k) {
      const int item = (k * 4 + lw) * 64 + lane;
      const int cp = item & 3, hv = item >> 2;
      const int pos = hv % S2F_POS, row = (hv / S2F_POS) % S2F_ROWS, pl = hv / S2F_PLANE_VOX;
      const int xi = pos < 17 ? pos : pos - 17;
      const int ck = (cp - 2 * (xi >> 2)) & 3;
      const int xoff = pos < 17 ? 2 * pos : 2 * (pos - 17) + 1;
      s_rel[k] = ((pl * p.H + row) * p.W + xoff) * p.x_cs + ck * 8;
      s_pk[k] = (pl == 0) | ((row == 0) << 1) | ((xoff == 0) << 2) | ((item >= S2F_ITEMS) << 3) | (xoff << 8);
    }
    const bf16_t* xn = p.x + xvox_n * p.x_cs + p.x_co;
    int l_oz = b_begin % p.Do, l_tx = (b_begin / p.Do) % p.tiles_x, l_ty = b_begin / (p.Do * p.tiles_x);
    auto issue = [&](int slot) {   // stages brick (l_oz, l_ty, l_tx) and advances the coordinates
      const int z0 = 2 * l_oz - 1, y0 = 2 * l_ty * S2F_OY - 1, x0 = 2 * l_tx * S2F_OX - 1;
      const long org = (((long)z0 * p.H + y0) * p.W + x0) * p.x_cs;
      const int tflg = (l_oz == 0) | ((l_ty == 0) << 1) | ((l_tx == 0) << 2) | 8;
      const int xlim = (p.W - x0) << 8;   // x offsets >= this lie beyond the volume (Wo need not be a multiple of 16)
      const unsigned dst = x_base + 2u * (unsigned)(slot * S2F_SLOT);
#pragma unroll
      for (int k = 0; k < S2F_PPW; ++k) {
        const bool oob = (s_pk[k] & tflg & 0xff) || (s_pk[k] >> 8 << 8) >= xlim;
        const bf16_t* src = oob ? g_zero_line_s2f : xn + org + s_rel[k];
        s2f_dma16(src, dst + 1024u * (unsigned)(k * 4 + lw));
      }
      if (++l_oz == p.Do) { l_oz = 0; if (++l_tx == p.tiles_x) { l_tx = 0; ++l_ty; } }
    };
    int issued = 0;
    if (nb > 0) { issue(0); ++issued; }
    if (nb > 1) { issue(1); ++issued; }
    if (nb > 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(S2F_PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // weights, bias table and brick 0 are in place  (the compute waves arrive at the same barrier)
    for (int i = 0; i < nb; ++i) {
      if (issued < nb) {
        issue(issued % S2F_RING);
        ++issued;
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(S2F_PPW) : "memory");   // brick i + 1 has landed, brick i + 2 stays in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
  } else {
    // ================= compute =================
    const int r = wave >> 1, nt = wave & 1;
    const int c0 = 8 * q + 4 * nt;   // this lane's 4 output channels
    // per-lane LDS byte addresses: A fragment row (nt*16 + v), chunk q; B fragment bases per kx (position + chunk rotation)
    const unsigned a_base = lds0 + 2u * (unsigned)((nt * 16 + v) * 32 + s2f_swz(q, v));
    unsigned b_base[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int pos = (kx == 1 ? 17 : (kx == 2 ? 1 : 0)) + v;
      const int xi = pos < 17 ? pos : pos - 17;
      b_base[kx] = 2u * (unsigned)(((2 * r) * S2F_POS + pos) * 32 + s2f_swz(q, xi));
    }
    typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
    int c_oz = b_begin % p.Do, c_tx = (b_begin / p.Do) % p.tiles_x, c_ty = b_begin / (p.Do * p.tiles_x);
    float st_p[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int i = 0; i < nb; ++i) {
      const unsigned sb = x_base + 2u * (unsigned)((i % S2F_RING) * S2F_SLOT);
      const int oz = c_oz, oy = c_ty * S2F_OY + r, ox = c_tx * S2F_OX + v;
      if (++c_oz == p.Do) { c_oz = 0; if (++c_tx == p.tiles_x) { c_tx = 0; ++c_ty; } }
      const bool valid = ox < p.Wo;
      const long vo = yvox_n + ((long)oz * p.Ho + oy) * p.Wo + ox;
      // the partial sum of earlier input-channel slices is requested first: it arrives under the MFMAs
      f32x4 pre = {0.f, 0.f, 0.f, 0.f};
      if (p.acc32 && valid) pre = *reinterpret_cast<const f32x4*>(p.acc32 + vo * p.a_cs + c0);
      if (p.res && valid) {
        const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(p.res + vo * p.r_cs + p.r_co + c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) pre[j] += bf2f(r4[j]);
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      // the 27 (weight, voxel) fragment pairs in three groups of nine (one input plane each), double-buffered in registers: group
      // g + 1 is requested before group g's MFMAs, and the fences keep that order (left alone, hipcc issues every ds_read right
      // before its MFMA and waits lgkmcnt(0) on it: 27 exposed LDS latencies per brick, more than the brick's HBM time)
      bf16x8 fa[2][9], fb[2][9];
      auto load_group = [&](int g, bf16x8 (&a)[9], bf16x8 (&b)[9]) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int ky = t / 3, kx = t % 3;
          a[t] = *(lds_frag)(a_base + 2u * (unsigned)((g * 9 + t) * 32 * 32));
          b[t] = *(lds_frag)(sb + b_base[kx] + 2u * (unsigned)((g * S2F_ROWS + ky) * S2F_POS * 32));
        }
      };
      load_group(0, fa[0], fb[0]);
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        if (g + 1 < 3) load_group(g + 1, fa[(g + 1) & 1], fb[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[g & 1][t], fb[g & 1][t], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      float ev[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) ev[j] = acc[j] + pre[j];
      if (HAS_BTAB) {
        const int k = ((oz == 0) ? 1 : (oz == p.Do - 1) ? 2 : 0) * 9 + ((oy == 0) ? 1 : (oy == p.Ho - 1) ? 2 : 0) * 3 +
                      ((ox == 0) ? 1 : (ox == p.Wo - 1) ? 2 : 0);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bL + k * 32 + c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) ev[j] += bb[j];
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ev[j] = ev[j] > 0.f ? ev[j] : 0.f;
      }
      if (valid) {
        if (p.y_fp32) {
          *reinterpret_cast<f32x4*>((float*)p.y + vo * p.y_cs + p.y_co + c0) = f32x4{ev[0], ev[1], ev[2], ev[3]};
        } else {
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = f2bf(ev[j]);
          *reinterpret_cast<bf16x4*>((bf16_t*)p.y + vo * p.y_cs + p.y_co + c0) = o;
          if (STAT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float rr = bf2f(o[j]); st_p[j] += rr; st_q[j] += rr * rr; }
          }
        }
      }
      __builtin_amdgcn_s_barrier();
    }
    if (STAT) {   // fold the 16 voxel lanes; the two rows' waves are added below
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          st_p[j] += __shfl_xor(st_p[j], o, 64);
          st_q[j] += __shfl_xor(st_q[j], o, 64);
        }
      float* red = bL + 27 * 32;   // [2 rows][32 ch][2]
      if (v == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[(r * 32 + c0 + j) * 2] = st_p[j]; red[(r * 32 + c0 + j) * 2 + 1] = st_q[j]; }
      }
    }
  }
  if (STAT) {
    __syncthreads();
    const float* red = bL + 27 * 32;
    if (tid < 64) p.stat_out[((long)bid * p.st_cs) * 2 + tid] = red[tid] + red[64 + tid];   // tid = channel * 2 + which
  }
}

static bool s2f_geometry_ok(const RtpAct* x, const RtpConvGeom* g) {
  static const bool disabled = getenv("RTP_DISABLE_TILED") != nullptr;
  if (disabled || !x || !g) return false;
  if (g->ks != 3 || g->stride != 2 || g->pad != 1 || g->ci != 32 || g->co != 32) return false;
  if (g->di != 2 * g->dov || g->hi != 2 * g->ho || g->wi != 2 * g->wo) return false;
  if (g->ho % S2F_OY || g->dov < 2 || g->ho < 2 || g->wo < 2) return false;
  if (x->cs % 32 || x->co % 8) return false;
  return true;
}

static int s2f_wgs_per_sample(const RtpConvGeom* g) {
  const int bricks = g->dov * (g->ho / S2F_OY) * ((g->wo + S2F_OX - 1) / S2F_OX);
  int wgs = 256 / g->n;
  if (wgs < 1) wgs = 1;
  if (wgs > bricks) wgs = bricks;
  return wgs;
}

// Statistics partials per sample of the fused epilogue (0: not this kernel's geometry)
int rtp_conv_s2_fwd_stat_slots(const RtpAct* x, const RtpConvGeom* g) { return s2f_geometry_ok(x, g) ? s2f_wgs_per_sample(g) : 0; }

struct TiledSlice { long w_sample_stride; int w_tap_stride, w_row_stride, bt_cs, st_cs; };   // conv_tiled.hip

// RTP_OK if handled, +1 if the geometry is not this kernel's, negative on error.
int rtp_conv_s2_fwd_try(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res, const RtpAct* y,
                        const RtpConvGeom* g, int relu, int y_fp32, float* stat_out, const float* acc32, int acc_cs, hipStream_t s,
                        const TiledSlice* slice) {
  if (!s2f_geometry_ok(x, g)) return 1;
  if (!wf || !y) return RTP_ERR_SHAPE;
  if (res && ((res->cs % 4) || (res->co % 4) || res->c < 32)) return RTP_ERR_ALIGN;
  if (stat_out && y_fp32) return RTP_ERR_UNSUPPORTED;
  if (acc32 && (acc_cs % 4 || acc_cs < 32)) return RTP_ERR_ALIGN;
  if ((y->cs % 4) || (y->co % 4)) return RTP_ERR_ALIGN;
  S2FwdParams p;
  p.x = (const bf16_t*)x->ptr; p.x_cs = x->cs; p.x_co = x->co;
  p.w = (const bf16_t*)wf; p.w_per_sample = w_per_sample;
  p.w_sample_stride = 27L * 32 * 32; p.w_tap_stride = 32 * 32; p.w_row_stride = 32; p.bt_cs = 32; p.st_cs = 32;
  if (slice) {
    p.w_sample_stride = slice->w_sample_stride; p.w_tap_stride = slice->w_tap_stride; p.w_row_stride = slice->w_row_stride;
    p.bt_cs = slice->bt_cs; p.st_cs = slice->st_cs;
    if (p.w_row_stride % 8 || p.w_tap_stride % 8 || p.w_sample_stride % 8) return RTP_ERR_ALIGN;
  }
  p.btab = btab; p.y = y->ptr; p.y_cs = y->cs; p.y_co = y->co; p.y_fp32 = y_fp32;
  p.acc32 = acc32; p.a_cs = acc_cs; p.stat_out = stat_out;
  p.res = res ? (const bf16_t*)res->ptr : nullptr; p.r_cs = res ? res->cs : 0; p.r_co = res ? res->co : 0;
  p.N = g->n; p.D = g->di; p.H = g->hi; p.W = g->wi; p.Do = g->dov; p.Ho = g->ho; p.Wo = g->wo;
  p.tiles_y = p.Ho / S2F_OY; p.tiles_x = (p.Wo + S2F_OX - 1) / S2F_OX;
  p.bricks_per_sample = p.Do * p.tiles_y * p.tiles_x;
  p.wgs_per_sample = s2f_wgs_per_sample(g);
  p.relu = relu;
  const size_t shm = sizeof(bf16_t) * (27 * 32 * 32 + (size_t)S2F_RING * S2F_SLOT) + sizeof(float) * (27 * 32 + 2 * 32 * 2) + 16;
  RtpProfScope prof(RTP_FAM_CONV_TILED, s);
  using Kern = void (*)(S2FwdParams);
  static const Kern tab[2][2] = {{conv_s2_fwd_kernel<false, false>, conv_s2_fwd_kernel<false, true>},
                                 {conv_s2_fwd_kernel<true, false>, conv_s2_fwd_kernel<true, true>}};
  static bool attr[RTP_MAX_DEVICES] = {};
  if (rtp_once_per_device(attr)) {
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) (void)hipFuncSetAttribute((const void*)tab[a][b], hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  }
  hipLaunchKernelGGL(tab[btab ? 1 : 0][stat_out ? 1 : 0], dim3(p.N * p.wgs_per_sample), dim3(512), shm, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
