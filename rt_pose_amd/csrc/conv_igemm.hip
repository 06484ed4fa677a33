// Implicit-GEMM 3-D convolution on MFMA for channels-last bf16 activations (gfx950).
//
// GEMM view: M = output voxels, N = Cout, K = taps * Cin.  One v_mfma_f32_16x16x32_bf16
// contracts the 32 channels of one tap:  A = weights [16 cout][32 cin],  B = activations [32 cin][16 voxels]
// so the accumulator holds, per lane, 4 consecutive output channels of one voxel (an 8-byte bf16 store).
//
// This file holds the GENERIC kernel: any Cin multiple of 32, Cout multiple of 16, ks in {1,3}, stride in
// {1,2}, normal or transposed (data-gradient) indexing; the B operand is gathered straight from global/L2.
// The LDS-tiled kernel for the full-resolution 32->32 layers lives in conv_tiled.hip.
#include "rtp_common.h"
#include "rtp_prof.h"

struct ConvParams {
  const bf16_t* x;
  const bf16_t* w;
  const float* btab;
  const bf16_t* res;
  void* y;
  int N, Di, Hi, Wi, Do, Ho, Wo;
  int Ci, Co;  // padded: Ci % 32 == 0, Co % 16 == 0
  int ks, stride, pad, transposed;
  int x_cs, x_co, y_cs, y_co, r_cs, r_co;
  int relu, y_fp32;
  long w_sample_stride;  // elements between per-sample weight sets (0 = shared)
  int Vo;                // Do*Ho*Wo
  int blocks_per_sample;
};

// NT = number of 16-wide Cout tiles handled by one wave; MT voxel tiles of 16.
template <int NT, int MT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n = blockIdx.x / p.blocks_per_sample;
  const int blk = blockIdx.x - n * p.blocks_per_sample;
  const int co_base = blockIdx.y * (NT * 16);
  const int vbase = (blk * 4 + wave) * (MT * 16);
  if (vbase >= p.Vo) return;  // wave-uniform
  const int lv = lane & 15;   // voxel within tile (B column) / cout within tile (A row)
  const int q = lane >> 4;    // k sub-chunk (8 channels)

  int oz[MT], oy[MT], ox[MT];
  bool vok[MT];
  // Data gradient of a stride-2 conv: only taps with k = (u + pad) mod 2 contribute to output position u, so a tile
  // whose voxels share their parity needs 1-8 of the 27 taps.  Rows already share (z,y) parity; x is enumerated
  // evens-first so that 16-voxel tiles share x parity too, and taps no lane needs are skipped wave-uniformly below.
  const bool parity_x = p.transposed && p.stride == 2;
  const int half_w = (p.Wo + 1) >> 1;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int v = vbase + mt * 16 + lv;
    vok[mt] = v < p.Vo;
    vox_decode(vok[mt] ? v : 0, p.Ho, p.Wo, oz[mt], oy[mt], ox[mt]);
    if (parity_x) ox[mt] = (ox[mt] < half_w) ? 2 * ox[mt] : 2 * (ox[mt] - half_w) + 1;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // MFMA row (nt, 4q+r) <- output channel 8q + 4nt + r (NT == 2): the lane's two accumulator tiles then hold 8
  // contiguous channels = one 16-B chunk, so a store instruction covers whole 64-B voxel lines
  int arow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) arow[nt] = (NT == 2) ? 8 * (lv >> 2) + 4 * nt + (lv & 3) : nt * 16 + lv;

  const bf16_t* wbase = p.w + (long)n * p.w_sample_stride;
  const long x_n = (long)n * p.Di * p.Hi * p.Wi;
  const int kchunks = p.Ci >> 5;

  // Per-axis tap tables, computed once per tile: input coordinate offset (in voxels of the flattened sample) and
  // validity of each of the (up to) 3 taps along z, y, x.  The 27-tap loop then only ANDs three bits and adds three
  // offsets; recomputing coordinates and bounds per tap cost more VALU time than the MFMAs on the low-resolution and
  // stride-2-gradient layers.
  int offz[MT][3], offy[MT][3], offx[MT][3];
  unsigned okm[MT];  // bits 0-2: z taps, 3-5: y taps, 6-8: x taps
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    okm[mt] = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int iz, iy, ix;
      bool vz = k < p.ks, vy = k < p.ks, vx = k < p.ks;
      if (!p.transposed) {
        iz = oz[mt] * p.stride + k - p.pad;
        iy = oy[mt] * p.stride + k - p.pad;
        ix = ox[mt] * p.stride + k - p.pad;
      } else {
        int tz = oz[mt] + p.pad - k, ty = oy[mt] + p.pad - k, tx = ox[mt] + p.pad - k;
        vz = vz && tz >= 0; vy = vy && ty >= 0; vx = vx && tx >= 0;
        if (p.stride == 2) {
          vz = vz && !(tz & 1); vy = vy && !(ty & 1); vx = vx && !(tx & 1);
          tz >>= 1; ty >>= 1; tx >>= 1;
        }
        iz = tz; iy = ty; ix = tx;
      }
      vz = vz && (unsigned)iz < (unsigned)p.Di;
      vy = vy && (unsigned)iy < (unsigned)p.Hi;
      vx = vx && (unsigned)ix < (unsigned)p.Wi;
      offz[mt][k] = iz * p.Hi * p.Wi; offy[mt][k] = iy * p.Wi; offx[mt][k] = ix;
      okm[mt] |= ((unsigned)vz << k) | ((unsigned)vy << (3 + k)) | ((unsigned)vx << (6 + k));
    }
    if (!vok[mt]) okm[mt] = 0;
  }

  int tap = 0;
  for (int kz = 0; kz < p.ks; ++kz)
  for (int ky = 0; ky < p.ks; ++ky)
  for (int kx = 0; kx < p.ks; ++kx, ++tap) {
    long xoff[MT];
    bool ok[MT];
    bool any = false;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      ok[mt] = (okm[mt] >> kz) & (okm[mt] >> (3 + ky)) & (okm[mt] >> (6 + kx)) & 1u;
      any |= ok[mt];
      xoff[mt] = (x_n + offz[mt][kz] + offy[mt][ky] + offx[mt][kx]) * p.x_cs + p.x_co + q * 8;
    }
    if (!__any(any)) continue;  // wave-uniform: no lane of this wave has an in-range sample for this tap
    const bf16_t* wt = wbase + ((long)tap * p.Co + co_base) * p.Ci + q * 8;
    for (int kc = 0; kc < kchunks; ++kc) {
      bf16x8 a[NT], b[MT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) a[nt] = ld_bf16x8(wt + (long)arow[nt] * p.Ci + kc * 32);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) b[mt] = ok[mt] ? ld_bf16x8(p.x + xoff[mt] + kc * 32) : zero_bf16x8();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nt], b[mt], acc[mt][nt], 0, 0, 0);
    }
  }

  // epilogue: lane holds couts co_base + nt*16 + 4q .. +3 of voxel (mt, lv)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (!vok[mt]) continue;
    const long vo = (long)n * p.Vo + ((long)oz[mt] * p.Ho + oy[mt]) * p.Wo + ox[mt];
    int cls = 0;
    if (p.btab) cls = vox_class(oz[mt], oy[mt], ox[mt], p.Do, p.Ho, p.Wo);
    constexpr int CH = 4 * NT;
    const int c0 = co_base + q * CH;
    float val[CH];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) val[nt * 4 + j] = acc[mt][nt][j];
    if (p.btab) {
      const float* bp = p.btab + ((long)(p.w_sample_stride ? n : 0) * 64 + cls) * p.Co + c0;
#pragma unroll
      for (int k = 0; k < CH; k += 4) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bp + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) val[k + j] += bb[j];
      }
    }
    if (p.res) {
      const bf16_t* rp = p.res + vo * p.r_cs + p.r_co + c0;
#pragma unroll
      for (int k = 0; k < CH; k += 4) {
        const bf16x4 r = *reinterpret_cast<const bf16x4*>(rp + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) val[k + j] += bf2f(r[j]);
      }
    }
    if (p.relu) {
#pragma unroll
      for (int j = 0; j < CH; ++j) val[j] = val[j] > 0.f ? val[j] : 0.f;
    }
    if (p.y_fp32) {
      float* yp = (float*)p.y + vo * p.y_cs + p.y_co + c0;
#pragma unroll
      for (int k = 0; k < CH; k += 4) *reinterpret_cast<f32x4*>(yp + k) = f32x4{val[k], val[k + 1], val[k + 2], val[k + 3]};
    } else {
      bf16_t* yp = (bf16_t*)p.y + vo * p.y_cs + p.y_co + c0;
      if constexpr (NT == 2) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(val[j]);
        st_bf16x8(yp, o);
      } else {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(val[j]);
        *reinterpret_cast<bf16x4*>(yp) = o;
      }
    }
  }
}

int rtp_conv_tiled_try(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                       const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32,
                       const RtpAct* stat_x, float* stat_out, hipStream_t s);
int rtp_conv_tiled_stat_slots(const RtpAct* x, const RtpConvGeom* g, int transposed);

extern "C" int rtp_conv_stats_nsplit(const RtpAct* x, const RtpConvGeom* g, int transposed) {
  if (!x || !g) return 0;
  return rtp_conv_tiled_stat_slots(x, g, transposed);
}

static int conv_dispatch(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                         const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32,
                         const RtpAct* stat_x, float* stat_out, void* stream);

extern "C" int rtp_conv_igemm(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                              const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32,
                              void* stream) {
  return conv_dispatch(x, wf, w_per_sample, btab, res, y, g, relu, transposed, y_fp32, nullptr, nullptr, stream);
}

extern "C" int rtp_conv_igemm_stats(const RtpAct* x, const void* wf, int w_per_sample, const float* btab,
                                    const RtpAct* res, const RtpAct* y, const RtpConvGeom* g, int relu, int transposed,
                                    int y_fp32, const RtpAct* stat_x, float* stat_out, void* stream) {
  if (!stat_out) return RTP_ERR_SHAPE;
  if (stat_x && ((stat_x->cs % 8) || (stat_x->co % 8))) return RTP_ERR_ALIGN;
  return conv_dispatch(x, wf, w_per_sample, btab, res, y, g, relu, transposed, y_fp32, stat_x, stat_out, stream);
}

static int conv_dispatch(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                         const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32,
                         const RtpAct* stat_x, float* stat_out, void* stream) {
  if (!x || !y || !wf || !g) return RTP_ERR_SHAPE;
  if (g->ks != 1 && g->ks != 3) return RTP_ERR_UNSUPPORTED;
  if (g->stride != 1 && g->stride != 2) return RTP_ERR_UNSUPPORTED;
  if ((x->co % 8) || (x->cs % 8) || (y->co % 8) || (y->cs % 8)) return RTP_ERR_ALIGN;
  if (res && ((res->co % 4) || (res->cs % 4))) return RTP_ERR_ALIGN;
  {
    const int rc = rtp_conv_tiled_try(x, wf, w_per_sample, btab, res, y, g, relu, transposed, y_fp32, stat_x, stat_out,
                                      (hipStream_t)stream);
    if (rc <= 0) return rc;  // handled (or failed) by the LDS-tiled kernel
  }
  if (stat_out) return RTP_ERR_UNSUPPORTED;  // only geometries with rtp_conv_stats_nsplit() > 0 emit statistics
  ConvParams p;
  p.x = (const bf16_t*)x->ptr;
  p.w = (const bf16_t*)wf;
  p.btab = btab;
  p.res = res ? (const bf16_t*)res->ptr : nullptr;
  p.y = y->ptr;
  p.N = g->n;
  p.ks = g->ks; p.stride = g->stride; p.pad = g->pad; p.transposed = transposed;
  if (!transposed) {
    p.Di = g->di; p.Hi = g->hi; p.Wi = g->wi; p.Do = g->dov; p.Ho = g->ho; p.Wo = g->wo;
    p.Ci = g->ci; p.Co = g->co;
  } else {  // gradient flows output-side -> input-side; contraction over the forward conv's (padded) Cout
    p.Di = g->dov; p.Hi = g->ho; p.Wi = g->wo; p.Do = g->di; p.Ho = g->hi; p.Wo = g->wi;
    p.Ci = (g->co + 31) / 32 * 32; p.Co = g->ci;
  }
  if (p.Ci % 32 || p.Co % 16) return RTP_ERR_UNSUPPORTED;
  if (x->c < p.Ci || y->c < p.Co) return RTP_ERR_SHAPE;
  if ((x->co % 8) || (x->cs % 8) || (y->co % 4) || (y->cs % 4)) return RTP_ERR_ALIGN;
  if (res && ((res->co % 4) || (res->cs % 4))) return RTP_ERR_ALIGN;
  p.x_cs = x->cs; p.x_co = x->co; p.y_cs = y->cs; p.y_co = y->co;
  p.r_cs = res ? res->cs : 0; p.r_co = res ? res->co : 0;
  p.relu = relu; p.y_fp32 = y_fp32;
  const int ntap = p.ks * p.ks * p.ks;
  p.w_sample_stride = w_per_sample ? (long)ntap * p.Co * p.Ci : 0;
  p.Vo = p.Do * p.Ho * p.Wo;
  hipStream_t s = (hipStream_t)stream;
  const int nt = (p.Co % 32 == 0) ? 2 : 1;
  // 64 voxels per wave amortise the weight fragments; small (low-resolution) problems instead take 16 voxels per wave
  // so that they still spread over the chip (they are latency-, not throughput-bound)
  const bool small = (long)p.N * p.Vo * (p.Co / (16 * nt)) < 256L * 256 * 2;
  const int mt = small ? 1 : 4;
  p.blocks_per_sample = rtp_div_up(p.Vo, 4 * mt * 16);
  dim3 grid(p.N * p.blocks_per_sample, p.Co / (16 * nt));
  RtpProfScope prof(RTP_FAM_CONV, s);
  if (nt == 2 && !small) hipLaunchKernelGGL((conv_igemm_kernel<2, 4>), grid, dim3(256), 0, s, p);
  else if (nt == 2) hipLaunchKernelGGL((conv_igemm_kernel<2, 1>), grid, dim3(256), 0, s, p);
  else if (!small) hipLaunchKernelGGL((conv_igemm_kernel<1, 4>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((conv_igemm_kernel<1, 1>), grid, dim3(256), 0, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
