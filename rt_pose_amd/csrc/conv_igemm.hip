// Implicit-GEMM 3-D convolution on MFMA for channels-last bf16 activations (gfx950).
//
// GEMM view: M = output voxels, N = Cout, K = taps * Cin.  One v_mfma_f32_16x16x32_bf16
// contracts the 32 channels of one tap:  A = weights [16 cout][32 cin],  B = activations [32 cin][16 voxels]
// so the accumulator holds, per lane, 4 consecutive output channels of one voxel (an 8-byte bf16 store).
//
// This file holds the GENERIC kernel: any Cin multiple of 32, Cout multiple of 16, ks in {1,3}, stride in
// {1,2}, normal or transposed (data-gradient) indexing; the B operand is gathered straight from global/L2.
// The LDS-tiled kernel for the full-resolution 32->32 layers lives in conv_tiled.hip.
#include <stdlib.h>

#include "rtp_common.h"
#include "rtp_multi.h"
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
  // optional per-channel statistics of the stored output (one partial per block): sum y and sum y*y, or, with stat_x
  // (same shape as y), sum y*stat_x -- what rtp_chan_stats would compute in a separate read pass
  const bf16_t* stat_x; int s_cs, s_co; float* stat_out;
};

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// One MFMA operand fragment = 16 rows (voxels or output channels) x 64 B.  The MFMA wants lane (row = l & 15, 16-B chunk
// = l >> 4), which makes the four lanes of every load quad hit four different rows: the texture path then spends a
// cycle per LANE (64 per wave load -- the gather kernel measured 50 us for a layer whose bytes need 12).  So the load
// is issued row-major across lanes (lane l: row l >> 2, chunk l & 3: each quad reads 64 contiguous bytes) and the
// registers are moved to the MFMA's lanes with four ds_bpermute (crossbar only, no LDS memory).
__device__ __forceinline__ bf16x8 frag_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff, int src_lane4) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane4, (int)v[i]);
  return __builtin_bit_cast(bf16x8, o);
}

// per-axis coefficient of tap index k in the wave-uniform part of the input offset (see the kernel's addressing note)
__device__ __forceinline__ int tap_coef(int k, int transposed, int stride, int pad) {
  return !transposed ? k - pad : (stride == 2 ? -(k >> 1) : pad - k);
}

// NT = number of 16-wide Cout tiles handled by one wave; MT voxel tiles of 16; KS = kernel size (taps fully unrolled).
//
// Addressing.  The gather kernel was bound by instruction issue, not by memory (PMC: 58 SALU + 34 VALU instructions per
// 2 MFMAs, 51 % issue stalls), so the inner loop carries no address arithmetic at all: both operands are BUFFER loads
// whose address is  descriptor base + per-lane voffset (computed once per tile) + wave-uniform soffset (tap, channel
// chunk).  The input voxel of output voxel o for tap k separates into a per-lane and a uniform part in every mode:
//   conv            : i = o*stride + (k - pad)
//   data grad, s=1  : i = o + (pad - k)
//   data grad, s=2  : i = ((o + pad) >> 1) - (k >> 1)     (valid taps have k = (o + pad) mod 2)
// A lane whose tap falls outside the volume loads with voffset = 0x80000000, which the descriptor's range check turns
// into zeros without touching memory -- the zero padding costs one v_cndmask per tap instead of a data select.
// The descriptor base is moved back by P = Hi*Wi + Wi + 1 voxels so that every uniform part is >= 0.
template <int NT, int MT, int KS>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n = blockIdx.x / p.blocks_per_sample;
  const int blk = blockIdx.x - n * p.blocks_per_sample;
  const int co_base = blockIdx.y * (NT * 16);
  const int vbase = (blk * 4 + wave) * (MT * 16);
  // (a wave past the end of the volume has no valid voxel: every tap is skipped and nothing is stored, but it stays for
  // the statistics barrier below)
  const int lv = lane & 15;   // voxel within tile (B column) / cout within tile (A row)
  const int q = lane >> 4;    // k sub-chunk (8 channels)

  // Two voxel roles per lane: the LOAD role (row l >> 2 of the tile: addressing and tap validity) and the MFMA/epilogue
  // role (column l & 15: which voxel this lane's accumulators belong to).
  const int lrow = lane >> 2, lchunk = lane & 3;
  const int src_lane4 = ((lane & 15) * 4 + (lane >> 4)) * 4;  // ds_bpermute byte address: MFMA lane <- load lane
  int oz[MT], oy[MT], ox[MT], lz_[MT], ly_[MT], lx_[MT];
  bool vok[MT], lok[MT];
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
    // the load role's voxel is column lrow of the same tile: fetch its coordinates from that lane (crossbar) instead of
    // decoding a second index (two integer divisions)
    const int from = lrow * 4;
    lz_[mt] = __builtin_amdgcn_ds_bpermute(from, oz[mt]);
    ly_[mt] = __builtin_amdgcn_ds_bpermute(from, oy[mt]);
    lx_[mt] = __builtin_amdgcn_ds_bpermute(from, ox[mt]);
    lok[mt] = vbase + mt * 16 + lrow < p.Vo;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // MFMA row (nt, 4q+r) <- output channel 8q + 4nt + r (NT == 2): the lane's two accumulator tiles then hold 8
  // contiguous channels = one 16-B chunk, so a store instruction covers whole 64-B voxel lines
  int wvo[NT];  // per-lane byte offset of this lane's weight row fragment
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int arow = (NT == 2) ? 8 * (lrow >> 2) + 4 * nt + (lrow & 3) : nt * 16 + lrow;
    wvo[nt] = (arow * p.Ci + lchunk * 8) * 2;
  }
  const int kchunks = p.Ci >> 5;
  const int P = p.Hi * p.Wi + p.Wi + 1;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.x + ((long)n * p.Di * p.Hi * p.Wi - P) * p.x_cs + p.x_co), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.w + (long)n * p.w_sample_stride + (long)co_base * p.Ci), 0, 0x7fffffff, 0x00020000);

  // per-axis tap validity (bits 0-2: z taps, 3-5: y taps, 6-8: x taps) and the per-lane byte offset, once per tile
  int voff[MT];
  unsigned okm[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    okm[mt] = 0;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      int iz, iy, ix;
      bool vz = true, vy = true, vx = true;
      if (!p.transposed) {
        iz = lz_[mt] * p.stride + k - p.pad;
        iy = ly_[mt] * p.stride + k - p.pad;
        ix = lx_[mt] * p.stride + k - p.pad;
      } else {
        int tz = lz_[mt] + p.pad - k, ty = ly_[mt] + p.pad - k, tx = lx_[mt] + p.pad - k;
        vz = tz >= 0; vy = ty >= 0; vx = tx >= 0;
        if (p.stride == 2) {
          vz = vz && !(tz & 1); vy = vy && !(ty & 1); vx = vx && !(tx & 1);
          tz >>= 1; ty >>= 1; tx >>= 1;
        }
        iz = tz; iy = ty; ix = tx;
      }
      vz = vz && (unsigned)iz < (unsigned)p.Di;
      vy = vy && (unsigned)iy < (unsigned)p.Hi;
      vx = vx && (unsigned)ix < (unsigned)p.Wi;
      okm[mt] |= ((unsigned)vz << k) | ((unsigned)vy << (3 + k)) | ((unsigned)vx << (6 + k));
    }
    if (!lok[mt]) okm[mt] = 0;
    int lz, ly, lx;
    if (!p.transposed) { lz = lz_[mt] * p.stride; ly = ly_[mt] * p.stride; lx = lx_[mt] * p.stride; }
    else if (p.stride == 2) { lz = (lz_[mt] + p.pad) >> 1; ly = (ly_[mt] + p.pad) >> 1; lx = (lx_[mt] + p.pad) >> 1; }
    else { lz = lz_[mt]; ly = ly_[mt]; lx = lx_[mt]; }
    voff[mt] = (((lz * p.Hi + ly) * p.Wi + lx) * p.x_cs + lchunk * 8) * 2;
  }

  // Wave-uniform per-axis masks: a tap can only be needed if each of its three axis taps is valid for SOME lane.  The
  // data gradient of a stride-2 conv (1-8 live taps of 27) and border tiles then skip dead taps on scalar tests alone.
  unsigned axis_any = 0;
  {
    unsigned o = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) o |= okm[mt];
#pragma unroll
    for (int b = 0; b < 9; ++b)
      if (__any((o >> b) & 1u)) axis_any |= 1u << b;
  }
#pragma unroll
  for (int tap = 0; tap < KS * KS * KS; ++tap) {
    constexpr int K2 = KS * KS;
    const int kz = tap / K2, ky = (tap / KS) % KS, kx = tap % KS;  // compile-time after unrolling
    if (!((axis_any >> kz) & (axis_any >> (3 + ky)) & (axis_any >> (6 + kx)) & 1u)) continue;  // scalar
    int vo[MT];
    bool any = false;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bool ok = (okm[mt] >> kz) & (okm[mt] >> (3 + ky)) & (okm[mt] >> (6 + kx)) & 1u;
      any |= ok;
      vo[mt] = ok ? voff[mt] : (int)0x80000000;
    }
    if (!__any(any)) continue;  // wave-uniform: no lane of this wave has an in-range sample for this tap
    const int uz = tap_coef(kz, p.transposed, p.stride, p.pad), uy = tap_coef(ky, p.transposed, p.stride, p.pad),
              ux = tap_coef(kx, p.transposed, p.stride, p.pad);
    int sx = (((uz * p.Hi + uy) * p.Wi + ux + P) * p.x_cs) * 2;
    int sw = (tap * p.Co * p.Ci) * 2;
    for (int kc = 0; kc < kchunks; ++kc, sx += 64, sw += 64) {
      bf16x8 a[NT], b[MT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) a[nt] = frag_ld(rw, wvo[nt], sw, src_lane4);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) b[mt] = frag_ld(rx, vo[mt], sx, src_lane4);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nt], b[mt], acc[mt][nt], 0, 0, 0);
    }
  }

  // epilogue: lane holds couts co_base + nt*16 + 4q .. +3 of voxel (mt, lv)
  float st_p[4 * NT], st_q[4 * NT];
#pragma unroll
  for (int j = 0; j < 4 * NT; ++j) st_p[j] = st_q[j] = 0.f;
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
      float rnd[CH];  // the stored (rounded) values: the statistics are those a read-back pass would see
      if constexpr (NT == 2) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] = f2bf(val[j]); rnd[j] = bf2f(o[j]); }
        st_bf16x8(yp, o);
      } else {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = f2bf(val[j]); rnd[j] = bf2f(o[j]); }
        *reinterpret_cast<bf16x4*>(yp) = o;
      }
      if (p.stat_out) {
        float other[CH];
        if (p.stat_x) {
          const bf16_t* sp = p.stat_x + vo * p.s_cs + p.s_co + c0;
#pragma unroll
          for (int k = 0; k < CH; k += 4) {
            const bf16x4 r = *reinterpret_cast<const bf16x4*>(sp + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) other[k + j] = bf2f(r[j]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < CH; ++j) other[j] = rnd[j];
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) { st_p[j] += rnd[j]; st_q[j] += rnd[j] * other[j]; }
      }
    }
  }
  if (p.stat_out) {  // kernel-uniform
    // fold the 16 voxel lanes of each channel group, then the block's four waves in fixed order: one partial per block
    __shared__ float red[4][NT * 16][2];
    constexpr int CH = 4 * NT;
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        st_p[j] += __shfl_xor(st_p[j], o, 64);
        st_q[j] += __shfl_xor(st_q[j], o, 64);
      }
    if (lv == 0) {
#pragma unroll
      for (int j = 0; j < CH; ++j) { red[wave][q * CH + j][0] = st_p[j]; red[wave][q * CH + j][1] = st_q[j]; }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < NT * 16 * 2) {
      const int ch = t >> 1, which = t & 1;
      const float a = (red[0][ch][which] + red[1][ch][which]) + (red[2][ch][which] + red[3][ch][which]);
      p.stat_out[(((long)n * p.blocks_per_sample + blk) * p.Co + co_base + ch) * 2 + which] = a;
    }
  }
}

int rtp_conv_tiled_try(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                       const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32,
                       const RtpAct* stat_x, float* stat_out, const float* acc32, int acc_cs, hipStream_t s,
                       const struct TiledFuse* fuse = nullptr, const RtpGnFold* fold = nullptr, const struct TiledSlice* slice = nullptr);
struct TiledSlice { long w_sample_stride; int w_tap_stride, w_row_stride, bt_cs, st_cs; };   // conv_tiled.hip
int rtp_conv_tiled_stat_slots(const RtpAct* x, const RtpConvGeom* g, int transposed);
// conv_s2_tiled.hip: the LDS-tiled forward stride-2 conv (32 -> 32 channels per launch)
int rtp_conv_s2_fwd_stat_slots(const RtpAct* x, const RtpConvGeom* g);
int rtp_conv_s2_fwd_try(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res, const RtpAct* y,
                        const RtpConvGeom* g, int relu, int y_fp32, float* stat_out, const float* acc32, int acc_cs, hipStream_t s,
                        const TiledSlice* slice);


// tile shape of the generic kernel for a problem: (cout tiles per wave, voxel tiles per wave, blocks per sample)
static void igemm_shape(int N, int Vo, int Co, int* nt, int* mt, int* bps) {
  static const int force_mt = 0;
  *nt = (Co % 32 == 0) ? 2 : 1;
  // 64 voxels per wave amortise the weight fragments; small (low-resolution) problems instead take 16 voxels per wave
  // so that they still spread over the chip (they are latency-, not throughput-bound)
  const bool small = force_mt ? force_mt == 1 : (long)N * Vo * (Co / (16 * *nt)) < 256L * 256 * 2;
  *mt = small ? 1 : 4;
  *bps = rtp_div_up(Vo, 4 * *mt * 16);
}

// dgrad_s2_tiled.hip: the LDS-tiled data gradient of the stride-2 convs that read a full-resolution 32-channel tensor
struct S2Fuse;
int rtp_dgrad_s2_stat_slots(const RtpAct* gy, const RtpConvGeom* g);
int rtp_dgrad_s2_try(const RtpAct* gy, const void* wd, const RtpAct* dx, const RtpConvGeom* g, const RtpAct* stat_x, float* stat_out,
                     const S2Fuse* fuse, hipStream_t s);
int rtp_conv_tiled_stat_slots(const RtpAct* x, const RtpConvGeom* g, int transposed);
// conv64_tiled.hip: the 64 -> 64-channel stride-1 layers as ONE launch (weights in the waves' registers, no fp32 workspace)
int rtp_conv64_try(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res, const RtpAct* y,
                   const RtpConvGeom* g, int relu, int transposed, int y_fp32, const RtpAct* stat_x, float* stat_out, int wgs,
                   hipStream_t s);
int rtp_conv64_wgs(const RtpAct* x, const RtpConvGeom* g, int transposed);
// ---- wide 3x3x3 stride-1 convs (Cin = 32 K, Cout = 32 J, K * J > 1) as K x J launches of the LDS-tiled 32 -> 32 kernel
// (the 64- and 128-channel layers of the feat64 backbone, hrnet3D_config.py:149-177): output slice j is the sum over the
// input slices k of a 32 x 32 window of the SAME weight image (fold output [n][27][Co][Ci] / data-gradient packing
// [27][ci][cok]); the partial sum travels through the caller's fp32 workspace ws [n][voxels][32] in place, the last slice
// of a chain adds class bias / residual / ReLU, rounds once and emits the statistics partials of its 32 channels.
static void slice_geom(const RtpConvGeom* g, int transposed, RtpConvGeom* gs, int* K, int* J) {
  *gs = *g;
  const int Ci = transposed ? (g->co + 31) / 32 * 32 : g->ci, Co = transposed ? g->ci : g->co;
  *K = Ci / 32; *J = Co / 32;
  gs->ci = 32; gs->co = 32; gs->w_ci_total = 0; gs->w_ci_off = 0;
}

static int sliced_stat_slots(const RtpAct* x, const RtpConvGeom* g, int transposed) {
  static const bool disabled = getenv("RTP_DISABLE_TILED") != nullptr;
  if (disabled || !x || !g) return 0;
  if (g->ks != 3 || g->pad != 1 || (g->stride != 1 && !(g->stride == 2 && !transposed))) return 0;
  const int Ci = transposed ? (g->co + 31) / 32 * 32 : g->ci, Co = transposed ? g->ci : g->co;
  if (Ci % 32 || Co % 32 || Ci * Co <= 32 * 32 || Ci > 256 || Co > 256) return 0;
  if (!transposed && (g->co % 32)) return 0;
  {   // 64 -> 64, stride 1: ONE launch of conv64_tiled.hip with its own width (the slice chain below only without that kernel)
    const int w64 = rtp_conv64_wgs(x, g, transposed);
    if (w64 > 0) return w64;
  }
  RtpConvGeom gs; int K, J;
  slice_geom(g, transposed, &gs, &K, &J);
  RtpAct xs = *x; xs.c = 32;
  // stride 2: every slice launch streams its 32-channel window of x again, which pays for the volumes that fill the chip only
  // (measured at B = 8: 64 -> 64 from [4,16,40], four launches 25 us against 16 on the generic kernel)
  if (g->stride == 2 && K > 1 && (long)g->n * g->dov * g->ho * g->wo < 65536) return 0;
  return g->stride == 2 ? rtp_conv_s2_fwd_stat_slots(&xs, &gs) : rtp_conv_tiled_stat_slots(&xs, &gs, transposed);
}
static bool conv_sliced_geometry(const RtpAct* x, const RtpConvGeom* g, int transposed) { return sliced_stat_slots(x, g, transposed) > 0; }

/* 1 if rtp_conv_igemm_ws runs this conv as slices of the LDS-tiled kernel (a workspace of n * output voxels * 32 floats). */
extern "C" int rtp_conv_sliced_ok(const RtpAct* x, const RtpConvGeom* g, int transposed) {
  return conv_sliced_geometry(x, g, transposed) ? 1 : 0;
}

static int conv_sliced(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                       const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32,
                       const RtpAct* stat_x, float* stat_out, float* ws, hipStream_t s) {
  if (rtp_multi_capture()) return RTP_ERR_UNSUPPORTED;   // (chains of launches cannot be recorded for a shared launch)
  RtpConvGeom gs; int K, J;
  slice_geom(g, transposed, &gs, &K, &J);
  const int Ci = 32 * K, Co = 32 * J;
  if (x->c < Ci || y->c < Co || (res && res->c < Co) || (stat_x && stat_x->c < Co)) return RTP_ERR_SHAPE;
  if (stat_out && y_fp32) return RTP_ERR_UNSUPPORTED;
  if (const int w64 = rtp_conv64_wgs(x, g, transposed)) {   // the native 64 -> 64 kernel; the workspace stays untouched
    const int rc = rtp_conv64_try(x, wf, w_per_sample, btab, res, y, g, relu, transposed, y_fp32, stat_x, stat_out, w64, s);
    if (rc <= 0) return rc;
    if (stat_out) return RTP_ERR_UNSUPPORTED;   // (the slice chain would write another number of partials)
  }
  TiledSlice sl;
  // forward: wf [n][27][Co][Ci] (rows = output channels); transposed: wd [27][conv ci = Co here][cok = Ci here]
  sl.w_row_stride = Ci; sl.w_tap_stride = Co * Ci; sl.w_sample_stride = 27L * Co * Ci; sl.bt_cs = Co; sl.st_cs = Co;
  RtpAct wsa; wsa.ptr = ws; wsa.cs = 32; wsa.co = 0; wsa.c = 32;
  for (int j = 0; j < J; ++j)
    for (int k = 0; k < K; ++k) {
      const bool last = k == K - 1;
      RtpAct xs = *x; xs.co = x->co + 32 * k; xs.c = 32;
      RtpAct ys = *y; ys.co = y->co + 32 * j; ys.c = 32;
      RtpAct rs, sx;
      if (res) { rs = *res; rs.co = res->co + 32 * j; rs.c = 32; }
      if (stat_x) { sx = *stat_x; sx.co = stat_x->co + 32 * j; sx.c = 32; }
      const bf16_t* wjk = (const bf16_t*)wf + (long)(32 * j) * Ci + 32 * k;
      int rc;
      if (g->stride == 2) {
        if (stat_x) return RTP_ERR_UNSUPPORTED;
        rc = rtp_conv_s2_fwd_try(&xs, wjk, w_per_sample, (last && btab) ? btab + 32 * j : nullptr, (last && res) ? &rs : nullptr,
                                 last ? &ys : &wsa, &gs, last ? relu : 0,
                                 last ? y_fp32 : 1, (last && stat_out) ? stat_out + 2 * 32 * j : nullptr, k > 0 ? ws : nullptr, 32, s, &sl);
      } else {
        rc = rtp_conv_tiled_try(&xs, wjk, w_per_sample, (last && btab) ? btab + 32 * j : nullptr, (last && res) ? &rs : nullptr,
                                last ? &ys : &wsa, &gs, last ? relu : 0, transposed, last ? y_fp32 : 1,
                                (last && stat_out && stat_x) ? &sx : nullptr, (last && stat_out) ? stat_out + 2 * 32 * j : nullptr,
                                k > 0 ? ws : nullptr, 32, s, nullptr, nullptr, &sl);
      }
      if (rc != RTP_OK) return rc > 0 ? RTP_ERR_UNSUPPORTED : rc;
    }
  return RTP_OK;
}

/* 1 if this (slice) geometry runs on the LDS-tiled kernel -- the only one that accepts rtp_conv_igemm_acc. */
extern "C" int rtp_conv_tiled_ok(const RtpAct* x, const RtpConvGeom* g, int transposed) {
  return (x && g && rtp_conv_tiled_stat_slots(x, g, transposed) > 0) ? 1 : 0;
}

extern "C" int rtp_conv_stats_nsplit(const RtpAct* x, const RtpConvGeom* g, int transposed) {
  if (!x || !g) return 0;
  const int tiled = rtp_conv_tiled_stat_slots(x, g, transposed);
  if (tiled > 0) return tiled;
  {   // 64 -> 64, stride 1: conv64_tiled.hip (needs no workspace: rtp_conv_igemm_stats and rtp_conv_gn_fused take it too)
    const int w64 = rtp_conv64_wgs(x, g, transposed);
    if (w64 > 0) return w64;
  }
  if (!transposed) {
    const int s2 = rtp_conv_s2_fwd_stat_slots(x, g);
    if (s2 > 0) return s2;
  }

  if (transposed) {
    const int s2 = rtp_dgrad_s2_stat_slots(x, g);
    if (s2 > 0) return s2;
  }
  if ((g->ks != 1 && g->ks != 3) || (g->stride != 1 && g->stride != 2)) return 0;
  const int Co = transposed ? g->ci : g->co;
  const int Vo = transposed ? g->di * g->hi * g->wi : g->dov * g->ho * g->wo;
  if (Co % 16) return 0;
  int nt, mt, bps;
  igemm_shape(g->n, Vo, Co, &nt, &mt, &bps);
  return bps;  // one partial per block of the generic kernel
}

static int conv_dispatch(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                         const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32,
                         const RtpAct* stat_x, float* stat_out, void* stream, const float* acc32 = nullptr, int acc_cs = 0);

/* Statistics partials per sample that rtp_conv_igemm_ws (called WITH a workspace) writes: the sliced route's count where
 * rtp_conv_sliced_ok, rtp_conv_stats_nsplit otherwise. */
extern "C" int rtp_conv_stats_nsplit_ws(const RtpAct* x, const RtpConvGeom* g, int transposed) {
  const int sl = sliced_stat_slots(x, g, transposed);
  return sl > 0 ? sl : rtp_conv_stats_nsplit(x, g, transposed);
}

/* rtp_conv_igemm / rtp_conv_igemm_stats (stat_out may be NULL) with a caller-owned fp32 workspace ws of
 * n * (output voxels) * 32 floats: convs with rtp_conv_sliced_ok(x, g, transposed) run as channel slices of the LDS-tiled
 * kernel; every other geometry ignores ws and takes the usual route. */
extern "C" int rtp_conv_igemm_ws(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                                 const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32,
                                 const RtpAct* stat_x, float* stat_out, float* ws, void* stream) {
  if (!x || !y || !wf || !g) return RTP_ERR_SHAPE;
  if ((x->co % 8) || (x->cs % 8) || (y->co % 8) || (y->cs % 8)) return RTP_ERR_ALIGN;
  if (res && ((res->co % 8) || (res->cs % 8))) return RTP_ERR_ALIGN;
  if (stat_x && ((stat_x->cs % 8) || (stat_x->co % 8))) return RTP_ERR_ALIGN;
  if (stat_x && !stat_out) return RTP_ERR_SHAPE;
  if (ws && conv_sliced_geometry(x, g, transposed))
    return conv_sliced(x, wf, w_per_sample, btab, res, y, g, relu, transposed, y_fp32, stat_x, stat_out, ws, (hipStream_t)stream);
  return conv_dispatch(x, wf, w_per_sample, btab, res, y, g, relu, transposed, y_fp32, stat_x, stat_out, stream);
}

extern "C" int rtp_conv_igemm_acc(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                                  const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32,
                                  const float* acc32, int acc_cs, void* stream) {
  if (!acc32) return RTP_ERR_SHAPE;
  return conv_dispatch(x, wf, w_per_sample, btab, res, y, g, relu, transposed, y_fp32, nullptr, nullptr, stream, acc32, acc_cs);
}

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
                         const RtpAct* stat_x, float* stat_out, void* stream, const float* acc32, int acc_cs) {
  if (!x || !y || !wf || !g) return RTP_ERR_SHAPE;
  if (g->ks != 1 && g->ks != 3) return RTP_ERR_UNSUPPORTED;
  if (g->stride != 1 && g->stride != 2) return RTP_ERR_UNSUPPORTED;
  if ((x->co % 8) || (x->cs % 8) || (y->co % 8) || (y->cs % 8)) return RTP_ERR_ALIGN;
  if (res && ((res->co % 4) || (res->cs % 4))) return RTP_ERR_ALIGN;
  {
    const int rc = rtp_conv_tiled_try(x, wf, w_per_sample, btab, res, y, g, relu, transposed, y_fp32, stat_x, stat_out,
                                      acc32, acc_cs, (hipStream_t)stream);
    if (rc <= 0) return rc;  // handled (or failed) by the LDS-tiled kernel
  }
  if (rtp_multi_capture()) return RTP_ERR_UNSUPPORTED;   // only the stride-1 LDS-tiled kernel can be recorded for a shared launch
  if (!acc32) {
    if (const int w64 = rtp_conv64_wgs(x, g, transposed)) {   // 64 -> 64, stride 1: one launch of conv64_tiled.hip
      const int rc = rtp_conv64_try(x, wf, w_per_sample, btab, res, y, g, relu, transposed, y_fp32, stat_x, stat_out, w64, (hipStream_t)stream);
      if (rc <= 0) return rc;
      if (stat_out) return RTP_ERR_UNSUPPORTED;   // (the generic kernel would write another number of partials)
    }
  }
  if (!transposed && g->stride == 2 && !stat_x) {
    const int rc = rtp_conv_s2_fwd_try(x, wf, w_per_sample, btab, res, y, g, relu, y_fp32, stat_out, acc32, acc_cs, (hipStream_t)stream, nullptr);
    if (rc <= 0) return rc;
  }
  if (acc32) return RTP_ERR_UNSUPPORTED;  // partial-sum input: only the LDS-tiled kernels (rtp_conv_tiled_ok)
  if (transposed && g->stride == 2 && !btab && !res && !relu && !y_fp32) {
    const int rc = rtp_dgrad_s2_try(x, wf, y, g, stat_x, stat_out, nullptr, (hipStream_t)stream);
    if (rc <= 0) return rc;
  }
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
  int nt, mt;
  igemm_shape(p.N, p.Vo, p.Co, &nt, &mt, &p.blocks_per_sample);
  const bool small = mt == 1;
  if (stat_out && (y_fp32 || (stat_x && stat_x->c < p.Co))) return RTP_ERR_UNSUPPORTED;
  p.stat_out = stat_out;
  p.stat_x = (stat_out && stat_x) ? (const bf16_t*)stat_x->ptr : nullptr;
  p.s_cs = stat_x ? stat_x->cs : 0; p.s_co = stat_x ? stat_x->co : 0;
  dim3 grid(p.N * p.blocks_per_sample, p.Co / (16 * nt));
  RtpProfScope prof(RTP_FAM_CONV, s);
  using Kern = void (*)(ConvParams);
  static const Kern table[2][2][2] = {
      {{conv_igemm_kernel<1, 1, 1>, conv_igemm_kernel<1, 1, 3>}, {conv_igemm_kernel<1, 4, 1>, conv_igemm_kernel<1, 4, 3>}},
      {{conv_igemm_kernel<2, 1, 1>, conv_igemm_kernel<2, 1, 3>}, {conv_igemm_kernel<2, 4, 1>, conv_igemm_kernel<2, 4, 3>}}};
  hipLaunchKernelGGL(table[nt - 1][small ? 0 : 1][p.ks == 3 ? 1 : 0], grid, dim3(256), 0, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
