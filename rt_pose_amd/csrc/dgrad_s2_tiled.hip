// LDS-tiled data gradient of the 3x3x3 STRIDE-2 convs that read a full-resolution 32-channel tensor (transition1 and the
// first conv of every fuse chain from branch 0: hr_util/hr3d.py:162-197, 297-305) -- i.e. the transposed conv that WRITES a
// full-resolution gradient from a half-resolution one.
//
//   dx[u][ci] = sum_{tap, co} gy[v][co] * W[co][ci][tap]   with  u = 2 v + tap - 1  (per axis)
//
// Parity decomposition: per axis an even output u = 2m is reached by tap 1 only (v = m), an odd one u = 2m + 1 by tap 0
// (v = m + 1) and tap 2 (v = m).  The 8 parity classes of an output brick are therefore 8 small stride-1 problems over the
// SAME half-resolution gy tile with 1, 2, 2, 2, 4, 4, 4, 8 taps (27 in all: 27/8 taps per output voxel instead of the 27 the
// gather formulation walks through with scalar pre-tests).  The kernel is bound by the full-resolution write (+ the read of x
// for GroupNorm backward), so the structure is simple: 4-wave workgroups, two per CU (70 KB of LDS each), persistent over
// contiguous brick runs; weights [27][32][32] resident in LDS; the haloed gy tile (2 x 3 x 17 voxels) double-buffered; a
// brick = 1 x 2 x 16 gy voxels = 2 x 4 x 32 output voxels; wave w owns parity classes w and 7 - w (1+8 / 2+4 taps).
// Epilogues = conv_tiled.hip's: either dxhat + the (P, Q) statistics, or the FINISHED gradient
// [x > 0] (A acc + B x + C + sum other consumers' terms) with the coefficients computed in the prologue (P from the
// boundary-class sums of gy, Q from the weight-gradient slabs' contraction).
#include <stdlib.h>
#include <string.h>

#include "rtp_common.h"
#include "rtp_prof.h"

#define GX 17
#define GY 3
#define GZ 2
#define GVOX (GZ * GY * GX)   // 102 haloed gy voxels per brick
#define GITEMS (GVOX * 4)     // 16-B items

struct S2Params {
  const bf16_t* gy; int g_cs, g_co;
  const bf16_t* w;                    // wd [27][32 ci][32 co]
  const bf16_t* x; int x_cs, x_co;    // the conv's input (full resolution): Q statistics / fused epilogue
  bf16_t* dx; int d_cs, d_co;
  float* stat_out;                    // [N][wgs][32][2] = (sum dx, sum dx * x) or null
  int N, Do, Ho, Wo;                  // gy dims; dx dims are twice these
  int tiles_z, tiles_y, tiles_x, tiles_per_sample, wgs_per_sample;
  // fused epilogue (see conv_tiled.hip, TiledParams)
  const float* coef[4];
  const bf16_t* ex[3]; int ex_cs[3], ex_co[3];
  int nextra, mask;
  float* tot_out;
  const float* qpart; int q_nsplit; const float* csum; const float* gn_mr; const float* gn_gamma; int gn_groups;
  float gn_m; float* coef_out;
};

__device__ __forceinline__ int swz2(int chunk, int xi) { return ((chunk + 2 * (xi >> 2)) & 3) << 3; }  // bf16 elements

// MODE 0: dxhat (+ statistics when stat_out); MODE 1 + NEX: fused epilogue with NEX extra terms
template <int MODE>
__global__ __launch_bounds__(256, 2) void dgrad_s2_kernel(S2Params p) {
  constexpr bool FUSE = MODE > 0;
  constexpr int NEX = FUSE ? MODE - 1 : 0;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  bf16_t* wL = lds;                                   // [27][32][32]
  bf16_t* gL = lds + 27 * 32 * 32;                    // [2][GVOX][32]
  float* bL = reinterpret_cast<float*>(gL + 2 * GVOX * 32);   // coefficient table / scratch, 864 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int v = lane & 15, q = lane >> 4;
  const int c0 = q * 8;
  const int bid = (gridDim.x % 8 == 0) ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int n = bid / p.wgs_per_sample, wg = bid - n * p.wgs_per_sample;
  const int D = 2 * p.Do, H = 2 * p.Ho, W = 2 * p.Wo;

  // ---- weights -> LDS, rows permuted as in conv_tiled (lane (voxel, q) ends up owning channels 8q..8q+7)
  for (int i = tid; i < 27 * 32 * 4; i += 256) {
    const int ck = i & 3, row = i >> 2, tap = row >> 5, ci = row & 31;
    const int arow = ((ci >> 2) & 1) * 16 + (ci >> 3) * 4 + (ci & 3);
    st_bf16x8(wL + (tap * 32 + arow) * 32 + swz2(ck, arow), ld_bf16x8(p.w + (long)row * 32 + ck * 8));
  }

  if constexpr (FUSE) {
    float gq = 0.f, gp_ = 0.f, gmu = 0.f, gr = 0.f, ggam = 0.f;
    if (p.qpart) {
      float* Cs = reinterpret_cast<float*>(gL);        // [64][32] boundary-class sums of gy; the gy buffers are still unused
      float* CSs = Cs + 64 * 32;                       // [27][32]   (2 x 102 x 64 B = 13 KB: 8 + 3.4 + 1 KB fit)
      float* Pp = CSs + 27 * 32;                       // [8][32]
      for (int i = tid; i < 64 * 32; i += 256) Cs[i] = p.csum[(long)n * 64 * 32 + i];
      __syncthreads();
      // sum of gy over the output voxels whose tap stays in bounds.  Stride 2, pad 1, even input size: tap 0 leaves the
      // volume at the FIRST output plane only (input 2v - 1 < 0); taps 1 and 2 never do
      for (int i = tid; i < 27 * 32; i += 256) {
        const int tap = i >> 5, co = i & 31;
        const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
        float vv = 0.f;
        for (int cls = 0; cls < 64; ++cls) {
          const bool out = (kz == 0 && (cls & 1)) || (ky == 0 && (cls & 4)) || (kx == 0 && (cls & 16));
          if (!out) vv += Cs[cls * 32 + co];
        }
        CSs[i] = vv;
      }
      __syncthreads();
      {
        const int c = tid & 31, k = tid >> 5;   // 8 tap groups
        float pa = 0.f;
        for (int tap = k; tap < 27; tap += 8) {
          const bf16_t* wr = p.w + ((long)tap * 32 + c) * 32;
#pragma unroll
          for (int co = 0; co < 32; co += 8) {
            const bf16x8 w8 = ld_bf16x8(wr + co);
#pragma unroll
            for (int j = 0; j < 8; ++j) pa += bf2f(w8[j]) * CSs[tap * 32 + co + j];
          }
        }
        Pp[k * 32 + c] = pa;
      }
      __syncthreads();
      if (tid < 32) {
#pragma unroll
        for (int k = 0; k < 8; ++k) gp_ += Pp[k * 32 + tid];
      }
      float* scr = bL + 448;
      {
        const int c = tid & 31, k = tid >> 5;
        float qq = 0.f;
        for (int s_ = k; s_ < p.q_nsplit; s_ += 8) qq += p.qpart[((long)n * p.q_nsplit + s_) * 32 + c];
        scr[tid] = qq;
      }
      __syncthreads();
      if (tid < 32) {
        const int cg = 32 / p.gn_groups, g = tid / cg;
#pragma unroll
        for (int k = 0; k < 8; ++k) gq += scr[k * 32 + tid];
        gmu = p.gn_mr[((long)n * p.gn_groups + g) * 2];
        gr = p.gn_mr[((long)n * p.gn_groups + g) * 2 + 1];
        ggam = p.gn_gamma[tid];
        scr[256 + tid] = ggam * gp_;
        scr[288 + tid] = ggam * gr * (gq - gmu * gp_);
      }
      __syncthreads();
    }
    if (tid < 32) {
      float a0 = 1.f, bt = 0.f, ct = 0.f;
      if (p.coef[0]) { const float* k = p.coef[0] + ((long)n * 32 + tid) * 3; a0 = k[0]; bt = k[1]; ct = k[2]; }
      if (p.qpart) {
        const float* scr = bL + 448;
        const int cg = 32 / p.gn_groups, g0 = (tid / cg) * cg;
        float s1 = 0.f, s2 = 0.f;
        for (int j = g0; j < g0 + cg; ++j) { s1 += scr[256 + j]; s2 += scr[288 + j]; }
        a0 = gr * ggam;
        bt = -gr * gr * s2 / p.gn_m;
        ct = -gr * s1 / p.gn_m + gr * gr * gmu * s2 / p.gn_m;
        if (p.coef_out && wg == 0) {
          float* o = p.coef_out + ((long)n * 32 + tid) * 3;
          o[0] = a0; o[1] = bt; o[2] = ct;
          float* pt = p.coef_out + (long)p.N * 32 * 3 + ((long)n * 32 + tid) * 2;
          pt[0] = gr * (gq - gmu * gp_);
          pt[1] = gp_;
        }
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        float ae = 1.f;
        if (e < NEX && p.coef[1 + e]) { const float* k = p.coef[1 + e] + ((long)n * 32 + tid) * 3; ae = k[0]; bt += k[1]; ct += k[2]; }
        bL[(3 + e) * 32 + tid] = ae;
      }
      bL[tid] = a0; bL[32 + tid] = bt; bL[64 + tid] = ct;
    }
    bL[192 + tid] = 0.f;   // per-wave running totals [4 waves][32] (+ slack)
    __syncthreads();
  }

  const long vox_g = (long)n * p.Do * p.Ho * p.Wo;
  const long vox_x = (long)n * D * H * W;
  const bf16_t* gn = p.gy + vox_g * p.g_cs + p.g_co;
  const int t_begin = (int)((long)wg * p.tiles_per_sample / p.wgs_per_sample);
  const int my_tiles = (int)((long)(wg + 1) * p.tiles_per_sample / p.wgs_per_sample) - t_begin;

  // staging descriptors: this thread's (up to) two 16-B items of the haloed gy tile
  int s_rel[2], s_dst[2], s_h[2];   // element offset relative to the tile origin, LDS element offset, packed (hz,hy,hx)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = tid + k * 256;
    const int ck = i & 3, hv = (i >> 2) < GVOX ? (i >> 2) : 0;
    const int hx = hv % GX, hy = (hv / GX) % GY, hz = hv / (GX * GY);
    s_rel[k] = ((hz * p.Ho + hy) * p.Wo + hx) * p.g_cs + ck * 8;
    s_dst[k] = hv * 32 + swz2(ck, hx);
    s_h[k] = hz | (hy << 4) | (hx << 8) | ((i < GITEMS) ? 0 : (1 << 16));
  }
  auto tile_org = [&](int tile, int& mz0, int& my0, int& mx0) {
    const int tz = tile % p.tiles_z, tx = (tile / p.tiles_z) % p.tiles_x, ty = tile / (p.tiles_z * p.tiles_x);  // z fastest
    mz0 = tz; my0 = ty * 2; mx0 = tx * 16;
  };
  auto load_items = [&](int tile, bf16x8 (&it)[2]) {
    int mz0, my0, mx0;
    tile_org(tile, mz0, my0, mx0);
    const long org = (((long)mz0 * p.Ho + my0) * p.Wo + mx0) * p.g_cs;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int hz = s_h[k] & 15, hy = (s_h[k] >> 4) & 15, hx = (s_h[k] >> 8) & 255;
      const bool ok = !(s_h[k] >> 16) && (mz0 + hz < p.Do) && (my0 + hy < p.Ho) && (mx0 + hx < p.Wo);
      it[k] = ok ? ld_bf16x8(gn + org + s_rel[k]) : zero_bf16x8();
    }
  };
  auto store_items = [&](int buf, const bf16x8 (&it)[2]) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (!(s_h[k] >> 16)) st_bf16x8(gL + buf * GVOX * 32 + s_dst[k], it[k]);
  };

  bf16x8 items[2];
  if (my_tiles > 0) {
    load_items(t_begin, items);
    store_items(0, items);
  }
  __syncthreads();

  float st_p[8], st_q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) st_p[j] = st_q[j] = 0.f;
  typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) bf16_t*)lds;
  const unsigned a_base = lds0 + 2u * (v * 32 + swz2(q, v));

  for (int k = 0; k < my_tiles; ++k) {
    if (k + 1 < my_tiles) load_items(t_begin + k + 1, items);
    int mz0, my0, mx0;
    tile_org(t_begin + k, mz0, my0, mx0);
    const unsigned g_base = lds0 + 2u * (unsigned)(27 * 32 * 32 + (k & 1) * GVOX * 32);
    // 4 groups per wave: (class, y row); the x / extra operands of all four are requested up front
    bf16x8 xr[4], exr[NEX > 0 ? NEX : 1][4];
    long vo[4];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const int cls = (gi < 2) ? wave : 7 - wave, r = gi & 1;
      const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
      vo[gi] = vox_x + ((long)(2 * mz0 + pz) * H + (2 * (my0 + r) + py)) * W + 2 * (mx0 + v) + px;
      xr[gi] = zero_bf16x8();
      if (p.x) xr[gi] = ld_bf16x8(p.x + vo[gi] * p.x_cs + p.x_co + c0);
#pragma unroll
      for (int e = 0; e < NEX; ++e) exr[e][gi] = ld_bf16x8(p.ex[e] + vo[gi] * p.ex_cs[e] + p.ex_co[e] + c0);
    }
    float ka[8], kb[8], kc[8], ke[NEX > 0 ? NEX : 1][8];
    if constexpr (FUSE) {
#pragma unroll
      for (int kk = 0; kk < 8; kk += 4) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(bL + c0 + kk), b4 = *reinterpret_cast<const f32x4*>(bL + 32 + c0 + kk),
                    c4 = *reinterpret_cast<const f32x4*>(bL + 64 + c0 + kk);
#pragma unroll
        for (int j = 0; j < 4; ++j) { ka[kk + j] = a4[j]; kb[kk + j] = b4[j]; kc[kk + j] = c4[j]; }
#pragma unroll
        for (int e = 0; e < NEX; ++e) {
          const f32x4 e4 = *reinterpret_cast<const f32x4*>(bL + (3 + e) * 32 + c0 + kk);
#pragma unroll
          for (int j = 0; j < 4; ++j) ke[e][kk + j] = e4[j];
        }
      }
    }
    float tsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const int cls = (gi < 2) ? wave : 7 - wave, r = gi & 1;
      const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      for (int iz = 0; iz <= pz; ++iz) {
        const int kz = pz ? (iz ? 2 : 0) : 1, hz = (pz && !iz) ? 1 : 0;
        for (int iy = 0; iy <= py; ++iy) {
          const int ky = py ? (iy ? 2 : 0) : 1, hy = r + ((py && !iy) ? 1 : 0);
          for (int ix = 0; ix <= px; ++ix) {
            const int kx = px ? (ix ? 2 : 0) : 1, hx = v + ((px && !ix) ? 1 : 0);
            const int tap = (kz * 3 + ky) * 3 + kx;
            const bf16x8 a0 = *(lds_frag)(a_base + 2u * ((tap * 32) * 32));
            const bf16x8 a1 = *(lds_frag)(a_base + 2u * ((tap * 32 + 16) * 32));
            const bf16x8 b = *(lds_frag)(g_base + 2u * (unsigned)(((hz * GY + hy) * GX + hx) * 32 + swz2(q, hx)));
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc1, 0, 0, 0);
          }
        }
      }
      float ev[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { ev[j] = acc0[j]; ev[4 + j] = acc1[j]; }
      float aux[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) aux[j] = bf2f(xr[gi][j]);
      if constexpr (FUSE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ev[j] = ev[j] * ka[j] + (kb[j] * aux[j] + kc[j]);
#pragma unroll
        for (int e = 0; e < NEX; ++e)
#pragma unroll
          for (int j = 0; j < 8; ++j) ev[j] += ke[e][j] * bf2f(exr[e][gi][j]);
        if (p.mask) {
#pragma unroll
          for (int j = 0; j < 8; ++j) ev[j] = aux[j] > 0.f ? ev[j] : 0.f;
        }
      }
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(ev[j]);
      st_bf16x8(p.dx + vo[gi] * p.d_cs + p.d_co + c0, o);
      if constexpr (FUSE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) tsum[j] += bf2f(o[j]);
      } else {
        if (p.stat_out) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float rr = bf2f(o[j]);
            st_p[j] += rr;
            st_q[j] += rr * aux[j];
          }
        }
      }
    }
    if constexpr (FUSE) {
      if (p.tot_out) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          tsum[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tsum[j]), 0x111, 0xf, 0xf, true));
          tsum[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tsum[j]), 0x112, 0xf, 0xf, true));
          tsum[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tsum[j]), 0x114, 0xf, 0xf, true));
          tsum[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tsum[j]), 0x118, 0xf, 0xf, true));
        }
        if (v == 15) {
          float* tp = bL + 192 + wave * 32 + c0;
#pragma unroll
          for (int kk = 0; kk < 8; kk += 4) {
            f32x4 a4 = *reinterpret_cast<const f32x4*>(tp + kk);
#pragma unroll
            for (int j = 0; j < 4; ++j) a4[j] += tsum[kk + j];
            *reinterpret_cast<f32x4*>(tp + kk) = a4;
          }
        }
      }
    }
    if (k + 1 < my_tiles) store_items((k + 1) & 1, items);
    __syncthreads();
  }
  if constexpr (FUSE) {
    if (p.tot_out && tid < 32)
      p.tot_out[(long)bid * 32 + tid] = (bL[192 + tid] + bL[224 + tid]) + (bL[256 + tid] + bL[288 + tid]);
  } else {
    if (p.stat_out) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          st_p[j] += __shfl_xor(st_p[j], o, 64);
          st_q[j] += __shfl_xor(st_q[j], o, 64);
        }
      float* red = bL;   // [4 waves][32][2]
      if (v == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          red[(wave * 32 + c0 + j) * 2] = st_p[j];
          red[(wave * 32 + c0 + j) * 2 + 1] = st_q[j];
        }
      }
      __syncthreads();
      if (tid < 64) {
        float a = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) a += red[w4 * 64 + tid];
        p.stat_out[(long)bid * 64 + tid] = a;
      }
    }
  }
}

static bool s2_geometry_ok(const RtpAct* gy, const RtpConvGeom* g) {
  static const bool disabled = getenv("RTP_DISABLE_TILED") != nullptr;
  if (disabled) return false;
  if (g->ks != 3 || g->stride != 2 || g->pad != 1 || g->ci != 32 || (g->co + 31) / 32 * 32 != 32) return false;
  if (g->di != 2 * g->dov || g->hi != 2 * g->ho || g->wi != 2 * g->wo) return false;
  if (g->ho % 2 || g->wo % 16) return false;
  if (gy->cs % 8 || gy->co % 8) return false;
  return true;
}

static int s2_wgs_per_sample(const RtpConvGeom* g) {
  const int tiles = g->dov * (g->ho / 2) * (g->wo / 16);
  static const int total_wgs = 512;
  int wgs = total_wgs / g->n;
  if (wgs < 1) wgs = 1;
  if (wgs > tiles) wgs = tiles;
  return wgs;
}

// statistics / totals partials per sample of this kernel for the geometry (0: not this kernel's)
int rtp_dgrad_s2_stat_slots(const RtpAct* gy, const RtpConvGeom* g) { return s2_geometry_ok(gy, g) ? s2_wgs_per_sample(g) : 0; }

struct TiledFuse;   // conv_tiled.hip
struct S2Fuse { const float* coef[4]; const RtpAct* ex[3]; int nextra, mask; float* tot_out; const RtpGnBwd* gn; };

// +1: not this kernel's geometry; RTP_OK / negative otherwise.  stat_x: the conv's input when statistics or the fused epilogue
// are requested.
int rtp_dgrad_s2_try(const RtpAct* gy, const void* wd, const RtpAct* dx, const RtpConvGeom* g, const RtpAct* stat_x, float* stat_out,
                     const S2Fuse* fuse, hipStream_t s) {
  if (!s2_geometry_ok(gy, g)) return 1;
  if ((dx->cs % 8) || (dx->co % 8) || dx->c < 32 || gy->c < 32) return RTP_ERR_ALIGN;
  if (stat_out && !stat_x) return RTP_ERR_SHAPE;
  if (stat_x && ((stat_x->cs % 8) || (stat_x->co % 8) || stat_x->c < 32)) return RTP_ERR_ALIGN;
  S2Params p;
  memset(&p, 0, sizeof(p));
  p.gy = (const bf16_t*)gy->ptr; p.g_cs = gy->cs; p.g_co = gy->co;
  p.w = (const bf16_t*)wd;
  p.x = stat_x ? (const bf16_t*)stat_x->ptr : nullptr; p.x_cs = stat_x ? stat_x->cs : 0; p.x_co = stat_x ? stat_x->co : 0;
  p.dx = (bf16_t*)dx->ptr; p.d_cs = dx->cs; p.d_co = dx->co;
  p.stat_out = stat_out;
  p.N = g->n; p.Do = g->dov; p.Ho = g->ho; p.Wo = g->wo;
  p.tiles_z = p.Do; p.tiles_y = p.Ho / 2; p.tiles_x = p.Wo / 16;
  p.tiles_per_sample = p.tiles_z * p.tiles_y * p.tiles_x;
  p.wgs_per_sample = s2_wgs_per_sample(g);
  p.gn_groups = 1; p.gn_m = 1.f;
  int mode = 0;
  if (fuse) {
    if (!stat_x || stat_out || fuse->nextra < 0 || fuse->nextra > 3) return RTP_ERR_SHAPE;
    mode = 1 + fuse->nextra;
    p.nextra = fuse->nextra; p.mask = fuse->mask; p.tot_out = fuse->tot_out;
    p.coef[0] = fuse->coef[0];
    for (int e = 0; e < fuse->nextra; ++e) {
      if (!fuse->ex[e] || fuse->ex[e]->c < 32 || (fuse->ex[e]->cs % 8) || (fuse->ex[e]->co % 8)) return RTP_ERR_ALIGN;
      p.ex[e] = (const bf16_t*)fuse->ex[e]->ptr; p.ex_cs[e] = fuse->ex[e]->cs; p.ex_co[e] = fuse->ex[e]->co;
      p.coef[1 + e] = fuse->coef[1 + e];
    }
    if (fuse->gn) {
      const RtpGnBwd* q = fuse->gn;
      if (!q->qpart || !q->csum || !q->mr || !q->gamma || q->q_nsplit < 1 || q->groups < 1 || 32 % q->groups) return RTP_ERR_SHAPE;
      p.qpart = q->qpart; p.q_nsplit = q->q_nsplit; p.csum = q->csum; p.gn_mr = q->mr; p.gn_gamma = q->gamma;
      p.gn_groups = q->groups; p.gn_m = (float)(32 / q->groups) * (float)((long)g->di * g->hi * g->wi);
      p.coef_out = q->coeff_out;
    }
  }
  const size_t shm = sizeof(bf16_t) * (27 * 32 * 32 + 2 * GVOX * 32) + 864 * sizeof(float);
  using Kern = void (*)(S2Params);
  static const Kern tab[5] = {dgrad_s2_kernel<0>, dgrad_s2_kernel<1>, dgrad_s2_kernel<2>, dgrad_s2_kernel<3>, dgrad_s2_kernel<4>};
  static bool attr[RTP_MAX_DEVICES] = {};
  if (rtp_once_per_device(attr)) {
    for (int i = 0; i < 5; ++i) (void)hipFuncSetAttribute((const void*)tab[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  }
  RtpProfScope prof(RTP_FAM_CONV_TILED, s);
  hipLaunchKernelGGL(tab[mode], dim3(p.N * p.wgs_per_sample), dim3(256), shm, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
