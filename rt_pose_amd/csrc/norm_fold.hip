// GroupNorm statistics, GroupNorm->weight folding, and the small "algebra" kernels of the backward pass.
//
// GroupNorm(8,C) in front of every backbone conv (hr_util/common.py:57, hr3d.py:83,147,168,297,323) is never
// materialised: x_hat[c] = x[c]*scale[n,c] + shift[n,c] is folded into per-sample bf16 weights
// (W*scale) and a per-boundary-class bias (sum over in-bounds taps of W.shift), because zero padding is
// applied AFTER the norm.  The backward pass undoes the fold with per-(n,c) sums (P,Q) and per-class sums.
#include <stdlib.h>
#include <string.h>

#include "rtp_common.h"
#include "rtp_prof.h"

// ------------------------------------------------------------------------------------------------
// rtp_chan_stats
// ------------------------------------------------------------------------------------------------
template <bool HAS_B>
__global__ __launch_bounds__(256) void chan_stats_kernel(const bf16_t* a, int a_cs, int a_co, const bf16_t* b,
                                                         int b_cs, int b_co, int c, long vox, int nsplit,
                                                         float* out) {
  __shared__ float red[256 * 17];
  const int n = blockIdx.y, s = blockIdx.x;
  const int cpv = c >> 3;  // 8-channel chunks per voxel
  const int tid = threadIdx.x;
  const int chunk = tid % cpv, vsub = tid / cpv, vper = 256 / cpv;
  const long vps = (vox + nsplit - 1) / nsplit;
  const long v0 = s * vps, v1 = (v0 + vps < vox) ? v0 + vps : vox;
  float s0[8], s1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.f;
  if (vsub < vper) {
    for (long v = v0 + vsub; v < v1; v += vper) {
      const long vv = (long)n * vox + v;
      bf16x8 x = ld_bf16x8(a + vv * a_cs + a_co + chunk * 8);
      if (HAS_B) {
        bf16x8 y = ld_bf16x8(b + vv * b_cs + b_co + chunk * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float xf = bf2f(x[j]);
          s0[j] += xf;
          s1[j] += xf * bf2f(y[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float xf = bf2f(x[j]);
          s0[j] += xf;
          s1[j] += xf * xf;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[tid * 17 + j] = s0[j];
    red[tid * 17 + 8 + j] = s1[j];
  }
  __syncthreads();
  // thread t < 2*c : channel ch = t>>1, which = t&1 ; sum over vsub in fixed order (deterministic)
  for (int t = tid; t < 2 * c; t += 256) {
    const int ch = t >> 1, which = t & 1;
    const int ck = ch >> 3, j = ch & 7;
    float acc = 0.f;
    for (int vs = 0; vs < vper; ++vs) acc += red[(vs * cpv + ck) * 17 + which * 8 + j];
    out[(((long)n * nsplit + s) * c + ch) * 2 + which] = acc;
  }
}

extern "C" int rtp_chan_stats(const RtpAct* a, const RtpAct* b, int n, long vox, int nsplit, float* out,
                              void* stream) {
  if (!a || !out || nsplit < 1) return RTP_ERR_SHAPE;
  const int c = a->c;
  if (c % 8 || c > 256) return RTP_ERR_UNSUPPORTED;   // (256 % (c / 8) != 0, e.g. 192 channels: the last threads of a block idle)
  if ((a->cs % 8) || (a->co % 8) || (b && ((b->cs % 8) || (b->co % 8) || b->c != c))) return RTP_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  dim3 grid(nsplit, n);
  if (b)
    hipLaunchKernelGGL(chan_stats_kernel<true>, grid, dim3(256), 0, s, (const bf16_t*)a->ptr, a->cs, a->co,
                       (const bf16_t*)b->ptr, b->cs, b->co, c, vox, nsplit, out);
  else
    hipLaunchKernelGGL(chan_stats_kernel<false>, grid, dim3(256), 0, s, (const bf16_t*)a->ptr, a->cs, a->co,
                       (const bf16_t*)nullptr, 0, 0, c, vox, nsplit, out);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_fold_fwd
// ------------------------------------------------------------------------------------------------
struct FoldParams {
  const float* w; const float* bias; const float* gamma; const float* beta; const float* stats;
  int nsplit, groups; float eps;
  int ci_real, co_real, ci_pad, co_pad, ntap, ks, stride, pad, ci_total, ci_off;
  int di, hi, wi, dov, ho, wo;
  bf16_t* wf; float* btab; float* mr;
  bf16_t* wd; int cok;  // optional data-gradient packing [tap][ci_pad][cok] (written by the n == 0 blocks)
  int cos, nw;          // output channels per block; weight sets (n with GroupNorm, 1 without)
  int dbg;              // timing experiments only (RTP_FOLD_DBG): bit0 skip statistics, bit1 skip wf, bit2 skip bias table, bit3 skip weight staging
};

__device__ __forceinline__ bool tap_inb_class(int tap, int cls, const FoldParams& p) {
  const int ks = p.ks;
  const int kz = tap / (ks * ks), ky = (tap / ks) % ks, kx = tap % ks;
  return tap_inb_1d(kz, cls & 1, (cls >> 1) & 1, p.dov, p.di, p.stride, p.pad) &&
         tap_inb_1d(ky, (cls >> 2) & 1, (cls >> 3) & 1, p.ho, p.hi, p.stride, p.pad) &&
         tap_inb_1d(kx, (cls >> 4) & 1, (cls >> 5) & 1, p.wo, p.wi, p.stride, p.pad);
}

// Output channels per block: every block owns all taps of its couts, so the bias table needs no cross-block sum.  Chosen
// so the block's fp32 weight rows [cos][ci_real*ntap] fit 32 KB of LDS.
static int fold_cos(int ci_real, int ntap) {
  int cos = 4;
  while (cos > 1 && (size_t)cos * ci_real * ntap * sizeof(float) > 32 * 1024) cos >>= 1;
  return cos;
}
static size_t fold_shm(const FoldParams& p) {
  return sizeof(float) * (2 * (size_t)p.ci_pad + (size_t)((p.cos * p.ntap + 3) & ~3) + (size_t)p.cos * p.ci_real * p.ntap);
}

// The block first pulls its weight rows into LDS with coalesced loads (in the reference layout [co][ci][tap] consecutive
// input channels are 108 B apart: read in place, every lane of a wave hit its own cache line, 14 dependent rounds per
// block), and only then waits for the statistics; everything after that is LDS -> registers -> 16-B stores.
__device__ __forceinline__ void fold_fwd_body(const FoldParams& p, int n, int by, float* smem) {
  float* scale = smem;                 // [ci_pad]
  float* shift = scale + p.ci_pad;     // [ci_pad]
  float* T = shift + p.ci_pad;         // [cos][ntap]
  float* wS = T + ((p.cos * p.ntap + 3) & ~3);  // [cos][ci_real][ntap]
  const int tid = threadIdx.x, cos = p.cos, ntap = p.ntap;
  const int co0 = by * cos;
  const bool norm = p.stats != nullptr;
  const int L = p.ci_real * ntap;
  // Statistics partials [n][nsplit][C][2] are requested FIRST so that their latency overlaps the weight staging below:
  // thread t owns channel t % C and every (256/C)-th partial (C a power of two: shifts, no division; the loads of one
  // wave are consecutive float2s), keeping an fp32 running pair per thread.
  const bool do_norm = norm && p.wf && !(p.dbg & 1);
  const bool pow2 = (p.ci_real & (p.ci_real - 1)) == 0 && p.ci_real <= 256;
  float ps0 = 0.f, ps1 = 0.f;
  float2 pre4[4] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
  float gam_r = 0.f, bet_r = 0.f;   // this thread's channel's affine parameters, also requested up front
  if (do_norm && pow2 && tid < p.ci_real) { gam_r = p.gamma[tid]; bet_r = p.beta[tid]; }
  if (do_norm && pow2) {   // the first four partials of this thread: loads only, consumed after the weight staging
    const int C = p.ci_real, c = tid & (C - 1), sh = __ffs(C) - 1, s_first = tid >> sh, s_step = 256 >> sh;
    const float2* q = reinterpret_cast<const float2*>(p.stats) + ((long)n * p.nsplit) * C + c;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (s_first + k * s_step < p.nsplit) pre4[k] = q[(long)(s_first + k * s_step) * C];
  }
  for (int col = 0; col < cos && !(p.dbg & 8); ++col) {
    const int co = co0 + col;
    const float* row = p.w + ((long)co * p.ci_total + p.ci_off) * ntap;
    float* dst = wS + col * L;
    if (co < p.co_real) {
      int i = tid;
      for (; i + 3 * 256 < L; i += 4 * 256) {
        const float v0 = row[i], v1 = row[i + 256], v2 = row[i + 512], v3 = row[i + 768];
        dst[i] = v0; dst[i + 256] = v1; dst[i + 512] = v2; dst[i + 768] = v3;
      }
      for (; i < L; i += 256) dst[i] = row[i];
    } else {
      for (int i = tid; i < L; i += 256) dst[i] = 0.f;
    }
  }
  for (int c = tid; c < p.ci_pad; c += 256) { scale[c] = (c < p.ci_real && !norm) ? 1.f : 0.f; shift[c] = 0.f; }
  if (do_norm && pow2) {
    {
      const int C = p.ci_real, c = tid & (C - 1), sh = __ffs(C) - 1, s_first = tid >> sh, s_step = 256 >> sh;
      const float2* q = reinterpret_cast<const float2*>(p.stats) + ((long)n * p.nsplit) * C + c;
      ps0 = (pre4[0].x + pre4[1].x) + (pre4[2].x + pre4[3].x);
      ps1 = (pre4[0].y + pre4[1].y) + (pre4[2].y + pre4[3].y);
      for (int s_ = s_first + 4 * s_step; s_ < p.nsplit; s_ += s_step) { const float2 a0 = q[(long)s_ * C]; ps0 += a0.x; ps1 += a0.y; }
    }
    // fold: the 256/C threads of a channel and the cg channels of a group, in fixed order, in double
    __shared__ float2 part[256];
    part[tid] = make_float2(ps0, ps1);
    __syncthreads();
    const int C = p.ci_real, cg = C / p.groups, np = 256 / C;
    __shared__ float mr_l[64][2];
    if (tid < p.groups) {
      double s0 = 0.0, s1 = 0.0;
      for (int k = 0; k < np; ++k)
        for (int j = 0; j < cg; ++j) { const float2 v = part[k * C + tid * cg + j]; s0 += v.x; s1 += v.y; }
      // sums and the E[x^2] - E[x]^2 cancellation in double; the reciprocal square root in fp32 (as ATen's group_norm):
      // a double divide + sqrt on this one thread was ~1 us of every fold launch
      const double inv = 1.0 / ((double)cg * p.di * p.hi * p.wi);
      const double mean = s0 * inv;
      double var = s1 * inv - mean * mean;
      if (var < 0.0) var = 0.0;
      const float rstd = 1.0f / sqrtf((float)var + p.eps);
      mr_l[tid][0] = (float)mean;
      mr_l[tid][1] = rstd;
      if (p.mr && by == 0) {
        p.mr[((long)n * p.groups + tid) * 2] = (float)mean;
        p.mr[((long)n * p.groups + tid) * 2 + 1] = rstd;
      }
    }
    __syncthreads();
    if (tid < C) {
      const float mean = mr_l[tid / cg][0], sc = mr_l[tid / cg][1] * gam_r;
      scale[tid] = sc;
      shift[tid] = bet_r - mean * sc;
    }
    __syncthreads();
  } else {
  __syncthreads();
  if (do_norm) {
    const int cg = p.ci_real / p.groups;
    const double cnt = (double)cg * p.di * p.hi * p.wi;
    // one wave-sized team per group: lanes split the (channel, split) partials, then a shuffle reduction
    const int lane = tid & 63, team = tid >> 6;
    for (int g = team; g < p.groups; g += 4) {
      double s0 = 0.0, s1 = 0.0;
      const int items = cg * p.nsplit;
      for (int i = lane; i < items; i += 64) {
        const int c = g * cg + i / p.nsplit, s = i % p.nsplit;
        const float* q = p.stats + (((long)n * p.nsplit + s) * p.ci_real + c) * 2;
        s0 += q[0];
        s1 += q[1];
      }
      for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
      const double mean = s0 / cnt;
      double var = s1 / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
      if (p.mr && by == 0 && lane == 0) {
        p.mr[((long)n * p.groups + g) * 2] = (float)mean;
        p.mr[((long)n * p.groups + g) * 2 + 1] = rstd;
      }
      for (int c = g * cg + lane; c < (g + 1) * cg; c += 64) {
        const float sc = rstd * p.gamma[c];
        scale[c] = sc;
        shift[c] = p.beta[c] - (float)mean * sc;
      }
    }
    __syncthreads();
  }
  }
  if (p.wd && n == 0) {  // the un-folded weights, transposed for the data-gradient conv: wd[tap][ci_pad][cok]
    for (int i = tid; i < ntap * p.ci_pad; i += 256) {
      const int ci = i % p.ci_pad, tap = i / p.ci_pad;
      bf16_t* o = p.wd + ((long)tap * p.ci_pad + ci) * p.cok + co0;
      for (int col = 0; col < cos; ++col)
        if (co0 + col < p.cok) o[col] = f2bf(ci < p.ci_real ? wS[col * L + ci * ntap + tap] : 0.f);
    }
  }
  if (!p.wf) return;
  if (!(p.dbg & 2)) {
  // folded weights  wf[n][tap][co][ci] for co in [co0, co0+cos): one 16-B store per 8 input channels
  bf16_t* wf = p.wf + (long)n * ntap * p.co_pad * p.ci_pad;
  const int c8 = p.ci_pad >> 3;
  for (int i = tid; i < ntap * cos * c8; i += 256) {
    const int k = i % c8, col = (i / c8) % cos, tap = i / (c8 * cos);
    const int co = co0 + col;
    if (co >= p.co_pad) continue;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = k * 8 + j;
      o[j] = f2bf(ci < p.ci_real ? wS[col * L + ci * ntap + tap] * scale[ci] : 0.f);
    }
    st_bf16x8(wf + ((long)tap * p.co_pad + co) * p.ci_pad + k * 8, o);
  }
  }
  if (!p.btab || (p.dbg & 4)) return;
  // T[col][tap] = sum_ci w*shift : two lanes per dot product (halves of the input channels), four independent partial
  // sums per lane (the LDS reads of a serial chain each waited their full latency), joined by one shuffle
  {
    const int i = tid >> 1, half = tid & 1;
    for (int i0 = 0; i0 < cos * ntap; i0 += 128) {
      const int ii = i0 + i;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      if (norm && ii < cos * ntap) {
        const int tap = ii % ntap, col = ii / ntap;
        const int h0 = half * (p.ci_real >> 1), h1 = half ? p.ci_real : (p.ci_real >> 1);
        const float* wr = wS + col * L + tap;
        int ci = h0;
        for (; ci + 3 < h1; ci += 4) {
          a0 += wr[ci * ntap] * shift[ci];
          a1 += wr[(ci + 1) * ntap] * shift[ci + 1];
          a2 += wr[(ci + 2) * ntap] * shift[ci + 2];
          a3 += wr[(ci + 3) * ntap] * shift[ci + 3];
        }
        for (; ci < h1; ++ci) a0 += wr[ci * ntap] * shift[ci];
      }
      float acc = (a0 + a1) + (a2 + a3);
      acc += __shfl_xor(acc, 1, 64);
      if (ii < cos * ntap && half == 0) T[ii] = acc;
    }
  }
  __syncthreads();
  // bias of a boundary class = bias + sum of T over the taps that stay in bounds for that class.  The in-bounds test
  // factorises per axis: three 3-bit masks per class instead of 27 full tests.
  float* bt = p.btab + (long)n * 64 * p.co_pad;
  for (int i = tid; i < 64 * cos; i += 256) {
    const int col = i % cos, cls = i / cos, co = co0 + col;
    float acc = (p.bias && co < p.co_real) ? p.bias[co] : 0.f;
    if (norm) {
      const int ks = p.ks;
      int mz = 0, my = 0, mx = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k >= ks) break;
        mz |= (int)tap_inb_1d(k, cls & 1, (cls >> 1) & 1, p.dov, p.di, p.stride, p.pad) << k;
        my |= (int)tap_inb_1d(k, (cls >> 2) & 1, (cls >> 3) & 1, p.ho, p.hi, p.stride, p.pad) << k;
        mx |= (int)tap_inb_1d(k, (cls >> 4) & 1, (cls >> 5) & 1, p.wo, p.wi, p.stride, p.pad) << k;
      }
      const float* Tc = T + col * ntap;
      if (ks == 3) {   // straight-line: 27 predicated adds
#pragma unroll
        for (int t = 0; t < 27; ++t)
          acc += (((mz >> (t / 9)) & (my >> ((t / 3) % 3)) & (mx >> (t % 3))) & 1) ? Tc[t] : 0.f;
      } else {
        acc += ((mz & my & mx) & 1) ? Tc[0] : 0.f;
      }
    }
    if (co < p.co_pad) bt[cls * p.co_pad + co] = acc;
  }
}

__global__ __launch_bounds__(256) void fold_fwd_kernel(FoldParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  fold_fwd_body(p, blockIdx.x, blockIdx.y, smem);
}

static int fill_fold(FoldParams& p, const RtpConvGeom* g, int ci_real, int co_real) {
  if (!g || (g->ks != 1 && g->ks != 3)) return RTP_ERR_UNSUPPORTED;
  p.ci_real = ci_real; p.co_real = co_real; p.ci_pad = g->ci; p.co_pad = g->co;
  p.ks = g->ks; p.ntap = g->ks * g->ks * g->ks; p.stride = g->stride; p.pad = g->pad;
  p.ci_total = g->w_ci_total ? g->w_ci_total : ci_real; p.ci_off = g->w_ci_off;
  p.di = g->di; p.hi = g->hi; p.wi = g->wi; p.dov = g->dov; p.ho = g->ho; p.wo = g->wo;
  if (ci_real > p.ci_pad || co_real > p.co_pad) return RTP_ERR_SHAPE;
  return RTP_OK;
}

// wf == NULL: only the data-gradient packing wd (which does not depend on the statistics)
static int fill_fold_fwd(FoldParams& p, const float* w, const float* bias, const float* gamma, const float* beta,
                         const float* stats, int nsplit, int groups, float eps, const RtpConvGeom* g, int ci_real,
                         int co_real, void* wf, float* btab, float* mr, void* wd, int* nw, int* ny) {
  int rc = fill_fold(p, g, ci_real, co_real);
  if (rc) return rc;
  if (!w || (!wf && !wd)) return RTP_ERR_SHAPE;
  if (stats && (!gamma || !beta || groups < 1 || ci_real % groups)) return RTP_ERR_SHAPE;
  p.w = w; p.bias = bias; p.gamma = gamma; p.beta = beta; p.stats = stats;
  p.nsplit = nsplit; p.groups = groups; p.eps = eps;
  p.wf = (bf16_t*)wf; p.btab = wf ? btab : nullptr; p.mr = wf ? mr : nullptr;
  p.wd = (bf16_t*)wd; p.cok = (g->co + 31) / 32 * 32;
  p.cos = fold_cos(ci_real, p.ntap);
  static const int dbg = 0;
  p.dbg = dbg;
  if (p.ci_pad % 8) return RTP_ERR_UNSUPPORTED;
  if (fold_shm(p) > 60 * 1024) return RTP_ERR_UNSUPPORTED;
  *nw = (wf && stats) ? g->n : 1;
  *ny = (wd ? p.cok : p.co_pad) / p.cos;
  return RTP_OK;
}

extern "C" int rtp_fold_fwd(const float* w, const float* bias, const float* gamma, const float* beta,
                            const float* stats, int nsplit, int groups, float eps, const RtpConvGeom* g, int ci_real,
                            int co_real, void* wf, float* btab, float* mr, void* wd, void* stream) {
  FoldParams p;
  int nw, ny;
  int rc = fill_fold_fwd(p, w, bias, gamma, beta, stats, nsplit, groups, eps, g, ci_real, co_real, wf, btab, mr, wd, &nw, &ny);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  hipLaunchKernelGGL(fold_fwd_kernel, dim3(nw, ny), dim3(256), fold_shm(p), s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_pack_dgrad_w : wd[tap][ci_pad][cok] = w[co][ci][tap]
// ------------------------------------------------------------------------------------------------
__global__ void pack_dgrad_kernel(const float* w, int ci_real, int co_real, int ci_pad, int cok, int ntap, int ci_total,
                                  int ci_off, bf16_t* wd) {
  const long total = (long)ntap * ci_pad * cok;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = i % cok, ci = (i / cok) % ci_pad, tap = i / ((long)cok * ci_pad);
    float v = 0.f;
    if (co < co_real && ci < ci_real) v = w[((long)co * ci_total + ci_off + ci) * ntap + tap];
    wd[i] = f2bf(v);
  }
}

extern "C" int rtp_pack_dgrad_w(const float* w, const RtpConvGeom* g, int ci_real, int co_real, void* wd,
                                void* stream) {
  if (!g || !w || !wd) return RTP_ERR_SHAPE;
  const int ntap = g->ks * g->ks * g->ks;
  const int cok = (g->co + 31) / 32 * 32;
  const long total = (long)ntap * g->ci * cok;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  int blocks = rtp_div_up(total, 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3(blocks), dim3(256), 0, s, w, ci_real, co_real, g->ci, cok, ntap,
                     g->w_ci_total ? g->w_ci_total : ci_real, g->w_ci_off, (bf16_t*)wd);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_gn_bwd_coeffs   (single block; n is small)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_bwd_coeffs_kernel(const float* pq, int nsplit, const float* mr,
                                                            const float* gamma, int c, int groups, long vox,
                                                            float* coeff, float* part /*[n][c][2]*/) {
  __shared__ float P[256], Q[256], S1[64], S2[64];
  __shared__ float2 part_pq[256];
  const int tid = threadIdx.x, i = blockIdx.x;
  const int cg = c / groups;
  const float m = (float)cg * (float)vox;
  float pc = 0.f, qc = 0.f, mu = 0.f, r = 0.f, gam = 0.f;
  {
    // the partials of a channel are split over 256/c threads (a conv epilogue may leave hundreds per sample), then
    // folded in fixed order
    const int np = 256 / c, ch = tid % c, pt = tid / c;
    float2 a = make_float2(0.f, 0.f);
    if (pt < np)
      for (int s = pt; s < nsplit; s += np) {
        const float2 q = *reinterpret_cast<const float2*>(pq + (((long)i * nsplit + s) * c + ch) * 2);
        a.x += q.x;
        a.y += q.y;
      }
    part_pq[tid] = a;
    __syncthreads();
    if (tid < c)
      for (int k = 0; k < np; ++k) { pc += part_pq[k * c + tid].x; qc += part_pq[k * c + tid].y; }
  }
  if (tid < c) {
    const int g = tid / cg;
    mu = mr[((long)i * groups + g) * 2];
    r = mr[((long)i * groups + g) * 2 + 1];
    gam = gamma[tid];
    P[tid] = gam * pc;                  // gamma * sum dxhat
    Q[tid] = gam * r * (qc - mu * pc);  // gamma * sum dxhat * xnorm
    part[((long)i * c + tid) * 2] = r * (qc - mu * pc);
    part[((long)i * c + tid) * 2 + 1] = pc;
  }
  __syncthreads();
  if (tid < groups) {
    float s1 = 0.f, s2 = 0.f;
    for (int k = tid * cg; k < (tid + 1) * cg; ++k) { s1 += P[k]; s2 += Q[k]; }
    S1[tid] = s1;
    S2[tid] = s2;
  }
  __syncthreads();
  if (tid < c) {
    const int g = tid / cg;
    float* o = coeff + ((long)i * c + tid) * 3;
    o[0] = r * gam;
    o[1] = -r * r * S2[g] / m;
    o[2] = -r * S1[g] / m + r * r * mu * S2[g] / m;
  }
}

__global__ void gn_bwd_param_kernel(const float* part, int n, int c, float* dgamma, float* dbeta, int accumulate) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * c) return;
  const int ch = t >> 1, which = t & 1;
  float acc = 0.f;
  for (int i = 0; i < n; ++i) acc += part[((long)i * c + ch) * 2 + which];
  float* o = which ? dbeta + ch : dgamma + ch;
  if (accumulate) *o += acc; else *o = acc;
}

extern "C" int rtp_gn_bwd_coeffs(const float* pq, int nsplit, const float* mr, const float* gamma, int n, int c,
                                 int groups, long vox, float* coeff, float* dgamma, float* dbeta, int accumulate,
                                 void* stream) {
  if (c > 256 || groups > 64 || c % groups) return RTP_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  float* part = coeff + (long)n * c * 3;  // the coeff buffer carries n*c*2 floats of scratch behind the coefficients
  hipLaunchKernelGGL(gn_bwd_coeffs_kernel, dim3(n), dim3(256), 0, s, pq, nsplit, mr, gamma, c, groups, vox, coeff, part);
  if (dgamma && dbeta)  // NULL: the caller sums `part` later (rtp_tail_desc_gn_param)
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((2 * c + 255) / 256), dim3(256), 0, s, part, n, c, dgamma, dbeta, accumulate);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_gn_bwd_coeffs_cls : the same coefficients WITHOUT a pass over dxhat (one block per sample).
//   Q[ci] = sum_v dxhat[v][ci] x[v][ci] = sum_{tap,co} W[co][ci][tap] G[tap][co][ci]   (G = this sample's weight-gradient
//           correlation; the tiled weight-gradient kernel contracts each slab with the weights: qpart[n][slab][ci])
//   P[ci] = sum_v dxhat[v][ci]          = sum_{tap,co} W[co][ci][tap] * (sum of gy[.][co] over the voxels whose `tap` is in
//           bounds) -- the per-boundary-class sums of gy the bias / un-fold already needs.
// W is the bf16 data-gradient packing wd[tap][ci_pad][cok] -- the very values the data-gradient conv multiplies with.
// With P and Q known BEFORE the data gradient runs, its epilogue writes the finished dx = A*dxhat + B*x + C.
// ------------------------------------------------------------------------------------------------
struct GnClsParams {
  const float* qpart; int q_nsplit; const float* cls_part; int cls_nsplit; float* csum_out;
  const bf16_t* wd; const float* mr; const float* gamma; FoldParams f; int groups, co32; long vox; float* coeff; float* part;
  float* p_out;   // non-null: write P [n][ci] only (no Q, no coefficients)
};

// Shared by the coefficient kernel and by the last block of the class-sum scans: reduce class-sum partials of sample n into
// LDS (csum [64][co32]; `tot` != null: the interior class is the per-channel total minus the boundary classes), optionally
// store them, then CS[tap][co] = sum over the classes in which `tap` is in bounds, and P[c] = sum_{tap,co} wd * CS
// (partials left in red[k*C + c], k < 256/C).  256 threads; sh as laid out below.
__device__ __forceinline__ void reduce_csum_and_p(const float* cls_part, int cls_nsplit, const float* tot, int tot_nsplit,
                                                  float* csum_out, const bf16_t* wd, const FoldParams& f, int co32, int n,
                                                  float* sh) {
  const int tid = threadIdx.x;
  const int ntap = f.ntap, C = f.ci_real, cip = f.ci_pad;
  float* csum = sh;                       // [64][co32]
  float* CS = csum + 64 * co32;           // [ntap][co32]
  float* red = CS + ntap * co32;          // [256]
  unsigned* vm = reinterpret_cast<unsigned*>(red + 256 + 512 + 128);   // [ntap][2]: bit cls set = tap in bounds for that class
  for (int i = tid; i < ntap * 2; i += 256) vm[i] = 0u;
  for (int i = tid; i < 64 * co32; i += 256) {
    const float* src = cls_part + (long)n * cls_nsplit * 64 * co32 + i;
    float a = 0.f;
    int s_ = 0;
    for (; s_ + 8 <= cls_nsplit; s_ += 8) {   // eight loads in flight
      float g8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) g8[k] = src[(long)(s_ + k) * 64 * co32];
#pragma unroll
      for (int k = 0; k < 8; ++k) a += g8[k];
    }
    for (; s_ < cls_nsplit; ++s_) a += src[(long)s_ * 64 * co32];
    csum[i] = a;
  }
  __syncthreads();
  if (tot) {
    for (int ch = tid; ch < co32; ch += 256) {
      float t = 0.f;
      for (int s_ = 0; s_ < tot_nsplit; ++s_) t += tot[((long)n * tot_nsplit + s_) * co32 + ch];
      float bsum = 0.f;
      for (int cls = 1; cls < 64; ++cls) bsum += csum[cls * co32 + ch];
      csum[ch] = t - bsum;
    }
    __syncthreads();
  }
  if (csum_out)
    for (int i = tid; i < 64 * co32; i += 256) csum_out[(long)n * 64 * co32 + i] = csum[i];
  if (!wd) return;
  for (int i = tid; i < ntap * 64; i += 256) {
    const int tap = i >> 6, cls = i & 63;
    if (tap_inb_class(tap, cls, f)) atomicOr(&vm[tap * 2 + (cls >> 5)], 1u << (cls & 31));
  }
  __syncthreads();
  for (int i = tid; i < ntap * co32; i += 256) {
    const int tap = i / co32, co = i - tap * co32;
    float a = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      unsigned m = vm[tap * 2 + h];
      while (m) {
        const int b = __ffs(m) - 1;
        m &= m - 1;
        a += csum[(h * 32 + b) * co32 + co];
      }
    }
    CS[i] = a;
  }
  __syncthreads();
  // P: thread (c, k) takes taps k, k + np, ...; a tap's weight row wd[tap][c][0..co32) is contiguous
  const int np = 256 / C, c = tid % C, k = tid / C;
  float pacc = 0.f;
  if (k < np)
    for (int tap = k; tap < ntap; tap += np) {
      const bf16_t* wr = wd + ((long)tap * cip + c) * co32;
      const float* cr = CS + tap * co32;
      for (int co = 0; co < co32; co += 8) {
        const bf16x8 w8 = ld_bf16x8(wr + co);
#pragma unroll
        for (int j = 0; j < 8; ++j) pacc += bf2f(w8[j]) * cr[co + j];
      }
    }
  red[tid] = pacc;
  __syncthreads();
}
static size_t reduce_csum_shm(int co32, int ntap) { return sizeof(float) * ((size_t)64 * co32 + (size_t)ntap * co32 + 256 + 512 + 128 + 64); }

__global__ __launch_bounds__(256) void gn_bwd_coeffs_cls_kernel(GnClsParams p) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int tid = threadIdx.x, n = blockIdx.x;
  const int co32 = p.co32, ntap = p.f.ntap, C = p.f.ci_real, cip = p.f.ci_pad;
  float* red = sh + 64 * co32 + ntap * co32;   // [256]
  float* Pq = red + 256;                  // [2][256]: gamma*P, gamma*r*(Q - mu P) per channel
  float* S12 = Pq + 512;                  // [2][64]
  const int np = 256 / C;
  reduce_csum_and_p(p.cls_part, p.cls_nsplit, nullptr, 0, p.csum_out, p.wd, p.f, co32, n, sh);
  float pc = 0.f, qc = 0.f, mu = 0.f, r = 0.f, gam = 0.f;
  if (p.p_out) {
    if (tid < C) {
      for (int kk = 0; kk < np; ++kk) pc += red[kk * C + tid];
      p.p_out[(long)n * C + tid] = pc;
    }
    return;
  }
  const int cg = C / p.groups;
  const float m = (float)cg * (float)p.vox;
  if (tid < C) {
    for (int kk = 0; kk < np; ++kk) pc += red[kk * C + tid];
    const float* q = p.qpart + (long)n * p.q_nsplit * cip + tid;
    for (int s = 0; s < p.q_nsplit; ++s) qc += q[(long)s * cip];
    const int g = tid / cg;
    mu = p.mr[((long)n * p.groups + g) * 2];
    r = p.mr[((long)n * p.groups + g) * 2 + 1];
    gam = p.gamma[tid];
    Pq[tid] = gam * pc;
    Pq[256 + tid] = gam * r * (qc - mu * pc);
    p.part[((long)n * C + tid) * 2] = r * (qc - mu * pc);
    p.part[((long)n * C + tid) * 2 + 1] = pc;
  }
  __syncthreads();
  if (tid < p.groups) {
    float s1 = 0.f, s2 = 0.f;
    for (int j = tid * cg; j < (tid + 1) * cg; ++j) { s1 += Pq[j]; s2 += Pq[256 + j]; }
    S12[tid] = s1;
    S12[64 + tid] = s2;
  }
  __syncthreads();
  if (tid < C) {
    const int g = tid / cg;
    float* o = p.coeff + ((long)n * C + tid) * 3;
    o[0] = r * gam;
    o[1] = -r * r * S12[64 + g] / m;
    o[2] = -r * S12[g] / m + r * r * mu * S12[64 + g] / m;
  }
}


extern "C" int rtp_gn_bwd_coeffs_cls(const float* qpart, int q_nsplit, const float* cls_part, int cls_nsplit, float* csum_out,
                                     const void* wd, const float* mr, const float* gamma, const RtpConvGeom* g, int ci_real,
                                     int co_real, int groups, float* coeff, void* stream) {
  if (!qpart || !cls_part || !wd || !mr || !gamma || !g || !coeff || q_nsplit < 1 || cls_nsplit < 1) return RTP_ERR_SHAPE;
  GnClsParams p;
  int rc = fill_fold(p.f, g, ci_real, co_real);
  if (rc) return rc;
  if (ci_real > 256 || 256 % ci_real || groups > 64 || ci_real % groups || ci_real != p.f.ci_pad) return RTP_ERR_UNSUPPORTED;
  p.co32 = (g->co + 31) / 32 * 32;
  const size_t shm = reduce_csum_shm(p.co32, p.f.ntap);
  if (shm > 60 * 1024) return RTP_ERR_UNSUPPORTED;
  p.qpart = qpart; p.q_nsplit = q_nsplit; p.cls_part = cls_part; p.cls_nsplit = cls_nsplit; p.csum_out = csum_out;
  p.wd = (const bf16_t*)wd; p.mr = mr; p.gamma = gamma; p.groups = groups; p.vox = (long)g->di * g->hi * g->wi;
  p.coeff = coeff; p.part = coeff + (long)g->n * ci_real * 3; p.p_out = nullptr;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  hipLaunchKernelGGL(gn_bwd_coeffs_cls_kernel, dim3(g->n), dim3(256), shm, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// qpart[n][s][ci] = sum_{tap,co} wd[tap][ci][co] * gp[n][s][tap][co][ci]: one block per slab; thread (ci, k) walks the
// (tap, co) pairs k, k + 256/ci, ... -- slab reads are contiguous along ci, the weights come from L2.
__global__ __launch_bounds__(256) void qpart_from_slabs_kernel(const float* gp, int ntap, int co32, int ci, const bf16_t* wd,
                                                              float* qpart) {
  __shared__ float red[256];
  const int tid = threadIdx.x, c = tid % ci, k = tid / ci, np = 256 / ci;
  const float* slab = gp + (long)blockIdx.x * ntap * co32 * ci;
  float a = 0.f;
  for (int r = k; r < ntap * co32; r += np) {
    const int tap = r / co32, co = r - tap * co32;
    a += slab[(long)r * ci + c] * bf2f(wd[((long)tap * ci + c) * co32 + co]);
  }
  red[tid] = a;
  __syncthreads();
  if (tid < ci) {
    float t = 0.f;
    for (int kk = 0; kk < np; ++kk) t += red[kk * ci + tid];
    qpart[(long)blockIdx.x * ci + tid] = t;
  }
}

extern "C" int rtp_qpart_from_slabs(const float* gp, int n, int nsplit, int ntap, int co32, int ci, const void* wd, float* qpart,
                                    void* stream) {
  if (!gp || !wd || !qpart || n < 1 || nsplit < 1 || ntap < 1) return RTP_ERR_SHAPE;
  if (ci > 256 || 256 % ci || co32 % 32) return RTP_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  hipLaunchKernelGGL(qpart_from_slabs_kernel, dim3(n * nsplit), dim3(256), 0, s, gp, ntap, co32, ci, (const bf16_t*)wd, qpart);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

extern "C" int rtp_gn_bwd_p(const float* cls_part, int cls_nsplit, float* csum_out, const void* wd, const RtpConvGeom* g,
                            int ci_real, int co_real, float* p_out, void* stream) {
  if (!cls_part || !wd || !g || !p_out || cls_nsplit < 1) return RTP_ERR_SHAPE;
  GnClsParams p;
  int rc = fill_fold(p.f, g, ci_real, co_real);
  if (rc) return rc;
  if (ci_real > 256 || 256 % ci_real || ci_real != p.f.ci_pad) return RTP_ERR_UNSUPPORTED;
  p.co32 = (g->co + 31) / 32 * 32;
  const size_t shm = reduce_csum_shm(p.co32, p.f.ntap);
  if (shm > 60 * 1024) return RTP_ERR_UNSUPPORTED;
  p.qpart = nullptr; p.q_nsplit = 0; p.cls_part = cls_part; p.cls_nsplit = cls_nsplit; p.csum_out = csum_out;
  p.wd = (const bf16_t*)wd; p.mr = nullptr; p.gamma = nullptr; p.groups = 1; p.vox = 1; p.coeff = nullptr; p.part = nullptr;
  p.p_out = p_out;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  hipLaunchKernelGGL(gn_bwd_coeffs_cls_kernel, dim3(g->n), dim3(256), shm, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_class_sums : out[n][64][c]  (partials [n][nsplit][64][c] in `scratch`, then a fixed-order reduction)
// One wave per x-row: the (z,y) flags are row-uniform, only x==0 / x==W-1 differ, so lanes accumulate three
// register sets (interior / first / last) and touch the LDS buckets once per row.
// ------------------------------------------------------------------------------------------------
// boundary_only: voxels of the interior class (no face touched: the bulk of the tensor) are skipped -- their class sum is
// derived from the per-channel TOTAL the producing kernel emitted (class_sums_final_tot_kernel), so only ~1/6 of a
// 16 x 64 x 160 tensor is read.
__global__ __launch_bounds__(256) void class_sums_kernel(const bf16_t* g, int cs, int co, int c, int D, int H, int W,
                                                         int nsplit, float* part, int boundary_only) {
  extern __shared__ __attribute__((aligned(16))) float cls_sum[];  // [64][c]
  const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long vox = (long)D * H * W;
  for (int i = tid; i < 64 * c; i += 256) cls_sum[i] = 0.f;
  __syncthreads();
  const int cpv = c >> 3;           // 4, 8, 16 or 32 divide 64 (shuffle reduction); other widths (96 channels: 12) leave the
  const bool pow2 = (64 % cpv) == 0;  // last 64 % cpv lanes idle and every lane adds its own sums to the LDS table
  const int act = pow2 ? 64 : (64 / cpv) * cpv;
  const int chunk = lane % cpv;
  const int rows = D * H;
  const int rps = (rows + nsplit - 1) / nsplit;
  const int r0 = s * rps, r1 = (r0 + rps < rows) ? r0 + rps : rows;
  for (int r = r0 + wave; r < r1; r += 4) {
    const int z = r / H, y = r - z * H;
    const int czy = (z == 0) | ((z == D - 1) << 1) | ((y == 0) << 2) | ((y == H - 1) << 3);
    float a_in[8], a_f[8], a_l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a_in[j] = a_f[j] = a_l[j] = 0.f;
    const bf16_t* row = g + ((long)n * vox + (long)r * W) * cs + co;
    const bool edge_only = boundary_only && czy == 0 && W > 2;   // interior row: only its first and last voxel
    for (int i = lane; lane < act && i < (edge_only ? 2 : W) * cpv; i += act) {
      const int x = edge_only ? ((i / cpv) ? W - 1 : 0) : i / cpv;
      bf16x8 t = ld_bf16x8(row + (long)x * cs + chunk * 8);
      const bool first = (x == 0), last = (x == W - 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = bf2f(t[j]);
        if (first && last) a_f[j] += v;          // W == 1: bucket "first|last" (kept in a_f, flagged below)
        else if (first) a_f[j] += v;
        else if (last) a_l[j] += v;
        else a_in[j] += v;
      }
    }
    // reduce over lanes that share a chunk (lane, lane+cpv, ...)
    if (pow2) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        for (int o = 32; o >= cpv; o >>= 1) {
          a_in[j] += __shfl_xor(a_in[j], o, 64);
          a_f[j] += __shfl_xor(a_f[j], o, 64);
          a_l[j] += __shfl_xor(a_l[j], o, 64);
        }
    }
    if (pow2 ? lane < cpv : lane < act) {
      const int cf = czy | (1 << 4) | ((W == 1) << 5), cl = czy | (1 << 5);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&cls_sum[czy * c + chunk * 8 + j], a_in[j]);
        atomicAdd(&cls_sum[cf * c + chunk * 8 + j], a_f[j]);
        if (W > 1) atomicAdd(&cls_sum[cl * c + chunk * 8 + j], a_l[j]);
      }
    }
  }
  __syncthreads();
  float* o = part + ((long)n * nsplit + s) * 64 * c;
  for (int i = tid; i < 64 * c; i += 256) o[i] = cls_sum[i];
}

struct ClsRedParams { const float* part; float* out; int nsplit, per_n, n; };

__device__ __forceinline__ void class_reduce_body(const ClsRedParams& p, int bid) {
  const long i4 = (long)bid * 256 + threadIdx.x;       // one float4 of out[n][64*c] per thread
  const long total4 = (long)p.n * p.per_n / 4;
  if (i4 >= total4) return;
  const long n = (i4 * 4) / p.per_n, r = i4 * 4 - n * p.per_n;
  const float* src = p.part + n * p.nsplit * p.per_n + r;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= p.nsplit; s += 8) {
    f32x4 g[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = *reinterpret_cast<const f32x4*>(src + (long)(s + k) * p.per_n);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += g[k];
  }
  for (; s < p.nsplit; ++s) acc += *reinterpret_cast<const f32x4*>(src + (long)s * p.per_n);
  *reinterpret_cast<f32x4*>(p.out + i4 * 4) = acc;
}

__global__ __launch_bounds__(256) void class_sums_final(ClsRedParams p) { class_reduce_body(p, blockIdx.x); }

static void launch_class_final(const float* part, int nsplit, int n, int c, float* out, hipStream_t s) {
  ClsRedParams p;
  p.part = part; p.out = out; p.nsplit = nsplit; p.per_n = 64 * c; p.n = n;
  hipLaunchKernelGGL(class_sums_final, dim3(rtp_div_up((long)n * 64 * c / 4, 256)), dim3(256), 0, s, p);
}

extern "C" int rtp_class_sums(const RtpAct* gy, int n, int d, int h, int w, int nsplit, float* scratch, float* out,
                              void* stream) {
  if (!gy || !scratch) return RTP_ERR_SHAPE;
  const int c = gy->c;
  if (c % 8 || c > 256) return RTP_ERR_UNSUPPORTED;
  if ((gy->cs % 8) || (gy->co % 8)) return RTP_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  hipLaunchKernelGGL(class_sums_kernel, dim3(nsplit, n), dim3(256), sizeof(float) * 64 * c, s, (const bf16_t*)gy->ptr,
                     gy->cs, gy->co, c, d, h, w, nsplit, scratch, 0);
  if (out) launch_class_final(scratch, nsplit, n, c, out, s);  // out == NULL: partials only (reduced later, e.g. by the tail)
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// Boundary-only scan with the work dealt evenly (the faces are a small, very uneven part of the row space): every wave of
// the grid takes full rows of the z / y faces round-robin (all loads of a row in flight at once), then one lane per
// (interior row, x side, 16-B chunk) takes the x-face voxels.  D, H, W > 2.
__global__ __launch_bounds__(256) void class_sums_boundary_kernel(const bf16_t* g, int cs, int co, int c, int D, int H, int W,
                                                                  int nsplit, float* part) {
  extern __shared__ __attribute__((aligned(16))) float cls_sum[];  // [64][c]
  const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long vox = (long)D * H * W;
  for (int i = tid; i < 64 * c; i += 256) cls_sum[i] = 0.f;
  __syncthreads();
  const int cpv = c >> 3, chunk = lane % cpv;
  const int gw = s * 4 + wave, NW = nsplit * 4;
  const bf16_t* gn = g + (long)n * vox * cs + co;
  // ---- full rows: 2*H rows of the z faces, then 2*(D-2) rows of the y faces
  const int nfull = 2 * H + 2 * (D - 2);
  for (int fr = gw; fr < nfull; fr += NW) {
    int z, y;
    if (fr < 2 * H) { z = (fr < H) ? 0 : D - 1; y = fr % H; }
    else { const int k = fr - 2 * H; z = 1 + (k >> 1); y = (k & 1) ? H - 1 : 0; }
    const int czy = (z == 0) | ((z == D - 1) << 1) | ((y == 0) << 2) | ((y == H - 1) << 3);
    float a_in[8], a_f[8], a_l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a_in[j] = a_f[j] = a_l[j] = 0.f;
    const bf16_t* row = gn + ((long)z * H + y) * W * cs;
    const int items = W * cpv;
    for (int i0 = 0; i0 < items; i0 += 64 * 4) {
      bf16x8 t4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 64 + lane;
        t4[k] = (i < items) ? ld_bf16x8(row + (long)(i / cpv) * cs + chunk * 8) : zero_bf16x8();
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 64 + lane, x = i / cpv;
        const bool first = (x == 0), last = (x == W - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = bf2f(t4[k][j]);
          if (first) a_f[j] += v; else if (last) a_l[j] += v; else a_in[j] += v;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      for (int o = 32; o >= cpv; o >>= 1) {
        a_in[j] += __shfl_xor(a_in[j], o, 64);
        a_f[j] += __shfl_xor(a_f[j], o, 64);
        a_l[j] += __shfl_xor(a_l[j], o, 64);
      }
    if (lane < cpv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&cls_sum[czy * c + chunk * 8 + j], a_in[j]);
        atomicAdd(&cls_sum[(czy | 16) * c + chunk * 8 + j], a_f[j]);
        atomicAdd(&cls_sum[(czy | 32) * c + chunk * 8 + j], a_l[j]);
      }
    }
  }
  // ---- x faces of the interior rows: item = (row, side, chunk); lanes sharing (side, chunk) sit 2*cpv apart
  const int per_row = 2 * cpv, nint = (D - 2) * (H - 2);
  const int rows_per_wave = 64 / per_row;
  for (int r0 = gw * rows_per_wave; r0 < nint; r0 += NW * rows_per_wave) {
    const int r = r0 + lane / per_row, side = (lane / cpv) & 1;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    if (r < nint) {
      const int z = 1 + r / (H - 2), y = 1 + r % (H - 2);
      const bf16x8 t = ld_bf16x8(gn + (((long)z * H + y) * W + (side ? W - 1 : 0)) * cs + chunk * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = bf2f(t[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      for (int o = 32; o >= per_row; o >>= 1) a[j] += __shfl_xor(a[j], o, 64);
    if (lane < per_row) {
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(&cls_sum[(side ? 32 : 16) * c + chunk * 8 + j], a[j]);
    }
  }
  __syncthreads();
  float* o = part + ((long)n * nsplit + s) * 64 * c;
  for (int i = tid; i < 64 * c; i += 256) o[i] = cls_sum[i];
}

// Final reduction for the boundary-only scan: classes 1..63 from the partials, class 0 (interior) = total - their sum.
__global__ __launch_bounds__(512) void class_sums_final_tot_kernel(const float* part, int nsplit, const float* tot, int tot_nsplit,
                                                                   float* out, int c) {
  extern __shared__ __attribute__((aligned(16))) float shc[];  // [64][c]
  const int n = blockIdx.x, tid = threadIdx.x;
  const int per_n = 64 * c;
  for (int i4 = tid; i4 < per_n / 4; i4 += 512) {
    const float* src = part + (long)n * nsplit * per_n + i4 * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int s_ = 0;
    for (; s_ + 8 <= nsplit; s_ += 8) {
      f32x4 g8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) g8[k] = *reinterpret_cast<const f32x4*>(src + (long)(s_ + k) * per_n);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += g8[k];
    }
    for (; s_ < nsplit; ++s_) acc += *reinterpret_cast<const f32x4*>(src + (long)s_ * per_n);
    *reinterpret_cast<f32x4*>(shc + i4 * 4) = acc;
  }
  __syncthreads();
  for (int ch = tid; ch < c; ch += 512) {
    float t = 0.f;
    for (int s_ = 0; s_ < tot_nsplit; ++s_) t += tot[((long)n * tot_nsplit + s_) * c + ch];
    float b = 0.f;
    for (int cls = 1; cls < 64; ++cls) b += shc[cls * c + ch];
    shc[ch] = t - b;
  }
  __syncthreads();
  for (int i4 = tid; i4 < per_n / 4; i4 += 512)
    *reinterpret_cast<f32x4*>(out + (long)n * per_n + i4 * 4) = *reinterpret_cast<const f32x4*>(shc + i4 * 4);
}

// One launch for "class sums of gy, and P of the GroupNorm backward from them": the scan blocks write their partials, and
// the LAST block of each sample to arrive (agent-scope counter) runs the reduction + P -- instead of three dependent launches
// (scan, reduction, P) between the data gradient that produced gy and the one that needs P.
struct ClsScanParams {
  const bf16_t* g; int cs, co, c, D, H, W, nsplit; float* part;
  int boundary;                     // 1: only boundary voxels are read; the interior class comes from the totals
  const float* tot; int tot_nsplit;
  float* csum_out; const bf16_t* wd; FoldParams f; float* p_out;
  int* counters;                    // [n], zero before the first launch; the last block resets its sample's counter
};

__global__ __launch_bounds__(256) void class_scan_p_kernel(ClsScanParams p) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  __shared__ int is_last;
  const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = p.c, D = p.D, H = p.H, W = p.W;
  float* cls_sum = sh;
  const long vox = (long)D * H * W;
  for (int i = tid; i < 64 * c; i += 256) cls_sum[i] = 0.f;
  __syncthreads();
  const int cpv = c >> 3, chunk = lane % cpv;
  const bf16_t* gn = p.g + (long)n * vox * p.cs + p.co;
  const int gw = s * 4 + wave, NW = p.nsplit * 4;
  const bool thin = !(D > 2 && H > 2 && W > 2);
  // rows scanned in full: every row (full scan / thin volume), or the rows of the z and y faces
  const int nfull = (!p.boundary || thin) ? D * H : 2 * H + 2 * (D - 2);
  for (int fr = gw; fr < nfull; fr += NW) {
    int z, y;
    if (!p.boundary || thin) { z = fr / H; y = fr - z * H; }
    else if (fr < 2 * H) { z = (fr < H) ? 0 : D - 1; y = fr % H; }
    else { const int k = fr - 2 * H; z = 1 + (k >> 1); y = (k & 1) ? H - 1 : 0; }
    const int czy = (z == 0) | ((z == D - 1) << 1) | ((y == 0) << 2) | ((y == H - 1) << 3);
    float a_in[8], a_f[8], a_l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a_in[j] = a_f[j] = a_l[j] = 0.f;
    const bf16_t* row = gn + ((long)z * H + y) * W * p.cs;
    const int items = W * cpv;
    for (int i0 = 0; i0 < items; i0 += 64 * 4) {
      bf16x8 t4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 64 + lane;
        t4[k] = (i < items) ? ld_bf16x8(row + (long)(i / cpv) * p.cs + chunk * 8) : zero_bf16x8();
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 64 + lane, x = i / cpv;
        const bool first = (x == 0), last = (x == W - 1) && W > 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = bf2f(t4[k][j]);
          if (first) a_f[j] += v; else if (last) a_l[j] += v; else a_in[j] += v;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      for (int o = 32; o >= cpv; o >>= 1) {
        a_in[j] += __shfl_xor(a_in[j], o, 64);
        a_f[j] += __shfl_xor(a_f[j], o, 64);
        a_l[j] += __shfl_xor(a_l[j], o, 64);
      }
    if (lane < cpv) {
      const int cf = czy | 16 | ((W == 1) << 5);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&cls_sum[czy * c + chunk * 8 + j], a_in[j]);
        atomicAdd(&cls_sum[cf * c + chunk * 8 + j], a_f[j]);
        if (W > 1) atomicAdd(&cls_sum[(czy | 32) * c + chunk * 8 + j], a_l[j]);
      }
    }
  }
  if (p.boundary && !thin) {   // x faces of the interior rows: item = (row, side, chunk)
    const int per_row = 2 * cpv, nint = (D - 2) * (H - 2);
    const int rows_per_wave = 64 / per_row;
    for (int r0 = gw * rows_per_wave; r0 < nint; r0 += NW * rows_per_wave) {
      const int r = r0 + lane / per_row, side = (lane / cpv) & 1;
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = 0.f;
      if (r < nint) {
        const int z = 1 + r / (H - 2), y = 1 + r % (H - 2);
        const bf16x8 t = ld_bf16x8(gn + (((long)z * H + y) * W + (side ? W - 1 : 0)) * p.cs + chunk * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = bf2f(t[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        for (int o = 32; o >= per_row; o >>= 1) a[j] += __shfl_xor(a[j], o, 64);
      if (lane < per_row) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&cls_sum[(side ? 32 : 16) * c + chunk * 8 + j], a[j]);
      }
    }
  }
  __syncthreads();
  float* o = p.part + ((long)n * p.nsplit + s) * 64 * c;
  for (int i = tid; i < 64 * c; i += 256) o[i] = cls_sum[i];
  // ---- the last block of this sample finishes the job (release: every wave's stores are drained by the barrier, then one
  // lane's agent-scope read-modify-write; acquire: the same lane, then the barrier, then plain loads)
  __syncthreads();
  if (tid == 0) {
    const int prev = __hip_atomic_fetch_add(p.counters + n, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (prev == p.nsplit - 1);
    if (is_last) __hip_atomic_store(p.counters + n, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!is_last) return;
  const bool totals = p.boundary && !thin;
  reduce_csum_and_p(p.part, p.nsplit, totals ? p.tot : nullptr, p.tot_nsplit, p.csum_out, p.wd, p.f, c, n, sh);
  if (p.wd && p.p_out) {
    const int C = p.f.ci_real, np = 256 / C;
    const float* red = sh + 64 * c + p.f.ntap * c;
    if (tid < C) {
      float pc = 0.f;
      for (int kk = 0; kk < np; ++kk) pc += red[kk * C + tid];
      p.p_out[(long)n * C + tid] = pc;
    }
  }
}

// Class sums of gy [n][64][c] -> csum_out, and (wd != NULL) P [n][ci] of the GroupNorm backward of the conv whose output
// gradient gy is, in ONE launch.  tot_part != NULL: gy's per-channel totals are known (rtp_conv_dgrad_fused), only the
// boundary voxels are scanned.  counters: device int [n], zeroed once by the caller (self-resetting).
extern "C" int rtp_class_sums_p(const RtpAct* gy, int n, int d, int h, int w, int nsplit, float* scratch, const float* tot_part,
                                int tot_nsplit, float* csum_out, const void* wd, const RtpConvGeom* g, int ci_real, int co_real,
                                float* p_out, int* counters, void* stream) {
  if (!gy || !scratch || !csum_out || !counters || nsplit < 1 || (tot_part && tot_nsplit < 1)) return RTP_ERR_SHAPE;
  if (wd && (!g || !p_out)) return RTP_ERR_SHAPE;
  const int c = gy->c;
  if (c % 8 || c > 128 || (64 % (2 * (c / 8)))) return RTP_ERR_UNSUPPORTED;
  if ((gy->cs % 8) || (gy->co % 8)) return RTP_ERR_ALIGN;
  ClsScanParams p;
  memset(&p, 0, sizeof(p));
  p.f.ntap = 27;
  if (wd) {
    int rc = fill_fold(p.f, g, ci_real, co_real);
    if (rc) return rc;
    if (ci_real > 256 || 256 % ci_real || ci_real != p.f.ci_pad || (g->co + 31) / 32 * 32 != c) return RTP_ERR_UNSUPPORTED;
  }
  p.g = (const bf16_t*)gy->ptr; p.cs = gy->cs; p.co = gy->co; p.c = c; p.D = d; p.H = h; p.W = w; p.nsplit = nsplit;
  p.part = scratch; p.boundary = tot_part ? 1 : 0; p.tot = tot_part; p.tot_nsplit = tot_nsplit;
  p.csum_out = csum_out; p.wd = (const bf16_t*)wd; p.p_out = p_out; p.counters = counters;
  const size_t shm = reduce_csum_shm(c, p.f.ntap);
  if (shm > 60 * 1024) return RTP_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  hipLaunchKernelGGL(class_scan_p_kernel, dim3(nsplit, n), dim3(256), shm, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// rtp_class_sums for a tensor whose per-channel totals are already known (tot_part fp32 [n][tot_nsplit][c], e.g. from
// rtp_conv_dgrad_fused): scans only the boundary voxels; out fp32 [n][64][c] as rtp_class_sums.
extern "C" int rtp_class_sums_boundary(const RtpAct* gy, int n, int d, int h, int w, int nsplit, float* scratch,
                                       const float* tot_part, int tot_nsplit, float* out, void* stream) {
  if (!gy || !scratch || !tot_part || !out || nsplit < 1 || tot_nsplit < 1) return RTP_ERR_SHAPE;
  const int c = gy->c;
  if (c % 8 || c > 128 || (64 % (c / 8))) return RTP_ERR_UNSUPPORTED;
  if ((gy->cs % 8) || (gy->co % 8)) return RTP_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  if (d > 2 && h > 2 && w > 2 && 64 % (2 * (c / 8)) == 0)
    hipLaunchKernelGGL(class_sums_boundary_kernel, dim3(nsplit, n), dim3(256), sizeof(float) * 64 * c, s, (const bf16_t*)gy->ptr,
                       gy->cs, gy->co, c, d, h, w, nsplit, scratch);
  else   // thin volumes: (nearly) everything is boundary
    hipLaunchKernelGGL(class_sums_kernel, dim3(nsplit, n), dim3(256), sizeof(float) * 64 * c, s, (const bf16_t*)gy->ptr,
                       gy->cs, gy->co, c, d, h, w, nsplit, scratch, 1);
  hipLaunchKernelGGL(class_sums_final_tot_kernel, dim3(n), dim3(512), sizeof(float) * 64 * c, s, scratch, nsplit, tot_part,
                     tot_nsplit, out, c);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

extern "C" int rtp_class_sums_reduce(const float* scratch, int nsplit, int n, int c, float* out, void* stream) {
  if (!scratch || !out || nsplit < 1) return RTP_ERR_SHAPE;
  if (c % 8) return RTP_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  launch_class_final(scratch, nsplit, n, c, out, s);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_wgrad_fold : one block per (tap, group of R = 256/ci_pad output-channel rows)
// ------------------------------------------------------------------------------------------------
struct WFoldParams {
  const float* gp; int nsplit; const float* csum; const float* mr; const float* gamma; const float* beta;
  int groups, n; FoldParams f; int co32, csum_c; float* dw; float* dbias; int accumulate;
  const float* tg;   // optional (bias-only convs): subset-sum partials [n][nsplit][27][32] of gy from rtp_wgrad_tg; dbias = sum of slot 0
};

__host__ __device__ __forceinline__ int wfold_rows(int ci_pad) { return 256 / ci_pad; }  // rows a wave covers with one 1-KB read
static int wfold_blocks(const WFoldParams& p) {
  const int R = wfold_rows(p.f.ci_pad);
  return p.f.ntap * ((p.f.co_real + R - 1) / R);
}
static size_t wfold_shm(const WFoldParams& p) {
  const int R = wfold_rows(p.f.ci_pad);
  return sizeof(float) * (2 * (size_t)p.n * p.f.ci_pad + (size_t)((p.n * R + R + 3) & ~3) + 4 * 64 * 4);
}

// The slabs [n][split][tap][co32][ci_pad] are the bulk of the bytes (one slab per workgroup of the tiled weight-gradient
// kernel: 28 MB per full-resolution layer), so the mapping is built around them: a wave reads 1 KB of CONTIGUOUS slab
// (R rows) per load, the block's 4 waves take every 4th split, 8 loads are in flight per lane, and the sample's
// GroupNorm scale is applied on the fly from an LDS table.  Waves are folded through LDS in fixed order (deterministic).
__device__ __forceinline__ void wgrad_fold_body(const WFoldParams& p, int bid, float* sh) {
  const int ntap = p.f.ntap, ci_real = p.f.ci_real, ci_pad = p.f.ci_pad, co_real = p.f.co_real;
  const int R = wfold_rows(ci_pad);
  const int rgs = (co_real + R - 1) / R;
  const int tap = bid / rgs, rg = bid - tap * rgs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool norm = p.mr != nullptr;
  float* sc = sh;                          // [n][ci_pad] scale
  float* shc = sc + p.n * ci_pad;          // [n][ci_pad] shift
  float* sdy = shc + p.n * ci_pad;         // [n][R] sum of dy over the voxels whose tap is in bounds
  f32x4* red = reinterpret_cast<f32x4*>(sdy + ((p.n * R + R + 3) & ~3));  // [4 waves][64 lanes]
  const int cg = norm ? ci_real / p.groups : 1;
  for (int i = tid; i < p.n * ci_pad; i += 256) {
    const int n = i / ci_pad, ci = i - n * ci_pad;
    float s = (ci < ci_real) ? 1.f : 0.f, t = 0.f;
    if (norm && ci < ci_real) {
      const int gi = ci / cg;
      const float mu = p.mr[((long)n * p.groups + gi) * 2], r = p.mr[((long)n * p.groups + gi) * 2 + 1];
      s = r * p.gamma[ci];
      t = p.beta[ci] - mu * s;
    }
    sc[i] = s;
    shc[i] = t;
  }
  // per (sample, row): lanes = boundary classes, one shuffle reduction each
  if (p.csum && norm) {
    const bool inb = tap_inb_class(tap, lane, p.f);
    for (int q = wave; q < p.n * R; q += 4) {
      const int n = q / R, co = rg * R + (q - n * R);
      float v = (inb && co < co_real) ? p.csum[((long)n * 64 + lane) * p.csum_c + co] : 0.f;
      v = wave_sum(v);
      if (lane == 0) sdy[q] = v;
    }
  }
  if (p.dbias && tap == 0) {
    for (int r = wave; r < R; r += 4) {
      const int co = rg * R + r;
      float v = 0.f;
      if (co < co_real) {
        if (p.tg) {   // whole-volume sums (slot 0) of every workgroup partial, lanes over (sample, partial)
          for (int i = lane; i < p.n * p.nsplit; i += 64) v += p.tg[(long)i * 27 * 32 + co];
        } else {
          for (int n = 0; n < p.n; ++n) v += p.csum[((long)n * 64 + lane) * p.csum_c + co];
        }
      }
      v = wave_sum(v);
      if (lane == 0 && co < co_real) {
        if (p.accumulate) p.dbias[co] += v; else p.dbias[co] = v;
      }
    }
  }
  __syncthreads();
  const int r = (lane * 4) / ci_pad, ci = (lane * 4) - r * ci_pad, co = rg * R + r;
  const long slab = (long)ntap * p.co32 * ci_pad;
  const float* base = p.gp + ((long)tap * p.co32 + rg * R) * ci_pad + lane * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int n = 0; n < p.n; ++n) {
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc + n * ci_pad + ci);
    const float* bn = base + (long)n * p.nsplit * slab;
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
    int s = wave;
    for (; s + 28 < p.nsplit; s += 32) {
      f32x4 g[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) g[k] = *reinterpret_cast<const f32x4*>(bn + (long)(s + 4 * k) * slab);
#pragma unroll
      for (int k = 0; k < 8; ++k) part += g[k];
    }
    for (; s < p.nsplit; s += 4) part += *reinterpret_cast<const f32x4*>(bn + (long)s * slab);
    acc += s4 * part;
  }
  red[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave == 0) {
    f32x4 a = red[lane];
#pragma unroll
    for (int k = 1; k < 4; ++k) a += red[k * 64 + lane];
    if (norm && p.csum)
      for (int n = 0; n < p.n; ++n) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(shc + n * ci_pad + ci);
        a += t4 * sdy[n * R + r];
      }
    if (co < co_real) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (ci + j < ci_real) {
          float* o = p.dw + ((long)co * p.f.ci_total + p.f.ci_off + ci + j) * ntap + tap;
          if (p.accumulate) *o += a[j]; else *o = a[j];
        }
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_fold_kernel(WFoldParams p) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  wgrad_fold_body(p, blockIdx.x, sh);
}

static int fill_wfold(WFoldParams& p, const float* gp, int nsplit, const float* csum, const float* mr, const float* gamma,
                      const float* beta, int groups, const RtpConvGeom* g, int ci_real, int co_real, float* dw,
                      float* dbias, int accumulate) {
  int rc = fill_fold(p.f, g, ci_real, co_real);
  if (rc) return rc;
  if (!gp || !dw || nsplit < 1) return RTP_ERR_SHAPE;
  p.tg = nullptr;
  if ((mr || dbias) && !csum) return RTP_ERR_SHAPE;
  if (mr && (!gamma || !beta || groups < 1 || ci_real % groups)) return RTP_ERR_SHAPE;
  p.gp = gp; p.nsplit = nsplit; p.csum = csum; p.mr = mr; p.gamma = gamma; p.beta = beta;
  p.groups = groups; p.n = g->n; p.co32 = (g->co + 31) / 32 * 32; p.csum_c = p.co32;
  p.dw = dw; p.dbias = dbias; p.accumulate = accumulate;
  if (p.f.ci_pad != 32 && p.f.ci_pad != 64 && p.f.ci_pad != 128 && p.f.ci_pad != 256) return RTP_ERR_UNSUPPORTED;
  if (wfold_shm(p) > 60 * 1024) return RTP_ERR_UNSUPPORTED;
  return RTP_OK;
}

extern "C" int rtp_wgrad_fold(const float* gp, int nsplit, const float* csum, const float* mr, const float* gamma,
                              const float* beta, int groups, const RtpConvGeom* g, int ci_real, int co_real, float* dw,
                              float* dbias, int accumulate, void* stream) {
  WFoldParams p;
  int rc = fill_wfold(p, gp, nsplit, csum, mr, gamma, beta, groups, g, ci_real, co_real, dw, dbias, accumulate);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  hipLaunchKernelGGL(wgrad_fold_kernel, dim3(wfold_blocks(p)), dim3(256), wfold_shm(p), s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// Deferred tail of the backward sweep.  Class-sum reductions, slab folds and GroupNorm parameter gradients only feed
// the optimiser, and each is a few microseconds of work: launched per layer they are ~110 latency-bound launches per
// step.  The plan instead records one descriptor per item and runs them as ONE grid per dependency stage (class
// reductions + GroupNorm parameter sums, then the folds): block -> (descriptor, local block) by binary search.
// ------------------------------------------------------------------------------------------------
struct GnParParams { const float* part; float* dgamma; float* dbeta; int n, c, accumulate; };
// tap-major fp32 copy of a conv's master weights: wt[tap][co_pad][ci] = w[co][ci][tap] (rows co >= co_real are zeros) -- what
// rtp_conv_gn_fused's prologue reads with coalesced 16-byte loads, in the order it writes the LDS image
struct PackWtParams { const float* w; float* wt; int co_real, co_pad, ci, ntap; };
enum { TAIL_CLASS_REDUCE = 0, TAIL_WGRAD_FOLD = 1, TAIL_GN_PARAM = 2, TAIL_FOLD_FWD = 3, TAIL_PACK_WT = 4 };
struct RtpTailDesc {
  int kind, blocks;
  union { WFoldParams wf; ClsRedParams cr; GnParParams gp; FoldParams ff; PackWtParams pw; } u;
};

__device__ __forceinline__ void pack_wt_body(const PackWtParams& p, int bid) {
  const int i = bid * 256 + threadIdx.x;   // one output element: (tap, co, ci), ci fastest
  const int total = p.ntap * p.co_pad * p.ci;
  if (i >= total) return;
  const int ci = i % p.ci, co = (i / p.ci) % p.co_pad, tap = i / (p.ci * p.co_pad);
  p.wt[i] = co < p.co_real ? p.w[((long)co * p.ci + ci) * p.ntap + tap] : 0.f;
}

__device__ __forceinline__ void gn_param_body(const GnParParams& p, int bid) {
  const int t = bid * 256 + threadIdx.x;
  if (t >= 2 * p.c) return;
  const int ch = t >> 1, which = t & 1;
  float acc = 0.f;
  for (int i = 0; i < p.n; ++i) acc += p.part[((long)i * p.c + ch) * 2 + which];
  float* o = which ? p.dbeta + ch : p.dgamma + ch;
  if (p.accumulate) *o += acc; else *o = acc;
}

__global__ __launch_bounds__(256) void tail_kernel(const RtpTailDesc* descs, const int* starts, int count) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int bid = blockIdx.x;
  int lo = 0, hi = count;  // starts[lo] <= bid < starts[lo + 1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (starts[mid] <= bid) lo = mid; else hi = mid;
  }
  const RtpTailDesc& d = descs[lo];
  const int b = bid - starts[lo];
  if (d.kind == TAIL_WGRAD_FOLD) {
    const WFoldParams p = d.u.wf;
    wgrad_fold_body(p, b, sh);
  } else if (d.kind == TAIL_CLASS_REDUCE) {
    const ClsRedParams p = d.u.cr;
    class_reduce_body(p, b);
  } else if (d.kind == TAIL_FOLD_FWD) {
    const FoldParams p = d.u.ff;
    fold_fwd_body(p, b % p.nw, b / p.nw, sh);
  } else if (d.kind == TAIL_PACK_WT) {
    const PackWtParams p = d.u.pw;
    pack_wt_body(p, b);
  } else {
    const GnParParams p = d.u.gp;
    gn_param_body(p, b);
  }
}

extern "C" int rtp_tail_desc_bytes(void) { return (int)sizeof(RtpTailDesc); }

extern "C" int rtp_tail_desc_class_reduce(const float* scratch, int nsplit, int n, int c, float* out, void* desc,
                                          int* blocks, int* shm_bytes) {
  if (!scratch || !out || !desc || nsplit < 1 || n < 1) return RTP_ERR_SHAPE;
  if (c % 8) return RTP_ERR_ALIGN;
  RtpTailDesc d;
  memset(&d, 0, sizeof(d));
  d.kind = TAIL_CLASS_REDUCE;
  d.u.cr.part = scratch; d.u.cr.out = out; d.u.cr.nsplit = nsplit; d.u.cr.per_n = 64 * c; d.u.cr.n = n;
  d.blocks = rtp_div_up((long)n * 64 * c / 4, 256);
  memcpy(desc, &d, sizeof(d));
  if (blocks) *blocks = d.blocks;
  if (shm_bytes) *shm_bytes = 0;
  return RTP_OK;
}

extern "C" int rtp_tail_desc_wgrad_fold(const float* gp, int nsplit, const float* csum, const float* mr,
                                        const float* gamma, const float* beta, int groups, const RtpConvGeom* g,
                                        int ci_real, int co_real, float* dw, float* dbias, int accumulate, void* desc,
                                        int* blocks, int* shm_bytes) {
  if (!desc) return RTP_ERR_SHAPE;
  RtpTailDesc d;
  memset(&d, 0, sizeof(d));
  d.kind = TAIL_WGRAD_FOLD;
  int rc = fill_wfold(d.u.wf, gp, nsplit, csum, mr, gamma, beta, groups, g, ci_real, co_real, dw, dbias, accumulate);
  if (rc) return rc;
  d.blocks = wfold_blocks(d.u.wf);
  memcpy(desc, &d, sizeof(d));
  if (blocks) *blocks = d.blocks;
  if (shm_bytes) *shm_bytes = (int)wfold_shm(d.u.wf);
  return RTP_OK;
}

/* rtp_tail_desc_wgrad_fold for a conv with bias and WITHOUT GroupNorm whose weight gradient came from rtp_wgrad_tg: the bias
 * gradient is read off the subset-sum partials tg [n][nsplit][27][32] (slot 0 = the whole volume) instead of class sums. */
extern "C" int rtp_tail_desc_wgrad_fold_tg(const float* gp, int nsplit, const float* tg, const RtpConvGeom* g, int ci_real, int co_real,
                                           float* dw, float* dbias, int accumulate, void* desc, int* blocks, int* shm_bytes) {
  if (!desc || !tg || !dbias || !g) return RTP_ERR_SHAPE;
  if ((g->co + 31) / 32 * 32 != 32) return RTP_ERR_UNSUPPORTED;
  RtpTailDesc d;
  memset(&d, 0, sizeof(d));
  d.kind = TAIL_WGRAD_FOLD;
  int rc = fill_wfold(d.u.wf, gp, nsplit, nullptr, nullptr, nullptr, nullptr, 1, g, ci_real, co_real, dw, nullptr, accumulate);
  if (rc) return rc;
  d.u.wf.dbias = dbias;
  d.u.wf.tg = tg;
  d.blocks = wfold_blocks(d.u.wf);
  memcpy(desc, &d, sizeof(d));
  if (blocks) *blocks = d.blocks;
  if (shm_bytes) *shm_bytes = (int)wfold_shm(d.u.wf);
  return RTP_OK;
}

extern "C" int rtp_tail_desc_gn_param(const float* coeff, int n, int c, float* dgamma, float* dbeta, int accumulate,
                                      void* desc, int* blocks, int* shm_bytes) {
  if (!coeff || !dgamma || !dbeta || !desc || n < 1 || c < 1) return RTP_ERR_SHAPE;
  RtpTailDesc d;
  memset(&d, 0, sizeof(d));
  d.kind = TAIL_GN_PARAM;
  d.u.gp.part = coeff + (long)n * c * 3;  // the scratch rtp_gn_bwd_coeffs leaves behind the coefficients
  d.u.gp.dgamma = dgamma; d.u.gp.dbeta = dbeta; d.u.gp.n = n; d.u.gp.c = c; d.u.gp.accumulate = accumulate;
  d.blocks = rtp_div_up(2 * c, 256);
  memcpy(desc, &d, sizeof(d));
  if (blocks) *blocks = d.blocks;
  if (shm_bytes) *shm_bytes = 0;
  return RTP_OK;
}

extern "C" int rtp_tail_desc_fold_fwd(const float* w, const float* bias, const float* gamma, const float* beta,
                                      const float* stats, int nsplit, int groups, float eps, const RtpConvGeom* g,
                                      int ci_real, int co_real, void* wf, float* btab, float* mr, void* wd, void* desc,
                                      int* blocks, int* shm_bytes) {
  if (!desc) return RTP_ERR_SHAPE;
  RtpTailDesc d;
  memset(&d, 0, sizeof(d));
  d.kind = TAIL_FOLD_FWD;
  int nw, ny;
  int rc = fill_fold_fwd(d.u.ff, w, bias, gamma, beta, stats, nsplit, groups, eps, g, ci_real, co_real, wf, btab, mr, wd, &nw, &ny);
  if (rc) return rc;
  d.u.ff.nw = nw;
  d.blocks = nw * ny;
  memcpy(desc, &d, sizeof(d));
  if (blocks) *blocks = d.blocks;
  if (shm_bytes) *shm_bytes = (int)fold_shm(d.u.ff);
  return RTP_OK;
}

extern "C" int rtp_tail_desc_pack_wt(const float* w, int co_real, int co_pad, int ci, int ntap, float* wt, void* desc, int* blocks,
                                     int* shm_bytes) {
  if (!desc || !w || !wt || co_real < 1 || co_real > co_pad || ci < 1 || ntap < 1) return RTP_ERR_SHAPE;
  RtpTailDesc d;
  memset(&d, 0, sizeof(d));
  d.kind = TAIL_PACK_WT;
  d.u.pw.w = w; d.u.pw.wt = wt; d.u.pw.co_real = co_real; d.u.pw.co_pad = co_pad; d.u.pw.ci = ci; d.u.pw.ntap = ntap;
  d.blocks = (ntap * co_pad * ci + 255) / 256;
  memcpy(desc, &d, sizeof(d));
  if (blocks) *blocks = d.blocks;
  if (shm_bytes) *shm_bytes = 0;
  return RTP_OK;
}

extern "C" int rtp_tail_launch(const void* descs, const int* block_start, int count, int total_blocks, int shm_bytes,
                               void* stream) {
  if (!descs || !block_start || count < 1 || total_blocks < 1 || shm_bytes < 0 || shm_bytes > 60 * 1024) return RTP_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  hipLaunchKernelGGL(tail_kernel, dim3(total_blocks), dim3(256), (size_t)shm_bytes, s, (const RtpTailDesc*)descs, block_start, count);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
