// GroupNorm statistics, GroupNorm->weight folding, and the small "algebra" kernels of the backward pass.
//
// GroupNorm(8,C) in front of every backbone conv (hr_util/common.py:57, hr3d.py:83,147,168,297,323) is never
// materialised: x_hat[c] = x[c]*scale[n,c] + shift[n,c] is folded into per-sample bf16 weights
// (W*scale) and a per-boundary-class bias (sum over in-bounds taps of W.shift), because zero padding is
// applied AFTER the norm.  The backward pass undoes the fold with per-(n,c) sums (P,Q) and per-class sums.
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
  if (c % 8 || c > 256 || (256 % (c / 8))) return RTP_ERR_UNSUPPORTED;
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
};

__device__ __forceinline__ bool tap_inb_class(int tap, int cls, const FoldParams& p) {
  const int ks = p.ks;
  const int kz = tap / (ks * ks), ky = (tap / ks) % ks, kx = tap % ks;
  return tap_inb_1d(kz, cls & 1, (cls >> 1) & 1, p.dov, p.di, p.stride, p.pad) &&
         tap_inb_1d(ky, (cls >> 2) & 1, (cls >> 3) & 1, p.ho, p.hi, p.stride, p.pad) &&
         tap_inb_1d(kx, (cls >> 4) & 1, (cls >> 5) & 1, p.wo, p.wi, p.stride, p.pad);
}

#define FOLD_COS 4  // output channels per block: every block owns all taps of its couts, so the bias table needs no cross-block sum

__global__ __launch_bounds__(256) void fold_fwd_kernel(FoldParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* scale = smem;                 // [ci_pad]
  float* shift = scale + p.ci_pad;     // [ci_pad]
  float* T = shift + p.ci_pad;         // [FOLD_COS][ntap]
  const int n = blockIdx.x, tid = threadIdx.x;
  const int co0 = blockIdx.y * FOLD_COS;
  const bool norm = p.stats != nullptr;
  for (int c = tid; c < p.ci_pad; c += 256) { scale[c] = (c < p.ci_real && !norm) ? 1.f : 0.f; shift[c] = 0.f; }
  __syncthreads();
  if (norm) {
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
      if (p.mr && blockIdx.y == 0 && lane == 0) {
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
  // folded weights  wf[n][tap][co][ci] for co in [co0, co0+FOLD_COS)
  bf16_t* wf = p.wf + (long)n * p.ntap * p.co_pad * p.ci_pad;
  const int total = p.ntap * FOLD_COS * p.ci_pad;
  for (int i = tid; i < total; i += 256) {
    const int ci = i % p.ci_pad, col = (i / p.ci_pad) % FOLD_COS, tap = i / (p.ci_pad * FOLD_COS);
    const int co = co0 + col;
    float v = 0.f;
    if (ci < p.ci_real && co < p.co_real) v = p.w[((long)co * p.ci_total + p.ci_off + ci) * p.ntap + tap] * scale[ci];
    if (co < p.co_pad) wf[((long)tap * p.co_pad + co) * p.ci_pad + ci] = f2bf(v);
    if (p.wd && n == 0 && co < p.cok) {  // the un-folded weights, transposed for the data-gradient conv
      const float w0 = (ci < p.ci_real && co < p.co_real) ? p.w[((long)co * p.ci_total + p.ci_off + ci) * p.ntap + tap] : 0.f;
      p.wd[((long)tap * p.ci_pad + ci) * p.cok + co] = f2bf(w0);
    }
  }
  if (!p.btab) return;
  // T[col][tap] = sum_ci w*shift   (64 lanes per dot product would be overkill: ci <= 256, ntap*FOLD_COS <= 108 rows)
  for (int i = tid; i < FOLD_COS * p.ntap; i += 256) {
    const int tap = i % p.ntap, co = co0 + i / p.ntap;
    float acc = 0.f;
    if (norm && co < p.co_real)
      for (int ci = 0; ci < p.ci_real; ++ci) acc += p.w[((long)co * p.ci_total + p.ci_off + ci) * p.ntap + tap] * shift[ci];
    T[i] = acc;
  }
  __syncthreads();
  float* bt = p.btab + (long)n * 64 * p.co_pad;
  for (int i = tid; i < 64 * FOLD_COS; i += 256) {
    const int col = i % FOLD_COS, cls = i / FOLD_COS, co = co0 + col;
    float acc = (p.bias && co < p.co_real) ? p.bias[co] : 0.f;
    if (norm)
      for (int tap = 0; tap < p.ntap; ++tap)
        if (tap_inb_class(tap, cls, p)) acc += T[col * p.ntap + tap];
    if (co < p.co_pad) bt[cls * p.co_pad + co] = acc;
  }
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

extern "C" int rtp_fold_fwd(const float* w, const float* bias, const float* gamma, const float* beta,
                            const float* stats, int nsplit, int groups, float eps, const RtpConvGeom* g, int ci_real,
                            int co_real, void* wf, float* btab, float* mr, void* wd, void* stream) {
  FoldParams p;
  int rc = fill_fold(p, g, ci_real, co_real);
  if (rc) return rc;
  if (stats && (!gamma || !beta || groups < 1 || ci_real % groups)) return RTP_ERR_SHAPE;
  p.w = w; p.bias = bias; p.gamma = gamma; p.beta = beta; p.stats = stats;
  p.nsplit = nsplit; p.groups = groups; p.eps = eps;
  p.wf = (bf16_t*)wf; p.btab = btab; p.mr = mr;
  p.wd = (bf16_t*)wd; p.cok = (g->co + 31) / 32 * 32;
  const int nw = stats ? g->n : 1;
  const size_t shm = sizeof(float) * (2 * p.ci_pad + (size_t)FOLD_COS * p.ntap);
  if (p.co_pad % FOLD_COS) return RTP_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  hipLaunchKernelGGL(fold_fwd_kernel, dim3(nw, (wd ? p.cok : p.co_pad) / FOLD_COS), dim3(256), shm, s, p);
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
  const int tid = threadIdx.x, i = blockIdx.x;
  const int cg = c / groups;
  const float m = (float)cg * (float)vox;
  float pc = 0.f, qc = 0.f, mu = 0.f, r = 0.f, gam = 0.f;
  if (tid < c) {
    for (int s = 0; s < nsplit; ++s) {
      const float* q = pq + (((long)i * nsplit + s) * c + tid) * 2;
      pc += q[0];
      qc += q[1];
    }
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
  hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((2 * c + 255) / 256), dim3(256), 0, s, part, n, c, dgamma, dbeta, accumulate);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_class_sums : out[n][64][c]  (partials [n][nsplit][64][c] in `scratch`, then a fixed-order reduction)
// One wave per x-row: the (z,y) flags are row-uniform, only x==0 / x==W-1 differ, so lanes accumulate three
// register sets (interior / first / last) and touch the LDS buckets once per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void class_sums_kernel(const bf16_t* g, int cs, int co, int c, int D, int H, int W,
                                                         int nsplit, float* part) {
  extern __shared__ __attribute__((aligned(16))) float cls_sum[];  // [64][c]
  const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long vox = (long)D * H * W;
  for (int i = tid; i < 64 * c; i += 256) cls_sum[i] = 0.f;
  __syncthreads();
  const int cpv = c >> 3;           // 4, 8, 16 or 32: divides 64
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
    for (int i = lane; i < W * cpv; i += 64) {
      const int x = i / cpv;
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
#pragma unroll
    for (int j = 0; j < 8; ++j)
      for (int o = 32; o >= cpv; o >>= 1) {
        a_in[j] += __shfl_xor(a_in[j], o, 64);
        a_f[j] += __shfl_xor(a_f[j], o, 64);
        a_l[j] += __shfl_xor(a_l[j], o, 64);
      }
    if (lane < cpv) {
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

__global__ void class_sums_final(const float* part, int nsplit, int per_n, float* out) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;  // over n*64*c
  const long n = i / per_n, r = i - n * per_n;
  float acc = 0.f;
  for (int s = 0; s < nsplit; ++s) acc += part[(n * nsplit + s) * per_n + r];
  out[i] = acc;
}

extern "C" int rtp_class_sums(const RtpAct* gy, int n, int d, int h, int w, int nsplit, float* scratch, float* out,
                              void* stream) {
  if (!gy || !out || !scratch) return RTP_ERR_SHAPE;
  const int c = gy->c;
  if (c % 8 || c > 256 || (64 % (c / 8))) return RTP_ERR_UNSUPPORTED;
  if ((gy->cs % 8) || (gy->co % 8)) return RTP_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  hipLaunchKernelGGL(class_sums_kernel, dim3(nsplit, n), dim3(256), sizeof(float) * 64 * c, s, (const bf16_t*)gy->ptr,
                     gy->cs, gy->co, c, d, h, w, nsplit, scratch);
  const long total = (long)n * 64 * c;  // multiple of 256
  hipLaunchKernelGGL(class_sums_final, dim3((int)(total / 256)), dim3(256), 0, s, scratch, nsplit, 64 * c, out);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

extern "C" int rtp_class_sums_reduce(const float* scratch, int nsplit, int n, int c, float* out, void* stream) {
  if (!scratch || !out || nsplit < 1) return RTP_ERR_SHAPE;
  if (c % 8) return RTP_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  const long total = (long)n * 64 * c;  // multiple of 256
  hipLaunchKernelGGL(class_sums_final, dim3((int)(total / 256)), dim3(256), 0, s, scratch, nsplit, 64 * c, out);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_wgrad_fold : one block per (co, tap); 8 slab groups x 32 input channels per pass
// ------------------------------------------------------------------------------------------------
struct WFoldParams {
  const float* gp; int nsplit; const float* csum; const float* mr; const float* gamma; const float* beta;
  int groups, n; FoldParams f; int co32, csum_c; float* dw; float* dbias; int accumulate;
};

// One block per (co, tap).  A thread owns 4 input channels (one 16-B load per slab row) and one of 256/(ci/4) slab
// groups; the (sample, split) slabs of a group are summed with the sample's GroupNorm scale applied on the fly, then the
// groups are folded through LDS in fixed order (deterministic).
__global__ __launch_bounds__(256) void wgrad_fold_kernel(WFoldParams p) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int ntap = p.f.ntap, ci_real = p.f.ci_real, ci_pad = p.f.ci_pad;
  const int co = blockIdx.x / ntap, tap = blockIdx.x - co * ntap, tid = threadIdx.x;
  float* sdy = sh;                        // [n]
  float* red = sdy + ((p.n + 3) & ~3);    // [groups_of_slabs][ci_real]
  const bool norm = p.mr != nullptr;
  for (int n = tid; n < p.n; n += 256) {
    float acc = 0.f;
    if (p.csum && norm)
      for (int cls = 0; cls < 64; ++cls)
        if (tap_inb_class(tap, cls, p.f)) acc += p.csum[((long)n * 64 + cls) * p.csum_c + co];
    sdy[n] = acc;
  }
  __syncthreads();
  if (p.dbias && tap == 0 && tid == 0) {
    float acc = 0.f;
    for (int n = 0; n < p.n; ++n)
      for (int cls = 0; cls < 64; ++cls) acc += p.csum[((long)n * 64 + cls) * p.csum_c + co];
    if (p.accumulate) p.dbias[co] += acc; else p.dbias[co] = acc;
  }
  const int quads = ci_real >> 2;          // 8, 16, 32 or 64 (ci_real is a multiple of 32)
  const int nsg = 256 / quads;             // slab groups
  const int cq = tid % quads, sg = tid / quads;
  const int cg = norm ? ci_real / p.groups : 1;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (sg < nsg) {
    const int total = p.n * p.nsplit;
    for (int i = sg; i < total; i += nsg) {
      const int n = i / p.nsplit;
      const f32x4 g = *reinterpret_cast<const f32x4*>(p.gp + (((long)i * ntap + tap) * p.co32 + co) * ci_pad + cq * 4);
      f32x4 sc = {1.f, 1.f, 1.f, 1.f};
      if (norm) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ci = cq * 4 + j, gi = ci / cg;
          sc[j] = p.mr[((long)n * p.groups + gi) * 2 + 1] * p.gamma[ci];
        }
      }
      acc += sc * g;
    }
    if (norm && sg == 0) {
      for (int n = 0; n < p.n; ++n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ci = cq * 4 + j, gi = ci / cg;
          const float mu = p.mr[((long)n * p.groups + gi) * 2], r = p.mr[((long)n * p.groups + gi) * 2 + 1];
          acc[j] += (p.beta[ci] - mu * r * p.gamma[ci]) * sdy[n];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[sg * ci_real + cq * 4 + j] = acc[j];
  }
  __syncthreads();
  for (int ci = tid; ci < ci_real; ci += 256) {
    float a = 0.f;
    for (int k = 0; k < nsg; ++k) a += red[k * ci_real + ci];
    float* o = p.dw + ((long)co * p.f.ci_total + p.f.ci_off + ci) * ntap + tap;
    if (p.accumulate) *o += a; else *o = a;
  }
}

extern "C" int rtp_wgrad_fold(const float* gp, int nsplit, const float* csum, const float* mr, const float* gamma,
                              const float* beta, int groups, const RtpConvGeom* g, int ci_real, int co_real, float* dw,
                              float* dbias, int accumulate, void* stream) {
  WFoldParams p;
  int rc = fill_fold(p.f, g, ci_real, co_real);
  if (rc) return rc;
  if ((mr || dbias) && !csum) return RTP_ERR_SHAPE;
  p.gp = gp; p.nsplit = nsplit; p.csum = csum; p.mr = mr; p.gamma = gamma; p.beta = beta;
  p.groups = groups; p.n = g->n; p.co32 = (g->co + 31) / 32 * 32; p.csum_c = p.co32;
  p.dw = dw; p.dbias = dbias; p.accumulate = accumulate;
  if (ci_real % 32 || ci_real > 256) return RTP_ERR_UNSUPPORTED;
  const size_t shm = sizeof(float) * ((size_t)((p.n + 3) & ~3) + (size_t)(256 / (ci_real / 4)) * ci_real);
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_NORM, s);
  hipLaunchKernelGGL(wgrad_fold_kernel, dim3(co_real * p.f.ntap), dim3(256), shm, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
