// HBM-bound point-wise kernels of the HRRadarPose path: gradient combine (ReLU mask + GroupNorm backward
// apply + residual fan-in), fuse-row sums with on-the-fly trilinear upsampling, the upsample adjoint, the
// Cin=1 stem conv, and NCDHW<->channels-last packing.  All move 16 bytes (8 bf16 channels) per lane.
#include <stdlib.h>

#include "rtp_common.h"
#include "rtp_prof.h"

// ------------------------------------------------------------------------------------------------
// rtp_grad_combine
// ------------------------------------------------------------------------------------------------
// lazy (rtp_grad_combine_cls_lazy): the GroupNorm-backward coefficients of a GN term are computed in the kernel's prologue from
// the statistics partials pq [n][pq_nsplit][c][2] the data-gradient launch left behind (P = sum dxhat, Q = sum dxhat * x) --
// the arithmetic of rtp_gn_bwd_coeffs, without a launch of its own between the data gradient and the fan-in pass.
struct CombTerm { const bf16_t* t; int cs, co; const float* coeff;
                  const float* pq; int pq_nsplit; const float* mr; const float* gamma; int groups; float* coef_out; };
struct CombParams {
  CombTerm terms[RTP_MAX_TERMS]; int nterms;
  const bf16_t* x; int x_cs, x_co;
  const bf16_t* relu; int r_cs, r_co;
  bf16_t* out; int o_cs, o_co;
  int c, n; long vox;
};

// Combine for CU_ items at once: every load of every item is issued before the first use (two phases), which is what
// keeps enough bytes in flight for an HBM-bound pass (one item at a time, a wave waited on each term in turn: 2.9 TB/s).
// cfs: GroupNorm-backward coefficients [term][c][3] for this sample (LDS or global).
#define CU_ 2
__device__ __forceinline__ void combine_items(const CombParams& p, const long (&vv)[CU_], const bool (&live)[CU_], int chunk,
                                              const float* const (&cfs)[RTP_MAX_TERMS], bool same_rx, bf16x8 (&o)[CU_]) {
  bf16x8 tv[CU_][RTP_MAX_TERMS], xv[CU_], rv[CU_];
  bool need_x = false;
  for (int k = 0; k < p.nterms; ++k) need_x |= p.terms[k].coeff != nullptr;
#pragma unroll
  for (int u = 0; u < CU_; ++u) {
    xv[u] = rv[u] = zero_bf16x8();
    if (!live[u]) continue;
#pragma unroll
    for (int k = 0; k < RTP_MAX_TERMS; ++k)
      if (k < p.nterms) tv[u][k] = ld_bf16x8(p.terms[k].t + vv[u] * p.terms[k].cs + p.terms[k].co + chunk * 8);
    if (need_x) xv[u] = ld_bf16x8(p.x + vv[u] * p.x_cs + p.x_co + chunk * 8);
    if (p.relu && !(same_rx && need_x)) rv[u] = ld_bf16x8(p.relu + vv[u] * p.r_cs + p.r_co + chunk * 8);
  }
#pragma unroll
  for (int u = 0; u < CU_; ++u) {
    if (!live[u]) continue;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < RTP_MAX_TERMS; ++k) {
      if (k >= p.nterms) break;
      if (p.terms[k].coeff) {
        const float* cf = cfs[k] + chunk * 8 * 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += cf[j * 3] * bf2f(tv[u][k][j]) + cf[j * 3 + 1] * bf2f(xv[u][j]) + cf[j * 3 + 2];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += bf2f(tv[u][k][j]);
      }
    }
    if (p.relu) {
      const bf16x8 r = (same_rx && need_x) ? xv[u] : rv[u];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = bf2f(r[j]) > 0.f ? acc[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) o[u][j] = f2bf(acc[j]);
  }
}

__global__ __launch_bounds__(256) void grad_combine_kernel(CombParams p) {
  const int cpv = p.c >> 3;
  const long total = (long)p.n * p.vox * cpv;
  const bool same_rx = p.relu == p.x && p.r_cs == p.x_cs && p.r_co == p.x_co;
  const long stride = (long)gridDim.x * 256;
  for (long i0 = blockIdx.x * 256L + threadIdx.x; i0 < total; i0 += CU_ * stride) {
    long vv[CU_];
    bool live[CU_];
    const int ck = (int)(i0 % cpv);  // stride is a multiple of cpv: every item of this lane has the same chunk
    int n_[CU_];
#pragma unroll
    for (int u = 0; u < CU_; ++u) {
      const long i = i0 + u * stride;
      live[u] = i < total;
      vv[u] = live[u] ? i / cpv : 0;
      n_[u] = (int)(vv[u] / p.vox);
    }
    // coefficients depend on the sample; a lane's items may straddle a sample boundary, so they go one by one
    bf16x8 o[CU_];
    if (n_[0] == n_[CU_ - 1]) {
      const float* cfs[RTP_MAX_TERMS];
#pragma unroll
      for (int k = 0; k < RTP_MAX_TERMS; ++k)
        cfs[k] = (k < p.nterms && p.terms[k].coeff) ? p.terms[k].coeff + (long)n_[0] * p.c * 3 : nullptr;
      combine_items(p, vv, live, ck, cfs, same_rx, o);
    } else {
#pragma unroll
      for (int u = 0; u < CU_; ++u) {
        long v1[CU_];
        bool l1[CU_];
        bf16x8 o1[CU_];
#pragma unroll
        for (int q = 0; q < CU_; ++q) { v1[q] = vv[u]; l1[q] = (q == 0) && live[u]; }
        const float* cfs[RTP_MAX_TERMS];
#pragma unroll
        for (int k = 0; k < RTP_MAX_TERMS; ++k)
          cfs[k] = (k < p.nterms && p.terms[k].coeff) ? p.terms[k].coeff + (long)n_[u] * p.c * 3 : nullptr;
        combine_items(p, v1, l1, ck, cfs, same_rx, o1);
        o[u] = o1[0];
      }
    }
#pragma unroll
    for (int u = 0; u < CU_; ++u)
      if (live[u]) st_bf16x8(p.out + vv[u] * p.o_cs + p.o_co + ck * 8, o[u]);
  }
}

static inline int grid_for(long items) {
  long b = (items + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

extern "C" int rtp_grad_combine(const RtpTerm* terms, int nterms, const RtpAct* x, const RtpAct* relu_src,
                                const RtpAct* out, int n, long vox, void* stream) {
  if (!terms || nterms < 1 || nterms > RTP_MAX_TERMS || !out) return RTP_ERR_SHAPE;
  CombParams p;
  p.nterms = nterms;
  p.c = out->c;
  if (p.c % 8 || (out->cs % 8) || (out->co % 8)) return RTP_ERR_ALIGN;
  bool need_x = false;
  for (int k = 0; k < nterms; ++k) {
    if (terms[k].t.c != p.c || (terms[k].t.cs % 8) || (terms[k].t.co % 8)) return RTP_ERR_ALIGN;
    p.terms[k] = CombTerm{(const bf16_t*)terms[k].t.ptr, terms[k].t.cs, terms[k].t.co, terms[k].coeff, nullptr, 0, nullptr, nullptr, 1, nullptr};
    need_x |= terms[k].coeff != nullptr;
  }
  if (need_x && (!x || (x->cs % 8) || (x->co % 8))) return RTP_ERR_SHAPE;
  p.x = x ? (const bf16_t*)x->ptr : nullptr; p.x_cs = x ? x->cs : 0; p.x_co = x ? x->co : 0;
  p.relu = relu_src ? (const bf16_t*)relu_src->ptr : nullptr;
  p.r_cs = relu_src ? relu_src->cs : 0; p.r_co = relu_src ? relu_src->co : 0;
  p.out = (bf16_t*)out->ptr; p.o_cs = out->cs; p.o_co = out->co;
  p.n = n; p.vox = vox;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  hipLaunchKernelGGL(grad_combine_kernel, dim3(grid_for((long)n * vox * (p.c / 8))), dim3(256), 0, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_grad_combine_cls : the same combine, plus per-boundary-class channel sums of the (bf16-rounded) result, so the
// weight-gradient un-fold needs no separate scan of the gradient tensor.  One wave per x-row: the (z,y) flags are
// row-uniform and only x==0 / x==W-1 differ, so a lane keeps three register sets (interior / first / last) that are
// flushed into the wave's own LDS slice when the row class changes -- no atomics, fixed summation order.
// GroupNorm-backward coefficients of the block's sample sit in LDS.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_combine_cls_kernel(CombParams p, int D, int H, int W, int nsplit,
                                                               float* part) {
  extern __shared__ __attribute__((aligned(16))) float cls_lds[];  // [4][64][c] class sums, then [nterms][c][3] coeffs
  const int n = blockIdx.y, s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = p.c, cpv = c >> 3, chunk = lane % cpv;
  float* mine = cls_lds + wave * 64 * c;
  float* cfs = cls_lds + 4 * 64 * c;
  for (int k = 0; k < p.nterms; ++k) {
    const CombTerm& t = p.terms[k];
    if (t.pq) {   // kernel-uniform: lazy coefficients of this term for sample n (the class-sum region is scratch until zeroed below)
      float2* part_pq = reinterpret_cast<float2*>(cls_lds);   // [256]
      float* P = cls_lds + 512;                               // [c]
      float* Q = P + 64;                                      // [c]
      float* S1 = Q + 64;                                     // [groups]
      float* S2 = S1 + 64;
      const int cg = c / t.groups;
      const float m = (float)cg * (float)((long)D * H * W);
      float pc = 0.f, qc = 0.f, mu = 0.f, r = 0.f, gam = 0.f;
      {
        const int np = 256 / c, ch = tid % c, pt = tid / c;
        float2 a = make_float2(0.f, 0.f);
        if (pt < np)
          for (int s_ = pt; s_ < t.pq_nsplit; s_ += np) {
            const float2 q2 = *reinterpret_cast<const float2*>(t.pq + (((long)n * t.pq_nsplit + s_) * c + ch) * 2);
            a.x += q2.x;
            a.y += q2.y;
          }
        part_pq[tid] = a;
        __syncthreads();
        if (tid < c)
          for (int k2 = 0; k2 < np; ++k2) { pc += part_pq[k2 * c + tid].x; qc += part_pq[k2 * c + tid].y; }
      }
      if (tid < c) {
        const int g = tid / cg;
        mu = t.mr[((long)n * t.groups + g) * 2];
        r = t.mr[((long)n * t.groups + g) * 2 + 1];
        gam = t.gamma[tid];
        P[tid] = gam * pc;
        Q[tid] = gam * r * (qc - mu * pc);
        if (s == 0 && t.coef_out) {   // dgamma / dbeta partials for the deferred parameter sums (rtp_tail_desc_gn_param)
          float* pt2 = t.coef_out + (long)gridDim.y * c * 3 + ((long)n * c + tid) * 2;
          pt2[0] = r * (qc - mu * pc);
          pt2[1] = pc;
        }
      }
      __syncthreads();
      if (tid < t.groups) {
        float s1 = 0.f, s2 = 0.f;
        for (int k2 = tid * cg; k2 < (tid + 1) * cg; ++k2) { s1 += P[k2]; s2 += Q[k2]; }
        S1[tid] = s1;
        S2[tid] = s2;
      }
      __syncthreads();
      if (tid < c) {
        const int g = tid / cg;
        const float a0 = r * gam, b0 = -r * r * S2[g] / m, c0 = -r * S1[g] / m + r * r * mu * S2[g] / m;
        float* o = cfs + k * c * 3 + tid * 3;
        o[0] = a0; o[1] = b0; o[2] = c0;
        if (s == 0 && t.coef_out) {
          float* og = t.coef_out + ((long)n * c + tid) * 3;
          og[0] = a0; og[1] = b0; og[2] = c0;
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < 4 * 64 * c; i += 256) cls_lds[i] = 0.f;
  for (int k = 0; k < p.nterms; ++k)
    if (p.terms[k].coeff && !p.terms[k].pq)
      for (int i = tid; i < c * 3; i += 256) cfs[k * c * 3 + i] = p.terms[k].coeff[(long)n * c * 3 + i];
  __syncthreads();
  const bool same_rx = p.relu == p.x && p.r_cs == p.x_cs && p.r_co == p.x_co;
  const float* cfp[RTP_MAX_TERMS];
#pragma unroll
  for (int k = 0; k < RTP_MAX_TERMS; ++k) cfp[k] = cfs + k * c * 3;
  const long vox = (long)D * H * W;
  const int rows = D * H;
  const int rps = (rows + nsplit - 1) / nsplit;
  const int r0 = s * rps, r1 = (r0 + rps < rows) ? r0 + rps : rows;
  const int items = W * cpv;
  float a_in[8], a_f[8], a_l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a_in[j] = a_f[j] = a_l[j] = 0.f;
  int cur = -1;
  auto flush = [&](int czy) {
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
        mine[czy * c + chunk * 8 + j] += a_in[j];
        mine[cf * c + chunk * 8 + j] += a_f[j];
        if (W > 1) mine[cl * c + chunk * 8 + j] += a_l[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) a_in[j] = a_f[j] = a_l[j] = 0.f;
  };
  for (int r = r0 + wave; r < r1; r += 4) {
    const int z = r / H, y = r - z * H;
    const int czy = (z == 0) | ((z == D - 1) << 1) | ((y == 0) << 2) | ((y == H - 1) << 3);
    if (czy != cur) {
      if (cur >= 0) flush(cur);
      cur = czy;
    }
    const long vrow = (long)n * vox + (long)r * W;
    for (int i0 = lane; i0 < items; i0 += CU_ * 64) {
      long vv[CU_];
      bool live[CU_];
      int xs[CU_];
#pragma unroll
      for (int u = 0; u < CU_; ++u) {
        const int i = i0 + u * 64;
        live[u] = i < items;
        xs[u] = live[u] ? i / cpv : 0;
        vv[u] = vrow + xs[u];
      }
      bf16x8 o[CU_];
      combine_items(p, vv, live, chunk, cfp, same_rx, o);
#pragma unroll
      for (int u = 0; u < CU_; ++u) {
        if (!live[u]) continue;
        st_bf16x8(p.out + vv[u] * p.o_cs + p.o_co + chunk * 8, o[u]);
        const bool first = (xs[u] == 0), last = (xs[u] == W - 1) && !first;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = bf2f(o[u][j]);
          a_f[j] += first ? v : 0.f;
          a_l[j] += last ? v : 0.f;
          a_in[j] += (first || last) ? 0.f : v;
        }
      }
    }
  }
  if (cur >= 0) flush(cur);
  __syncthreads();
  float* o = part + ((long)n * nsplit + s) * 64 * c;
  for (int i = tid; i < 64 * c; i += 256)
    o[i] = (cls_lds[i] + cls_lds[64 * c + i]) + (cls_lds[2 * 64 * c + i] + cls_lds[3 * 64 * c + i]);
}

static int comb_params(CombParams& p, const RtpTerm* terms, int nterms, const RtpAct* x, const RtpAct* relu_src,
                       const RtpAct* out) {
  if (!terms || nterms < 1 || nterms > RTP_MAX_TERMS || !out) return RTP_ERR_SHAPE;
  p.nterms = nterms;
  p.c = out->c;
  if (p.c % 8 || (out->cs % 8) || (out->co % 8)) return RTP_ERR_ALIGN;
  bool need_x = false;
  for (int k = 0; k < nterms; ++k) {
    if (terms[k].t.c != p.c || (terms[k].t.cs % 8) || (terms[k].t.co % 8)) return RTP_ERR_ALIGN;
    p.terms[k] = CombTerm{(const bf16_t*)terms[k].t.ptr, terms[k].t.cs, terms[k].t.co, terms[k].coeff, nullptr, 0, nullptr, nullptr, 1, nullptr};
    need_x |= terms[k].coeff != nullptr;
  }
  if (need_x && (!x || (x->cs % 8) || (x->co % 8))) return RTP_ERR_SHAPE;
  p.x = x ? (const bf16_t*)x->ptr : nullptr; p.x_cs = x ? x->cs : 0; p.x_co = x ? x->co : 0;
  p.relu = relu_src ? (const bf16_t*)relu_src->ptr : nullptr;
  p.r_cs = relu_src ? relu_src->cs : 0; p.r_co = relu_src ? relu_src->co : 0;
  p.out = (bf16_t*)out->ptr; p.o_cs = out->cs; p.o_co = out->co;
  return RTP_OK;
}

extern "C" int rtp_grad_combine_cls_lazy(const RtpTerm* terms, int nterms, const RtpGnLazy* lazy, const RtpAct* x, const RtpAct* relu_src,
                                         const RtpAct* out, int n, int d, int h, int w, int nsplit, float* cls_scratch,
                                         void* stream);

extern "C" int rtp_grad_combine_cls(const RtpTerm* terms, int nterms, const RtpAct* x, const RtpAct* relu_src,
                                    const RtpAct* out, int n, int d, int h, int w, int nsplit, float* cls_scratch,
                                    void* stream) {
  return rtp_grad_combine_cls_lazy(terms, nterms, nullptr, x, relu_src, out, n, d, h, w, nsplit, cls_scratch, stream);
}

extern "C" int rtp_grad_combine_cls_lazy(const RtpTerm* terms, int nterms, const RtpGnLazy* lazy, const RtpAct* x, const RtpAct* relu_src,
                                         const RtpAct* out, int n, int d, int h, int w, int nsplit, float* cls_scratch,
                                         void* stream) {
  CombParams p;
  int rc = comb_params(p, terms, nterms, x, relu_src, out);
  if (rc != RTP_OK) return rc;
  for (int k = 0; lazy && k < nterms; ++k) {
    if (!lazy[k].pq) continue;
    if (!terms[k].coeff || !lazy[k].mr || !lazy[k].gamma || lazy[k].nsplit < 1 || lazy[k].groups < 1 || p.c % lazy[k].groups || p.c < 8)
      return RTP_ERR_SHAPE;
    p.terms[k].pq = lazy[k].pq; p.terms[k].pq_nsplit = lazy[k].nsplit; p.terms[k].mr = lazy[k].mr; p.terms[k].gamma = lazy[k].gamma;
    p.terms[k].groups = lazy[k].groups; p.terms[k].coef_out = const_cast<float*>(terms[k].coeff);
  }
  if (!cls_scratch || nsplit < 1) return RTP_ERR_SHAPE;
  if (p.c > 64 || (64 % (p.c / 8))) return RTP_ERR_UNSUPPORTED;
  p.n = n; p.vox = (long)d * h * w;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  const size_t lds = sizeof(float) * (4 * 64 * p.c + RTP_MAX_TERMS * p.c * 3);
  static bool attr_done[RTP_MAX_DEVICES] = {};
  if (rtp_once_per_device(attr_done)) {
    (void)hipFuncSetAttribute((const void*)grad_combine_cls_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  }
  hipLaunchKernelGGL(grad_combine_cls_kernel, dim3(nsplit, n), dim3(256), lds, s, p, d, h, w, nsplit, cls_scratch);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_fuse_sum : trilinear align_corners=True, PyTorch index rule (upsample_trilinear3d)
// ------------------------------------------------------------------------------------------------
struct FuseTerm { const bf16_t* t; int cs, co, d, h, w; float sz, sy, sx; int same; int span; };  // span: source voxels an FX-output run touches (rows kernel)  // s*: align_corners scale (I-1)/(O-1), computed in fp32 like ATen
struct FuseParams {
  FuseTerm terms[RTP_MAX_TERMS]; int nterms;
  const float* bias; bf16_t* out; int o_cs, o_co;
  int c, n, d, h, w, relu;
  // optional (rtp_fuse_sum_stats): per-channel (sum y, sum y^2) of the stored (rounded) row, one partial per block
  // [n][stat_blocks][c][2] -- what rtp_chan_stats would compute in a read pass of its own; the grid is then (stat_blocks, n)
  float* stat_out; int stat_blocks;
};

// statistics of a thread's items -> one partial per block.  Every item of a thread has the same 8-channel chunk (the index
// stride is a multiple of 256 and c / 8 divides 256), threads tid = chunk (mod c / 8) own the same channels.
__device__ __forceinline__ void stats_flush(float* stat_out, int stat_blocks, int c, int n, const float (&sy)[8], const float (&sq)[8]) {
  __shared__ float red[256][17];
  const int tid = threadIdx.x, cpv = c >> 3;
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[tid][j] = sy[j]; red[tid][8 + j] = sq[j]; }
  __syncthreads();
  for (int t = tid; t < cpv * 16; t += 256) {
    const int ck = t >> 4, j = t & 15;
    float a = 0.f;
    for (int m = ck; m < 256; m += cpv) a += red[m][j];   // fixed order
    stat_out[(((long)n * stat_blocks + blockIdx.x) * c + ck * 8 + (j & 7)) * 2 + (j >> 3)] = a;
  }
}
__device__ __forceinline__ void fuse_stats_flush(const FuseParams& p, int n, const float (&sy)[8], const float (&sq)[8]) {
  stats_flush(p.stat_out, p.stat_blocks, p.c, n, sy, sq);
}

__host__ __device__ __forceinline__ float ac_scale(int I, int O) { return (O > 1) ? (float)(I - 1) / (float)(O - 1) : 0.f; }

__device__ __forceinline__ void src_index(int o, int I, int O, float scale, int& i0, int& i1, float& l0, float& l1) {
  if (I == O) { i0 = i1 = o; l0 = 1.f; l1 = 0.f; return; }
  const float src = scale * (float)o;
  i0 = (int)src;
  i1 = i0 + ((i0 < I - 1) ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void fuse_sum_kernel(FuseParams p) {
  const int cpv = p.c >> 3;
  const long vox = (long)p.d * p.h * p.w;
  const bool per_sample = p.stat_out != nullptr;   // grid (stat_blocks, n): a block stays inside one sample
  const long total = (per_sample ? 1 : (long)p.n) * vox * cpv, ibase = per_sample ? (long)blockIdx.y * vox * cpv : 0;
  float sy[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long i_ = blockIdx.x * 256L + threadIdx.x; i_ < total; i_ += (long)gridDim.x * 256) {
    const long i = ibase + i_;
    const int ck = (int)(i % cpv);
    const long vv = i / cpv;
    const int n = (int)(vv / vox);
    int z, y, x;
    vox_decode((int)(vv - (long)n * vox), p.h, p.w, z, y, x);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = p.bias ? p.bias[ck * 8 + j] : 0.f;
    for (int k = 0; k < p.nterms; ++k) {   // NOT unrolled: six copies of this body spill the descriptor SGPRs
      const FuseTerm t = p.terms[k];
      if (t.same) {
        bf16x8 tv = ld_bf16x8(t.t + vv * t.cs + t.co + ck * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += bf2f(tv[j]);
      } else {
        int z0, z1, y0, y1, x0, x1;
        float lz0, lz1, ly0, ly1, lx0, lx1;
        src_index(z, t.d, p.d, t.sz, z0, z1, lz0, lz1);
        src_index(y, t.h, p.h, t.sy, y0, y1, ly0, ly1);
        src_index(x, t.w, p.w, t.sx, x0, x1, lx0, lx1);
        const long base = (long)n * t.d * t.h * t.w;
        float up[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) up[j] = 0.f;
#pragma unroll
        for (int cz = 0; cz < 2; ++cz)
#pragma unroll
          for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
              const float wgt = (cz ? lz1 : lz0) * (cy ? ly1 : ly0) * (cx ? lx1 : lx0);
              const long sv = base + ((long)(cz ? z1 : z0) * t.h + (cy ? y1 : y0)) * t.w + (cx ? x1 : x0);
              bf16x8 tv = ld_bf16x8(t.t + sv * t.cs + t.co + ck * 8);
#pragma unroll
              for (int j = 0; j < 8; ++j) up[j] += wgt * bf2f(tv[j]);
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += up[j];
      }
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(p.relu ? (acc[j] > 0.f ? acc[j] : 0.f) : acc[j]);
    st_bf16x8(p.out + vv * p.o_cs + p.o_co + ck * 8, o);
    if (per_sample) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float r = bf2f(o[j]); sy[j] += r; sq[j] += r * r; }
    }
  }
  if (per_sample) fuse_stats_flush(p, blockIdx.y, sy, sq);
}

// Row-run variant: a thread owns FX consecutive output voxels of one x-row (one 8-channel chunk).  For an up-sampled
// term the 2x2 (z,y) corner rows are blended ONCE into a short source row (<= FSPAN voxels: the x scale is < 1), from
// which the FX outputs are interpolated in registers -- 4*FSPAN/FX = 3 sixteen-byte gathers per output instead of 8
// (the point-per-thread kernel above is bound by L1/TA traffic: 26 gathers per output for a 4-branch fuse row).
#define FX 4
#define FSPAN 5
// One up-sampled term of a run of FX outputs (fuse_sum_rows_kernel): the 2x2 (z, y) corner rows blended into SPAN source columns,
// the FX outputs interpolated from them.  The kernel is VALU-bound on these terms (~100 vector instructions per source column).
template <int SPAN>
__device__ __forceinline__ void fuse_up_term(const FuseParams& p, const FuseTerm& t, int n, int z, int y, int xs, int ck, float (&acc)[FX][8]) {
  int z0, z1, y0, y1, xb, xdummy;
  float lz0, lz1, ly0, ly1, ldum0, ldum1;
  src_index(z, t.d, p.d, t.sz, z0, z1, lz0, lz1);
  src_index(y, t.h, p.h, t.sy, y0, y1, ly0, ly1);
  src_index(xs, t.w, p.w, t.sx, xb, xdummy, ldum0, ldum1);
  const long base = (long)n * t.d * t.h * t.w;
  float row[SPAN][8];
#pragma unroll
  for (int q = 0; q < SPAN; ++q)
#pragma unroll
    for (int c = 0; c < 8; ++c) row[q][c] = 0.f;
  int qo[SPAN];  // clamped source offsets: columns past the row end carry zero weight below, so any valid address does
#pragma unroll
  for (int q = 0; q < SPAN; ++q) qo[q] = ((xb + q < t.w) ? q : t.w - 1 - xb) * t.cs;
#pragma unroll
  for (int cz = 0; cz < 2; ++cz)
#pragma unroll
    for (int cy = 0; cy < 2; ++cy) {
      const float wgt = (cz ? lz1 : lz0) * (cy ? ly1 : ly0);
      const bf16_t* src = t.t + (base + ((long)(cz ? z1 : z0) * t.h + (cy ? y1 : y0)) * t.w + xb) * t.cs + t.co + ck * 8;
      bf16x8 tv[SPAN];
#pragma unroll
      for (int q = 0; q < SPAN; ++q) tv[q] = ld_bf16x8(src + qo[q]);
#pragma unroll
      for (int q = 0; q < SPAN; ++q)
#pragma unroll
        for (int c = 0; c < 8; ++c) row[q][c] += wgt * bf2f(tv[q][c]);
    }
#pragma unroll
  for (int j = 0; j < FX; ++j) {
    int x0, x1;
    float lx0, lx1;
    src_index(xs + j, t.w, p.w, t.sx, x0, x1, lx0, lx1);
    x0 -= xb;
    x1 -= xb;
#pragma unroll
    for (int q = 0; q < SPAN; ++q) {
      const float wq = (q == x0 ? lx0 : 0.f) + (q == x1 ? lx1 : 0.f);
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[j][c] += wq * row[q][c];
    }
  }
}

__global__ __launch_bounds__(256) void fuse_sum_rows_kernel(FuseParams p) {
  const int cpv = p.c >> 3, runs = p.w / FX;
  const bool per_sample = p.stat_out != nullptr;   // grid (stat_blocks, n): a block stays inside one sample
  const long per_n = (long)p.d * p.h * runs * cpv;
  const long total = (per_sample ? 1 : (long)p.n) * per_n, ibase = per_sample ? (long)blockIdx.y * per_n : 0;
  float sy[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long i_ = blockIdx.x * 256L + threadIdx.x; i_ < total; i_ += (long)gridDim.x * 256) {
    const long i = ibase + i_;
    const int ck = (int)(i % cpv);
    long r = i / cpv;
    const int xs = (int)(r % runs) * FX;
    r /= runs;
    const int y = (int)(r % p.h);
    r /= p.h;
    const int z = (int)(r % p.d), n = (int)(r / p.d);
    const long vv0 = (((long)n * p.d + z) * p.h + y) * p.w + xs;
    float acc[FX][8];
#pragma unroll
    for (int j = 0; j < FX; ++j)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[j][c] = p.bias ? p.bias[ck * 8 + c] : 0.f;
    // The same-resolution terms are the HBM traffic of this kernel (the up-sampled sources sit in L2): their loads
    // are issued first and consumed last, so they are in flight during every up-sampled term's gather/blend phases
    // (launched one dependent phase after another the kernel was latency-bound at 1/3 of the HBM rate).
    bf16x8 pre[2][FX];
    int npre = 0;
    for (int k = 0; k < p.nterms && npre < 2; ++k) {
      const FuseTerm t = p.terms[k];
      if (!t.same) continue;
      if (npre == 0) {
#pragma unroll
        for (int j = 0; j < FX; ++j) pre[0][j] = ld_bf16x8(t.t + (vv0 + j) * t.cs + t.co + ck * 8);
      } else {
#pragma unroll
        for (int j = 0; j < FX; ++j) pre[1][j] = ld_bf16x8(t.t + (vv0 + j) * t.cs + t.co + ck * 8);
      }
      ++npre;
    }
    int nsame = 0;
    for (int k = 0; k < p.nterms; ++k) {   // NOT unrolled (code size, descriptor SGPRs)
      const FuseTerm t = p.terms[k];
      if (t.same) {
        if (nsame++ < 2) continue;          // prefetched above
#pragma unroll
        for (int j = 0; j < FX; ++j) {
          bf16x8 tv = ld_bf16x8(t.t + (vv0 + j) * t.cs + t.co + ck * 8);
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[j][c] += bf2f(tv[c]);
        }
      } else {
        // the first t.span source columns of a run can carry weight (launch-uniform per term: 4 / 3 / 2 for the x2 / x4 / x8 branches,
        // computed by the host with this kernel's own index rule): one straight-line body per span
        switch (t.span) {
          case 2: fuse_up_term<2>(p, t, n, z, y, xs, ck, acc); break;
          case 3: fuse_up_term<3>(p, t, n, z, y, xs, ck, acc); break;
          case 4: fuse_up_term<4>(p, t, n, z, y, xs, ck, acc); break;
          default: fuse_up_term<FSPAN>(p, t, n, z, y, xs, ck, acc); break;
        }
      }
    }
    if (npre > 0) {
#pragma unroll
      for (int j = 0; j < FX; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[j][c] += bf2f(pre[0][j][c]);
    }
    if (npre > 1) {
#pragma unroll
      for (int j = 0; j < FX; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[j][c] += bf2f(pre[1][j][c]);
    }
#pragma unroll
    for (int j = 0; j < FX; ++j) {
      bf16x8 o;
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = f2bf(p.relu ? (acc[j][c] > 0.f ? acc[j][c] : 0.f) : acc[j][c]);
      st_bf16x8(p.out + (vv0 + j) * p.o_cs + p.o_co + ck * 8, o);
      if (per_sample) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { const float r = bf2f(o[c]); sy[c] += r; sq[c] += r * r; }
      }
    }
  }
  if (per_sample) fuse_stats_flush(p, blockIdx.y, sy, sq);
}

// can every up-sampled term's FX-output run be served from FSPAN source voxels?
static bool fuse_rows_ok(FuseParams& p) {   // (also fills every up-sampled term's span)
  if (p.w % FX) return false;
  for (int k = 0; k < p.nterms; ++k) {
    FuseTerm& t = p.terms[k];
    t.span = FSPAN;
    if (t.same) continue;
    int span = 2;
    if (t.w == p.w || p.w < 2) return false;  // same width but another depth/height: not a shape of this path
    const float scale = t.sx;
    for (int xs = 0; xs < p.w; xs += FX) {    // exact check with the kernel's own index rule
      const int lo = (int)(scale * (float)xs);
      int hi = (int)(scale * (float)(xs + FX - 1));
      hi += (hi < t.w - 1) ? 1 : 0;
      if (hi - lo >= FSPAN) return false;
      if (hi - lo + 1 > span) span = hi - lo + 1;
    }
    t.span = span;
  }
  return true;
}

// Statistics partials per sample the fused epilogue writes: enough blocks to fill the chip, at most 128 partials to fold later.
extern "C" int rtp_fuse_stats_nsplit(int n, int c, int d, int h, int w) {
  if (c < 8 || c % 8 || 256 % (c / 8) || n < 1) return 0;
  const long items = (long)d * h * ((w % FX) ? w : w / FX) * (c / 8);
  long s = (items + 255) / 256;          // blocks that would each do one pass
  s = (s + 3) / 4;                       // ~4 items per thread
  const long want = (2048 + n - 1) / n;  // ... but >= ~2048 blocks in the launch when the row is that large
  if (s > want) s = want;
  if (s > 128) s = 128;
  return (int)(s < 1 ? 1 : s);
}

extern "C" int rtp_fuse_sum_stats(const RtpTerm* terms, int nterms, const float* bias, const RtpAct* out, int n, int d, int h,
                                  int w, int relu, float* stat_out, int nsplit, void* stream);

extern "C" int rtp_fuse_sum(const RtpTerm* terms, int nterms, const float* bias, const RtpAct* out, int n, int d, int h,
                            int w, int relu, void* stream) {
  return rtp_fuse_sum_stats(terms, nterms, bias, out, n, d, h, w, relu, nullptr, 0, stream);
}

extern "C" int rtp_fuse_sum_stats(const RtpTerm* terms, int nterms, const float* bias, const RtpAct* out, int n, int d, int h,
                                  int w, int relu, float* stat_out, int nsplit, void* stream) {
  if (!terms || nterms < 1 || nterms > RTP_MAX_TERMS || !out) return RTP_ERR_SHAPE;
  if (stat_out && (nsplit < 1 || 256 % (out->c / 8))) return RTP_ERR_SHAPE;
  FuseParams p;
  p.stat_out = stat_out; p.stat_blocks = stat_out ? nsplit : 0;
  p.nterms = nterms; p.c = out->c;
  if (p.c % 8 || (out->cs % 8) || (out->co % 8)) return RTP_ERR_ALIGN;
  for (int k = 0; k < nterms; ++k) {
    if (terms[k].t.c != p.c || (terms[k].t.cs % 8) || (terms[k].t.co % 8)) return RTP_ERR_ALIGN;
    p.terms[k] = FuseTerm{(const bf16_t*)terms[k].t.ptr, terms[k].t.cs, terms[k].t.co, terms[k].d, terms[k].h, terms[k].w,
                          ac_scale(terms[k].d, d), ac_scale(terms[k].h, h), ac_scale(terms[k].w, w),
                          terms[k].d == d && terms[k].h == h && terms[k].w == w, FSPAN};
  }
  p.bias = bias; p.out = (bf16_t*)out->ptr; p.o_cs = out->cs; p.o_co = out->co;
  p.n = n; p.d = d; p.h = h; p.w = w; p.relu = relu;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  const dim3 grid_rows = stat_out ? dim3(nsplit, n) : dim3(grid_for((long)n * d * h * (w / FX) * (p.c / 8)));
  const dim3 grid_pts = stat_out ? dim3(nsplit, n) : dim3(grid_for((long)n * d * h * w * (p.c / 8)));
  if (fuse_rows_ok(p))
    hipLaunchKernelGGL(fuse_sum_rows_kernel, grid_rows, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(fuse_sum_kernel, grid_pts, dim3(256), 0, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_upsample_bwd : the trilinear operator is separable, so its adjoint is three 1-D gathers (x, then y, then z).
// Each pass reads its input once (16 B per lane, neighbours hit L1) instead of the 3-D gather's (2/scale+1)^3-fold
// re-read of the full-resolution gradient; intermediates stay fp32, only the final result is rounded to bf16.
// ------------------------------------------------------------------------------------------------
// weight of high-res position o on low-res index il along one dim
__device__ __forceinline__ float adj_w(int o, int il, int I, int O) {
  int i0, i1; float l0, l1;
  src_index(o, I, O, ac_scale(I, O), i0, i1, l0, l1);
  return (i0 == il ? l0 : 0.f) + (i1 == il ? l1 : 0.f);
}

__device__ __forceinline__ void support(int il, int I, int O, int& lo, int& hi) {
  if (I == O) { lo = hi = il; return; }
  if (I == 1 || O == 1) { lo = 0; hi = O - 1; return; }
  const float inv = (float)(O - 1) / (float)(I - 1);
  lo = (int)floorf((float)(il - 1) * inv) - 1;
  hi = (int)ceilf((float)(il + 1) * inv) + 1;
  if (lo < 0) lo = 0;
  if (hi > O - 1) hi = O - 1;
}

// Tap table of the adjoint along one axis, built once per block in LDS: for every low-resolution index il the high-resolution
// positions o with a NON-ZERO weight (ascending) and their weights.  The kernels below used to recompute support() and adj_w()
// (floorf / ceilf, an int<->float round trip and two compares per candidate tap) for every tap of every output: ~30 vector
// instructions per 16-byte load -- the passes ran at 1.6 TB/s of a streaming read, VALU-bound.  Same taps, same order, same
// weights (a round's padding taps add 0 x value): results equal the on-the-fly form, which is kept as the fall-back for axes that
// do not fit the table.
#define ADJ_MAXI 96
#define ADJ_MAXT 24
struct AdjTab {
  float w[ADJ_MAXI][ADJ_MAXT];
  short o[ADJ_MAXI][ADJ_MAXT];
  short cnt[ADJ_MAXI];
  short ok;   // 0: an index has more than ADJ_MAXT taps -> callers fall back
};

__device__ __forceinline__ void adj_tab_build(AdjTab& t, int I, int O) {   // whole block; __syncthreads() inside
  if (threadIdx.x == 0) t.ok = (I <= ADJ_MAXI) ? 1 : 0;
  __syncthreads();
  if (I <= ADJ_MAXI) {
    for (int il = threadIdx.x; il < I; il += blockDim.x) {
      int lo, hi, k = 0;
      support(il, I, O, lo, hi);
      for (int o = lo; o <= hi; ++o) {
        const float wgt = adj_w(o, il, I, O);
        if (wgt == 0.f) continue;
        if (k < ADJ_MAXT) { t.w[il][k] = wgt; t.o[il][k] = (short)o; }
        ++k;
      }
      t.cnt[il] = (short)(k <= ADJ_MAXT ? k : 0);
      if (k > ADJ_MAXT) t.ok = 0;
    }
  }
  __syncthreads();
}

// out[outer][il][inner][c] = sum_o w(o -> il) * in[outer][o][inner][c]   (channels-last, 8 channels per lane)
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void adjoint_axis_kernel(const TIN* in, int in_cs, int in_co, TOUT* out, int out_cs,
                                                           int out_co, int c, long outer, int I, int O, long inner) {
  __shared__ AdjTab tab;
  adj_tab_build(tab, I, O);
  const bool fast = tab.ok != 0;   // block-uniform
  const int cpv = c >> 3;
  const long total = outer * I * inner * cpv;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ck = (int)(i % cpv);
    long r = i / cpv;
    const long in_i = r % inner; r /= inner;
    const int il = (int)(r % I);
    const long ou = r / I;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (fast) {
      // four taps per round, all four loads issued before the first use (one load in flight per thread left these passes
      // parked on s_waitcnt 75-85 % of the time); a round's missing taps re-read the last one with weight 0
      const int cnt = tab.cnt[il];
      const TIN* base = in + (ou * O * inner + in_i) * in_cs + in_co + ck * 8;
      typedef TIN tin8 __attribute__((ext_vector_type(8)));
      for (int k0 = 0; k0 < cnt; k0 += 4) {
        tin8 v[4];
        float wg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + u < cnt ? k0 + u : cnt - 1;
          wg[u] = k0 + u < cnt ? tab.w[il][k] : 0.f;
          v[u] = *reinterpret_cast<const tin8*>(base + (long)tab.o[il][k] * inner * in_cs);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += wg[u] * (float)v[u][j];
      }
    } else {
      int lo, hi;
      support(il, I, O, lo, hi);
      for (int o = lo; o <= hi; ++o) {
        const float wgt = adj_w(o, il, I, O);
        if (wgt == 0.f) continue;
        const TIN* src = in + ((ou * O + o) * inner + in_i) * in_cs + in_co + ck * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wgt * (float)src[j];
      }
    }
    TOUT* dst = out + ((ou * I + il) * inner + in_i) * out_cs + out_co + ck * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = (TOUT)acc[j];
  }
}

// x and y passes in one kernel: out[n*d][hl][wl][c] = sum_oy wy(oy -> hl) * (sum_ox wx(ox -> wl) * in[n*d][oy][ox][c]) with the
// row sums formed first, in the same order as the two separate passes (bit-identical results) -- but the fp32 row
// intermediate [n][d][h][wl][c] (as many bytes as the bf16 input) is neither written nor read back: 346 -> 178 MB per
// level-1 -> level-0 adjoint.  Neighbouring outputs share their (2/scale + 2)^2 inputs through L1.
__global__ __launch_bounds__(256) void adjoint_xy_kernel(const bf16_t* in, int in_cs, int in_co, float* out, int c, long outer,
                                                         int HL, int H, int WL, int W) {
  __shared__ AdjTab taby, tabx;
  adj_tab_build(taby, HL, H);
  adj_tab_build(tabx, WL, W);
  const bool fast = taby.ok != 0 && tabx.ok != 0;   // block-uniform
  const int cpv = c >> 3;
  const long total = outer * HL * WL * cpv;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ck = (int)(i % cpv);
    long r = i / cpv;
    const int xl = (int)(r % WL); r /= WL;
    const int yl = (int)(r % HL);
    const long ou = r / HL;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (fast && tabx.cnt[xl] <= 5) {
      // the x taps of a row (at most 5 for the x2 adjoints) are loaded together, and the NEXT row's loads are issued before this
      // row's arithmetic: up to ten 16-byte loads in flight per thread instead of one
      const int ny = taby.cnt[yl], nx = tabx.cnt[xl];
      long xo[5];
      float wx[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int k = u < nx ? u : nx - 1;
        xo[u] = (long)tabx.o[xl][k] * in_cs;
        wx[u] = u < nx ? tabx.w[xl][k] : 0.f;
      }
      const bf16_t* plane = in + (ou * H * W) * in_cs + in_co + ck * 8;
      bf16x8 cur[5], nxt[5];
      {
        const bf16_t* srow = plane + (long)taby.o[yl][0] * W * in_cs;
#pragma unroll
        for (int u = 0; u < 5; ++u) cur[u] = ld_bf16x8(srow + xo[u]);
      }
      for (int ky = 0; ky < ny; ++ky) {
        const float wy = taby.w[yl][ky];
        const int kn = ky + 1 < ny ? ky + 1 : ky;
        const bf16_t* srow = plane + (long)taby.o[yl][kn] * W * in_cs;
#pragma unroll
        for (int u = 0; u < 5; ++u) nxt[u] = ld_bf16x8(srow + xo[u]);
        float row[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j] = 0.f;
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) row[j] += wx[u] * bf2f(cur[u][j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wy * row[j];
#pragma unroll
        for (int u = 0; u < 5; ++u) cur[u] = nxt[u];
      }
    } else if (fast) {
      const int ny = taby.cnt[yl], nx = tabx.cnt[xl];
      for (int ky = 0; ky < ny; ++ky) {
        const float wy = taby.w[yl][ky];
        float row[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j] = 0.f;
        const bf16_t* srow = in + ((ou * H + taby.o[yl][ky]) * W) * in_cs + in_co + ck * 8;
        for (int kx = 0; kx < nx; ++kx) {
          const float wx = tabx.w[xl][kx];
          const bf16_t* src = srow + (long)tabx.o[xl][kx] * in_cs;
#pragma unroll
          for (int j = 0; j < 8; ++j) row[j] += wx * (float)src[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wy * row[j];
      }
    } else {
      int ylo, yhi, xlo, xhi;
      support(yl, HL, H, ylo, yhi);
      support(xl, WL, W, xlo, xhi);
      for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = adj_w(oy, yl, HL, H);
        if (wy == 0.f) continue;
        float row[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j] = 0.f;
        const bf16_t* srow = in + ((ou * H + oy) * W) * in_cs + in_co + ck * 8;
        for (int ox = xlo; ox <= xhi; ++ox) {
          const float wx = adj_w(ox, xl, WL, W);
          if (wx == 0.f) continue;
          const bf16_t* src = srow + (long)ox * in_cs;
#pragma unroll
          for (int j = 0; j < 8; ++j) row[j] += wx * (float)src[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wy * row[j];
      }
    }
    float* dst = out + ((ou * HL + yl) * WL + xl) * c + ck * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = acc[j];
  }
}

extern "C" long rtp_upsample_bwd_scratch_floats(int n, int c, int d, int h, int w, int dl, int hl, int wl) {
  (void)w; (void)dl;
  return (long)n * c * ((long)d * h * wl + (long)d * hl * wl);
}

extern "C" int rtp_upsample_bwd(const RtpAct* ghi, int d, int h, int w, const RtpAct* glow, int dl, int hl, int wl,
                                int n, float* scratch, void* stream) {
  if (!ghi || !glow || !scratch) return RTP_ERR_SHAPE;
  if (ghi->c != glow->c || ghi->c % 8 || (ghi->cs % 8) || (ghi->co % 8) || (glow->cs % 8) || (glow->co % 8)) return RTP_ERR_ALIGN;
  const int c = ghi->c;
  float* t1 = scratch;                               // [n][d][h][wl][c]
  float* t2 = scratch + (long)n * d * h * wl * c;    // [n][d][hl][wl][c]
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  if (h <= 3 * hl && w <= 3 * wl) {
    // x and y fused (small supports: the x2 adjoints from the full-resolution level): outer = n*d slices
    hipLaunchKernelGGL(adjoint_xy_kernel, dim3(grid_for((long)n * d * hl * wl * (c / 8))), dim3(256), 0, s,
                       (const bf16_t*)ghi->ptr, ghi->cs, ghi->co, t2, c, (long)n * d, hl, h, wl, w);
  } else {
    // x: outer = n*d*h rows, inner = 1
    hipLaunchKernelGGL((adjoint_axis_kernel<bf16_t, float>), dim3(grid_for((long)n * d * h * wl * (c / 8))), dim3(256), 0, s,
                       (const bf16_t*)ghi->ptr, ghi->cs, ghi->co, t1, c, 0, c, (long)n * d * h, wl, w, 1L);
    // y: outer = n*d, inner = wl
    hipLaunchKernelGGL((adjoint_axis_kernel<float, float>), dim3(grid_for((long)n * d * hl * wl * (c / 8))), dim3(256), 0, s,
                       (const float*)t1, c, 0, t2, c, 0, c, (long)n * d, hl, h, (long)wl);
  }
  // z: outer = n, inner = hl*wl
  hipLaunchKernelGGL((adjoint_axis_kernel<float, bf16_t>), dim3(grid_for((long)n * dl * hl * wl * (c / 8))), dim3(256), 0, s,
                     (const float*)t2, c, 0, (bf16_t*)glow->ptr, glow->cs, glow->co, c, (long)n, dl, d, (long)hl * wl);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// Cin == 1 stem (layer1.conv1, hr_util/common.py:111-113)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* x, const float* w, const float* b, bf16_t* y,
                                                       int y_cs, int y_co, int c, long total_vox) {
  const int cpv = c >> 3;
  const long total = total_vox * cpv;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ck = (int)(i % cpv);
    const long vv = i / cpv;
    const float xv = x[vv];
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(xv * w[ck * 8 + j] + b[ck * 8 + j]);
    st_bf16x8(y + vv * y_cs + y_co + ck * 8, o);
  }
}

// ... with the per-channel statistics (sum y, sum y^2 of the STORED values) the first GroupNorm wants, one partial per block:
// grid (nsplit, n), a block stays inside one sample -- the read pass of rtp_chan_stats over the 84-MB stem output is gone.
__global__ __launch_bounds__(256) void stem_fwd_stats_kernel(const float* x, const float* w, const float* b, bf16_t* y, int y_cs, int y_co,
                                                             int c, long vox, float* stat_out, int stat_blocks) {
  const int cpv = c >> 3, n = blockIdx.y;
  const long total = vox * cpv;
  float sy[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int ck = (int)(threadIdx.x % cpv);      // the same chunk for every item of a thread (the stride is a multiple of 256, cpv | 256)
  float wr[8], br[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { wr[j] = w[ck * 8 + j]; br[j] = b[ck * 8 + j]; }
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long vv = (long)n * vox + i / cpv;
    const float xv = x[vv];
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o[j] = f2bf(xv * wr[j] + br[j]);
      const float r = bf2f(o[j]);
      sy[j] += r; sq[j] += r * r;
    }
    st_bf16x8(y + vv * y_cs + y_co + ck * 8, o);
  }
  stats_flush(stat_out, stat_blocks, c, n, sy, sq);
}

extern "C" int rtp_stem_stats_nsplit(int n, int c, long vox) {
  if (c < 8 || c % 8 || 256 % (c / 8) || n < 1) return 0;
  long s = (vox * (c / 8) + 1023) / 1024;   // ~4 items per thread
  const long want = (2048 + n - 1) / n;
  if (s > want) s = want;
  if (s > 128) s = 128;
  return (int)(s < 1 ? 1 : s);
}

extern "C" int rtp_stem_fwd_stats(const float* x, const float* w, const float* b, const RtpAct* y, int n, long vox, float* stat_out,
                                  int nsplit, void* stream) {
  if (!x || !w || !b || !y || !stat_out || nsplit < 1 || y->c % 8 || (y->cs % 8) || (y->co % 8) || 256 % (y->c / 8)) return RTP_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  hipLaunchKernelGGL(stem_fwd_stats_kernel, dim3(nsplit, n), dim3(256), 0, s, x, w, b, (bf16_t*)y->ptr, y->cs, y->co, y->c, vox, stat_out,
                     nsplit);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

extern "C" int rtp_stem_fwd(const float* x, const float* w, const float* b, const RtpAct* y, int n, long vox,
                            void* stream) {
  if (!x || !w || !b || !y || y->c % 8 || (y->cs % 8) || (y->co % 8)) return RTP_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  hipLaunchKernelGGL(stem_fwd_kernel, dim3(grid_for((long)n * vox * (y->c / 8))), dim3(256), 0, s, x, w, b,
                     (bf16_t*)y->ptr, y->cs, y->co, y->c, (long)n * vox);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

#define STEM_BWD_BLOCKS 256
extern "C" int rtp_stem_bwd_blocks(void) { return STEM_BWD_BLOCKS; }

__global__ __launch_bounds__(256) void stem_bwd_partial(const float* x, const bf16_t* g, int g_cs, int g_co, int c,
                                                        long total_vox, float* scratch) {
  __shared__ float red[256 * 17];
  const int cpv = c >> 3, tid = threadIdx.x;
  const int ck = tid % cpv, vsub = tid / cpv, vper = 256 / cpv;
  float sw[8], sb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) sw[j] = sb[j] = 0.f;
  if (vsub < vper) {
    const long step = (long)gridDim.x * vper;
    for (long v0 = (long)blockIdx.x * vper + vsub; v0 < total_vox; v0 += 4 * step) {   // four voxels in flight per lane
      float xv[4];
      bf16x8 gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long v = v0 + u * step;
        xv[u] = 0.f;
        gv[u] = zero_bf16x8();
        if (v < total_vox) { xv[u] = x[v]; gv[u] = ld_bf16x8(g + v * g_cs + g_co + ck * 8); }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float gf = bf2f(gv[u][j]); sw[j] += xv[u] * gf; sb[j] += gf; }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[tid * 17 + j] = sw[j]; red[tid * 17 + 8 + j] = sb[j]; }
  __syncthreads();
  for (int t = tid; t < 2 * c; t += 256) {
    const int ch = t >> 1, which = t & 1;
    float acc = 0.f;
    for (int vs = 0; vs < vper; ++vs) acc += red[(vs * cpv + (ch >> 3)) * 17 + which * 8 + (ch & 7)];
    scratch[((long)blockIdx.x * c + ch) * 2 + which] = acc;
  }
}

// 256 threads: output (channel, which) = t % 2c, the block partials dealt over 256/(2c) threads and folded in fixed order
__global__ __launch_bounds__(256) void stem_bwd_final(const float* scratch, int nblk, int c, float* dw, float* db, int accumulate) {
  __shared__ float red[256];
  const int t = threadIdx.x, no = 2 * c, np = 256 / no;
  const int o_ = t % no, pt = t / no;
  float acc = 0.f;
  if (pt < np)
    for (int b = pt; b < nblk; b += np) acc += scratch[(long)b * no + o_];   // [blk][c][2] flattened
  red[t] = acc;
  __syncthreads();
  if (t >= no) return;
  float a = 0.f;
  for (int k = 0; k < np; ++k) a += red[k * no + t];
  const int ch = t >> 1, which = t & 1;
  float* o = which ? db + ch : dw + ch;
  if (accumulate) *o += a; else *o = a;
}

extern "C" int rtp_stem_bwd(const float* x, const RtpAct* gy, int n, long vox, float* scratch, float* dw, float* db,
                            int accumulate, void* stream) {
  if (!x || !gy || !scratch || gy->c % 8 || gy->c > 128 || (256 % (gy->c / 8))) return RTP_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  hipLaunchKernelGGL(stem_bwd_partial, dim3(STEM_BWD_BLOCKS), dim3(256), 0, s, x, (const bf16_t*)gy->ptr, gy->cs,
                     gy->co, gy->c, (long)n * vox, scratch);
  hipLaunchKernelGGL(stem_bwd_final, dim3(1), dim3(256), 0, s, scratch, STEM_BWD_BLOCKS, gy->c, dw, db, accumulate);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// NCDHW fp32 <-> channels-last bf16
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kernel(const float* x, bf16_t* y, int y_cs, int y_co, int n, int c,
                                                   int cpad, long vox) {
  const int cpv = cpad >> 3;
  const long total = (long)n * vox * cpv;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long v = i % vox;  // voxel fastest across threads -> coalesced per channel plane
    const int ck = (int)((i / vox) % cpv);
    const int nn = (int)(i / (vox * cpv));
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = ck * 8 + j;
      o[j] = f2bf(ch < c ? x[((long)nn * c + ch) * vox + v] : 0.f);
    }
    st_bf16x8(y + ((long)nn * vox + v) * y_cs + y_co + ck * 8, o);
  }
}

// y = bf16([relu](x (+ x2))): the hand-off from the fp32 NCHW deformable-convolution operator back to the plan's layout
__global__ __launch_bounds__(256) void pack_ex_kernel(const float* x, const float* x2, bf16_t* y, int y_cs, int y_co, int n, int c,
                                                      int cpad, long vox, int relu) {
  const int cpv = cpad >> 3;
  const long total = (long)n * vox * cpv;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long v = i % vox;
    const int ck = (int)((i / vox) % cpv);
    const int nn = (int)(i / (vox * cpv));
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = ck * 8 + j;
      float a = 0.f;
      if (ch < c) {
        a = x[((long)nn * c + ch) * vox + v];
        if (x2) a += x2[((long)nn * c + ch) * vox + v];
        if (relu) a = a > 0.f ? a : 0.f;
      }
      o[j] = f2bf(a);
    }
    st_bf16x8(y + ((long)nn * vox + v) * y_cs + y_co + ck * 8, o);
  }
}

// fp32 channels-last -> NC(D)HW through an LDS tile of 64 voxels x all channels: 16-byte reads along the channels, 256-byte
// rows along the voxels on the way out (the per-element version read with a 320-byte stride: 0.9 TB/s on the DCN head's
// 72-channel offsets).  x_cs % 4 == 0, x_co % 4 == 0.
__global__ __launch_bounds__(256) void unpack_f32_tile_kernel(const float* x, int x_cs, int x_co, float* y, int c, long vox) {
  extern __shared__ float ut_lds[];   // [c4 * 4][65]
  const int tid = threadIdx.x, c4 = (c + 3) >> 2;
  const long v0 = (long)blockIdx.x * 64;
  const int nn = blockIdx.y;
  const int nv = (vox - v0 < 64) ? (int)(vox - v0) : 64;
  for (int i = tid; i < nv * c4; i += 256) {
    const int k = i % c4, v = i / c4;
    const f32x4 t = *reinterpret_cast<const f32x4*>(x + ((long)nn * vox + v0 + v) * x_cs + x_co + k * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) ut_lds[(k * 4 + j) * 65 + v] = t[j];
  }
  __syncthreads();
  for (int i = tid; i < c * 64; i += 256) {
    const int v = i & 63, ch = i >> 6;
    if (v < nv) y[((long)nn * c + ch) * vox + v0 + v] = ut_lds[ch * 65 + v];
  }
}

__global__ __launch_bounds__(256) void unpack_f32_kernel(const float* x, int x_cs, int x_co, float* y, int n, int c, long vox) {
  const long total = (long)n * c * vox;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long v = i % vox;
    const int ch = (int)((i / vox) % c);
    const int nn = (int)(i / (vox * c));
    y[i] = x[((long)nn * vox + v) * x_cs + x_co + ch];
  }
}

// fp32 channels-last [n][vox][cs] (channels [co, co + c)) -> fp32 NC(D)HW [n][c][vox]: hands an fp32 conv output of the plan
// (the offsets of the DCN head) to the NCHW deformable-convolution operator
extern "C" int rtp_unpack_ncdhw_f32(const float* x, int x_cs, int x_co, float* y, int n, int c, long vox, void* stream) {
  if (!x || !y || n < 1 || c < 1 || x_co + c > x_cs) return RTP_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  if (x_cs % 4 == 0 && x_co % 4 == 0 && x_co + ((c + 3) / 4) * 4 <= x_cs && c <= 256 && (vox + 63) / 64 < (1L << 31) && n < 65536) {
    const size_t lds = sizeof(float) * ((c + 3) / 4) * 4 * 65;
    hipLaunchKernelGGL(unpack_f32_tile_kernel, dim3((unsigned)((vox + 63) / 64), n), dim3(256), lds, s, x, x_cs, x_co, y, c, vox);
    RTP_CHECK_LAUNCH();
    return RTP_OK;
  }
  hipLaunchKernelGGL(unpack_f32_kernel, dim3(grid_for((long)n * vox * c)), dim3(256), 0, s, x, x_cs, x_co, y, n, c, vox);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// The two hand-offs around the NCHW operator through an LDS tile of 64 voxels x all channels (256-byte rows on the NCHW side,
// 16-byte chunks on the channels-last side) instead of one strided element / chunk per thread.
__global__ __launch_bounds__(256) void pack_ex_tile_kernel(const float* x, const float* x2, bf16_t* y, int y_cs, int y_co, int c,
                                                           int cpad, long vox, int relu) {
  extern __shared__ float pt_lds[];   // [cpad][65]
  const int tid = threadIdx.x, nn = blockIdx.y;
  const long v0 = (long)blockIdx.x * 64;
  const int nv = (vox - v0 < 64) ? (int)(vox - v0) : 64;
  for (int i = tid; i < cpad * 64; i += 256) {
    const int v = i & 63, ch = i >> 6;
    float a = 0.f;
    if (ch < c && v < nv) {
      a = x[((long)nn * c + ch) * vox + v0 + v];
      if (x2) a += x2[((long)nn * c + ch) * vox + v0 + v];
      if (relu) a = a > 0.f ? a : 0.f;
    }
    pt_lds[ch * 65 + v] = a;
  }
  __syncthreads();
  const int cpv = cpad >> 3;
  for (int i = tid; i < nv * cpv; i += 256) {
    const int ck = i % cpv, v = i / cpv;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(pt_lds[(ck * 8 + j) * 65 + v]);
    st_bf16x8(y + ((long)nn * vox + v0 + v) * y_cs + y_co + ck * 8, o);
  }
}

__global__ __launch_bounds__(256) void unpack_tile_kernel(const bf16_t* x, int x_cs, int x_co, float* y, int c, long vox) {
  extern __shared__ float pt_lds[];   // [c8 * 8][65]
  const int tid = threadIdx.x, nn = blockIdx.y, c8 = (c + 7) >> 3;
  const long v0 = (long)blockIdx.x * 64;
  const int nv = (vox - v0 < 64) ? (int)(vox - v0) : 64;
  for (int i = tid; i < nv * c8; i += 256) {
    const int k = i % c8, v = i / c8;
    const bf16x8 t = ld_bf16x8(x + ((long)nn * vox + v0 + v) * x_cs + x_co + k * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) pt_lds[(k * 8 + j) * 65 + v] = bf2f(t[j]);
  }
  __syncthreads();
  for (int i = tid; i < c * 64; i += 256) {
    const int v = i & 63, ch = i >> 6;
    if (v < nv) y[((long)nn * c + ch) * vox + v0 + v] = pt_lds[ch * 65 + v];
  }
}

extern "C" int rtp_pack_ncdhw_ex(const float* x, const float* x2, const RtpAct* y, int n, int c, long vox, int relu, void* stream) {
  if (!x || !y || y->c % 8 || c > y->c || (y->cs % 8) || (y->co % 8)) return RTP_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  if (y->c <= 128 && n < 65536) {
    hipLaunchKernelGGL(pack_ex_tile_kernel, dim3((unsigned)((vox + 63) / 64), n), dim3(256), sizeof(float) * y->c * 65, s, x, x2,
                       (bf16_t*)y->ptr, y->cs, y->co, c, y->c, vox, relu);
    RTP_CHECK_LAUNCH();
    return RTP_OK;
  }
  hipLaunchKernelGGL(pack_ex_kernel, dim3(grid_for((long)n * vox * (y->c / 8))), dim3(256), 0, s, x, x2, (bf16_t*)y->ptr,
                     y->cs, y->co, n, c, y->c, vox, relu);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

extern "C" int rtp_pack_ncdhw(const float* x, const RtpAct* y, int n, int c, long vox, void* stream) {
  if (!x || !y || y->c % 8 || c > y->c || (y->cs % 8) || (y->co % 8)) return RTP_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  hipLaunchKernelGGL(pack_kernel, dim3(grid_for((long)n * vox * (y->c / 8))), dim3(256), 0, s, x, (bf16_t*)y->ptr,
                     y->cs, y->co, n, c, y->c, vox);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

__global__ __launch_bounds__(256) void unpack_kernel(const bf16_t* x, int x_cs, int x_co, float* y, int n, int c,
                                                     long vox) {
  const long total = (long)n * c * vox;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long v = i % vox;
    const int ch = (int)((i / vox) % c);
    const int nn = (int)(i / (vox * c));
    y[i] = bf2f(x[((long)nn * vox + v) * x_cs + x_co + ch]);
  }
}

extern "C" int rtp_unpack_ncdhw(const RtpAct* x, float* y, int n, int c, long vox, void* stream) {
  if (!x || !y || c > x->c) return RTP_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  if ((x->cs % 8) == 0 && (x->co % 8) == 0 && x->co + ((c + 7) / 8) * 8 <= x->cs && c <= 128 && n < 65536) {
    hipLaunchKernelGGL(unpack_tile_kernel, dim3((unsigned)((vox + 63) / 64), n), dim3(256), sizeof(float) * ((c + 7) / 8) * 8 * 65, s,
                       (const bf16_t*)x->ptr, x->cs, x->co, y, c, vox);
    RTP_CHECK_LAUNCH();
    return RTP_OK;
  }
  hipLaunchKernelGGL(unpack_kernel, dim3(grid_for((long)n * vox * c)), dim3(256), 0, s, (const bf16_t*)x->ptr, x->cs,
                     x->co, y, n, c, vox);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
