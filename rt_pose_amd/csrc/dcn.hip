// Deformable convolution (DCNv1 + modulated DCNv2) for gfx950 -- C-ABI replacements of the five entry points of the
// reference's pybind module deform_conv_cuda (det3d/ops/dcn/src/deform_conv_cuda.cpp:152-157, 262-268, 376-381,
// 490-496, 571-578).  Contiguous NCHW fp32 tensors as in the reference; offsets [N, dg*2*kh*kw, Ho, Wo] ordered
// (dh, dw) per tap; mask [N, dg*kh*kw, Ho, Wo].
//
// Structure (what each reference kernel became):
//   deformable_im2col (+modulated)      -> dcn_im2col_kernel: one thread per (channel, image, ho, wo), wo fastest so the
//                                          column matrix is written fully coalesced; the 4-corner gather is the
//                                          irregular HBM/L2 access of the operator
//   addmm_ per group (cuBLAS)            -> sgemm_kernel: LDS-tiled 64x64x16 fp32 GEMM with strided "matrix views", so
//                                          the [Co][step*P] -> [N][Co][P] output transpose of the reference
//                                          (cpp:243-246) is folded into the store
//   deformable_col2im (atomicAdd)        -> dcn_col2im_kernel (hardware fp32 global atomics, 4 live corners instead of
//                                          the reference's 5x5 scan)
//   deformable_col2im_coord              -> dcn_col2im_coord_kernel (also emits the mask gradient for v2)
// The `columns`/`ones` scratch tensors of the reference become one caller-provided workspace.
#include "rtp_common.h"
#include "rtp_prof.h"

struct DcnGeom {
  int n, c, h, w, co, kh, kw, sh, sw, ph, pw, dh, dw, group, dg, ho, wo;
};

// Two layouts of the caller's workspace:
//   columns          [c*K + t][bl*P + p]           written by im2col, read by the GEMMs of forward / weight gradient
//   column GRADIENT  [bl][dg][t][p][c % cpg]       written by the column-gradient GEMM, read by the coordinate gradient and
//                                                  col2im: a position's channels of one deformable group are contiguous
//                                                  (one 16/32-byte load instead of cpg loads 23 MB apart)
__host__ __device__ __forceinline__ long dcn_cg_index(const DcnGeom& g, int c, int t, int bl, int p) {
  const int cpg = g.c / g.dg, K = g.kh * g.kw, P = g.ho * g.wo;
  return ((((long)bl * g.dg + c / cpg) * K + t) * P + p) * cpg + c % cpg;
}

static inline void dcn_out_size(DcnGeom& g) {
  g.ho = (g.h + 2 * g.ph - (g.dh * (g.kh - 1) + 1)) / g.sh + 1;
  g.wo = (g.w + 2 * g.pw - (g.dw * (g.kw - 1) + 1)) / g.sw + 1;
}

__device__ __forceinline__ float dcn_bilinear(const float* im, int H, int W, float h, float w) {
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h - h_low, lw = w - w_low, hh = 1.f - lh, hw = 1.f - lw;
  float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
  if (h_low >= 0 && w_low >= 0) v1 = im[h_low * W + w_low];
  if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * W + w_high];
  if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * W + w_low];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * W + w_high];
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

// columns[(c*K + tap)][bl*P + p]   for images b0 .. b0+step-1
__global__ __launch_bounds__(256) void dcn_im2col_kernel(const float* x, const float* offset, const float* mask,
                                                         float* col, DcnGeom g, int b0, int step) {
  const int P = g.ho * g.wo, K = g.kh * g.kw;
  const long total = (long)g.c * step * P;
  const int cpg = g.c / g.dg;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int p = (int)(idx % P);
    const int bl = (int)((idx / P) % step);
    const int c = (int)(idx / ((long)P * step));
    const int wo = p % g.wo, ho = p / g.wo;
    const int b = b0 + bl, dgi = c / cpg;
    const float* im = x + ((long)b * g.c + c) * g.h * g.w;
    const float* off = offset + ((long)b * g.dg + dgi) * 2 * K * P;
    const float* mk = mask ? mask + ((long)b * g.dg + dgi) * K * P : nullptr;
    const int h_in = ho * g.sh - g.ph, w_in = wo * g.sw - g.pw;
    float* dst = col + ((long)c * K * step + bl) * P + p;
    for (int t = 0; t < K; ++t) {
      const int i = t / g.kw, j = t - i * g.kw;
      const float hi = h_in + i * g.dh + off[(2 * t) * P + p];
      const float wi = w_in + j * g.dw + off[(2 * t + 1) * P + p];
      float v = 0.f;
      if (hi > -1.f && wi > -1.f && hi < g.h && wi < g.w) v = dcn_bilinear(im, g.h, g.w, hi, wi);
      if (mk) v *= mk[t * P + p];
      dst[(long)t * step * P] = v;
    }
  }
}

// im2col with the sample set-up shared by a deformable group: one thread per (image, deformable group, position) computes
// each tap's coordinates and bilinear weights once and walks the group's channels (the kernel above recomputes them per
// channel); corners come as 8-byte column pairs through a buffer resource over the image (out-of-image rows = offset
// past num_records = zeros; see dcn_fused_fwd_kernel).  Column writes stay coalesced along p.
typedef unsigned dcn_u32x2 __attribute__((ext_vector_type(2)));
#define DCN_OOB ((int)0x80000000)
__global__ __launch_bounds__(256) void dcn_im2col_group_kernel(const float* x, const float* offset, const float* mask,
                                                               float* col, DcnGeom g, int b0, int step) {
  const int P = g.ho * g.wo, K = g.kh * g.kw, HW = g.h * g.w;
  const long total = (long)step * g.dg * P;
  const int cpg = g.c / g.dg;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int p = (int)(idx % P);
    const int dgi = (int)((idx / P) % g.dg);
    const int bl = (int)(idx / ((long)P * g.dg)), b = b0 + bl;
    const int wo = p % g.wo, ho = p / g.wo;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)b * g.c * HW), 0, g.c * HW * 4, 0x00020000);
    const float* off = offset + ((long)b * g.dg + dgi) * 2 * K * P + p;
    const float* mk = mask ? mask + ((long)b * g.dg + dgi) * K * P + p : nullptr;
    const int h_in = ho * g.sh - g.ph, w_in = wo * g.sw - g.pw;
    for (int t = 0; t < K; ++t) {
      const int i = t / g.kw, j = t - i * g.kw;
      const float hi = h_in + i * g.dh + off[(long)(2 * t) * P];
      const float wi = w_in + j * g.dw + off[(long)(2 * t + 1) * P];
      const float m = mk ? mk[(long)t * P] : 1.f;
      const bool in = hi > -1.f && wi > -1.f && hi < g.h && wi < g.w;
      const float hf = floorf(hi), wf = floorf(wi);
      const int h_low = (int)hf, w_low = (int)wf;
      const float lh = hi - hf, lw = wi - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool c0 = w_low >= 0, c1 = w_low + 1 <= g.w - 1;
      const float w1 = c0 ? hh * hw : 0.f, w2 = c1 ? hh * lw : 0.f, w3 = c0 ? lh * hw : 0.f, w4 = c1 ? lh * lw : 0.f;
      const int xs = w_low < 0 ? 0 : (c1 ? w_low : g.w - 2);
      const int base = (h_low * g.w + xs) * 4;
      const int a0 = (in && h_low >= 0) ? base : DCN_OOB;
      const int a1 = (in && h_low + 1 <= g.h - 1) ? base + g.w * 4 : DCN_OOB;
      float* dst = col + (((long)dgi * cpg * K + t) * step + bl) * P + p;
      for (int cc = 0; cc < cpg; ++cc) {
        const int coff = (dgi * cpg + cc) * HW * 4;
        const dcn_u32x2 p0 = __builtin_amdgcn_raw_buffer_load_b64(rx, a0 + coff, 0, 0);
        const dcn_u32x2 p1 = __builtin_amdgcn_raw_buffer_load_b64(rx, a1 + coff, 0, 0);
        const float v1 = __uint_as_float(c1 ? p0.x : p0.y), v2 = __uint_as_float(c0 ? p0.y : p0.x);
        const float v3 = __uint_as_float(c1 ? p1.x : p1.y), v4 = __uint_as_float(c0 ? p1.y : p1.x);
        float v = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
        if (mk) v *= m;
        dst[(long)cc * K * step * P] = v;
      }
    }
  }
}

// grad_im[b][c] += bilinear-adjoint of columns-gradient (atomic scatter to the 4 live corners)
// col2im with the image plane in LDS: one block owns one (sample, channel) plane of grad_im (H*W fp32 <= 60 KB), adds
// all of that plane's 4-corner contributions (every tap x every output position) with LDS atomics and then adds the
// plane to global memory with plain stores -- no global atomics (the scatter kernel below issues 4 per column element:
// 13.6 ms per 64-sample chunk at [32,64,160]) and no other block touches the plane.
__global__ __launch_bounds__(256) void dcn_col2im_plane_kernel(const float* col, const float* offset, const float* mask,
                                                               float* grad_im, DcnGeom g, int b0, int step) {
  extern __shared__ float plane[];
  const int P = g.ho * g.wo, K = g.kh * g.kw, HW = g.h * g.w;
  const int c = blockIdx.x % g.c, bl = blockIdx.x / g.c;
  const int b = b0 + bl, cpg = g.c / g.dg, dgi = c / cpg;
  for (int i = threadIdx.x; i < HW; i += 256) plane[i] = 0.f;
  __syncthreads();
  const float* off = offset + ((long)b * g.dg + dgi) * 2 * K * P;
  for (int t = 0; t < K; ++t) {
    const int ki = t / g.kw, kj = t - ki * g.kw;
    const float* colp = col + dcn_cg_index(g, c, t, bl, 0);
    const int cstr = g.c / g.dg;  // channel-innermost column gradient: positions are cpg floats apart
    const float* mk = mask ? mask + ((long)b * g.dg + dgi) * K * P + (long)t * P : nullptr;
    // four positions per thread and round: their 12 loads are in flight together before the first LDS atomic
    for (int pb = threadIdx.x; pb < P; pb += 4 * 256) {
      float oh[4], ow[4], top[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = pb + u * 256;
        const bool ok = p < P;
        oh[u] = ok ? off[(2 * t) * P + p] : 0.f;
        ow[u] = ok ? off[(2 * t + 1) * P + p] : 0.f;
        top[u] = ok ? colp[(long)p * cstr] * (mk ? mk[p] : 1.f) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = pb + u * 256;
        if (p >= P) break;
        const int wo = p % g.wo, ho = p / g.wo;
        const float hi = ho * g.sh - g.ph + ki * g.dh + oh[u];
        const float wi = wo * g.sw - g.pw + kj * g.dw + ow[u];
        if (!(hi > -1.f && wi > -1.f && hi < g.h && wi < g.w)) continue;
        const int h_low = (int)floorf(hi), w_low = (int)floorf(wi);
        const float lh = hi - h_low, lw = wi - w_low;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const int yy = h_low + dy, xx = w_low + dx;
            if (yy >= 0 && yy <= g.h - 1 && xx >= 0 && xx <= g.w - 1) {
              const float wgt = (dy ? lh : 1.f - lh) * (dx ? lw : 1.f - lw);
              if (wgt != 0.f) atomicAdd(&plane[yy * g.w + xx], wgt * top[u]);
            }
          }
      }
    }
  }
  __syncthreads();
  float* gim = grad_im + ((long)b * g.c + c) * HW;
  for (int i = threadIdx.x; i < HW; i += 256) gim[i] += plane[i];
}

__global__ __launch_bounds__(256) void dcn_col2im_kernel(const float* col, const float* offset, const float* mask,
                                                         float* grad_im, DcnGeom g, int b0, int step) {
  const int P = g.ho * g.wo, K = g.kh * g.kw;
  const long total = (long)g.c * K * step * P;
  const int cpg = g.c / g.dg;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int p = (int)(idx % P);
    const int bl = (int)((idx / P) % step);
    const int t = (int)((idx / ((long)P * step)) % K);
    const int c = (int)(idx / ((long)P * step * K));
    const int wo = p % g.wo, ho = p / g.wo;
    const int b = b0 + bl, dgi = c / cpg;
    const int i = t / g.kw, j = t - i * g.kw;
    const float* off = offset + ((long)b * g.dg + dgi) * 2 * K * P;
    const float hi = ho * g.sh - g.ph + i * g.dh + off[(2 * t) * P + p];
    const float wi = wo * g.sw - g.pw + j * g.dw + off[(2 * t + 1) * P + p];
    if (!(hi > -1.f && wi > -1.f && hi < g.h && wi < g.w)) continue;
    float top = col[dcn_cg_index(g, c, t, bl, p)];
    if (mask) top *= mask[((long)b * g.dg + dgi) * K * P + t * P + p];
    const int h_low = (int)floorf(hi), w_low = (int)floorf(wi);
    const float lh = hi - h_low, lw = wi - w_low;
    float* gim = grad_im + ((long)b * g.c + c) * g.h * g.w;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int yy = h_low + dy, xx = w_low + dx;
        if (yy >= 0 && yy <= g.h - 1 && xx >= 0 && xx <= g.w - 1) {
          const float wgt = (dy ? lh : 1.f - lh) * (dx ? lw : 1.f - lw);
          if (wgt != 0.f) atomicAdd(gim + yy * g.w + xx, wgt * top);
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------ col2im as a gather
// grad_im without atomics for samples whose offsets are at most R pixels (the usual case: learned offsets are small).
// A thread owns one pixel (y, x) of CH channels of one deformable group.  A sample (output position p, tap t) with
// |offset| <= R can only touch the pixel if its undeformed tap position lies within R of it, so the thread walks those
// (2R+1)^2 candidate positions per tap, recomputes the sample's bilinear weight for ITS pixel (the reference's weight
// for whichever of the four corners the pixel is, deform_conv_cuda_kernel.cu:117-143) once for all CH channels and
// accumulates col * weight in registers: coalesced loads (lanes = consecutive x -> consecutive wo), a fixed summation
// order (bit-reproducible, unlike the atomic scatter), one plain read-modify-write of grad_im at the end.
// Samples with a larger offset ("outliers") are skipped here and scattered by dcn_col2im_outlier_kernel with global
// fp32 atomics, so any offset field gives the reference's result; only the speed depends on offsets being small.
__device__ __forceinline__ int dcn_floor_div(int a, int b) {  // b > 0
  const int q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}

// A block owns an 8 x 32 pixel tile of CH channels of one deformable group.  Per tap it stages the candidate output
// positions of the whole tile -- (8 + 2R) x (32 + 2R) at stride 1 -- in LDS: the two offsets and CH column values
// (pre-multiplied by the mask) per position, positions outside the output marked with a huge offset.  Every column
// element then crosses the memory system ~1.7x instead of once per candidate pixel (the un-tiled version of this kernel
// moved 17 GB per launch through L2 and ran at its bandwidth).
#define DCN_GT_Y 8
#define DCN_GT_X 32
template <int CH>
__global__ __launch_bounds__(256) void dcn_col2im_gather_kernel(const float* col, const float* offset, const float* mask,
                                                                float* grad_im, DcnGeom g, int b0, int step, int R,
                                                                int rh_max, int rw_max) {
  extern __shared__ float reg[];  // [2 + CH][rh_max * rw_max]
  const int P = g.ho * g.wo, K = g.kh * g.kw, HW = g.h * g.w;
  const int cpg = g.c / g.dg, chunks = cpg / CH;
  const int tiles_x = (g.w + DCN_GT_X - 1) / DCN_GT_X, tiles_y = (g.h + DCN_GT_Y - 1) / DCN_GT_Y;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int ck = bid % chunks; bid /= chunks;
  const int dgi = bid % g.dg, bl = bid / g.dg, b = b0 + bl;
  const int c0 = dgi * cpg + ck * CH;
  const int y0 = ty * DCN_GT_Y, x0 = tx * DCN_GT_X;
  const int x = x0 + (threadIdx.x & (DCN_GT_X - 1)), y = y0 + (threadIdx.x >> 5);
  const bool pix_ok = x < g.w && y < g.h;
  const float Rf = (float)R;
  const int plane = rh_max * rw_max;
  const float* off = offset + ((long)b * g.dg + dgi) * 2 * K * P;
  const float* mk = mask ? mask + ((long)b * g.dg + dgi) * K * P : nullptr;
  float acc[CH];
#pragma unroll
  for (int cc = 0; cc < CH; ++cc) acc[cc] = 0.f;
  for (int t = 0; t < K; ++t) {
    const int ki = t / g.kw, kj = t - ki * g.kw;
    const int hb = g.ph - ki * g.dh, wb = g.pw - kj * g.dw;
    // candidate output positions of the tile: undeformed tap position within R of some tile pixel
    const int rho0 = dcn_floor_div(y0 - R + hb + g.sh - 1, g.sh), rho1 = dcn_floor_div(y0 + DCN_GT_Y - 1 + R + hb, g.sh);
    const int rwo0 = dcn_floor_div(x0 - R + wb + g.sw - 1, g.sw), rwo1 = dcn_floor_div(x0 + DCN_GT_X - 1 + R + wb, g.sw);
    const int rh = rho1 - rho0 + 1, rw = rwo1 - rwo0 + 1;  // <= rh_max, rw_max
    const float* colt = col + dcn_cg_index(g, c0, t, bl, 0);  // [p][cpg]: the CH channels of a position are contiguous
    __syncthreads();
    for (int i = threadIdx.x; i < rh * rw; i += 256) {
      const int r = i / rw, cidx = i - r * rw;
      const int ho = rho0 + r, wo = rwo0 + cidx;
      const bool ok = ho >= 0 && ho < g.ho && wo >= 0 && wo < g.wo;
      const int p = ho * g.wo + wo;
      reg[i] = ok ? off[(long)(2 * t) * P + p] : 1e30f;
      reg[plane + i] = ok ? off[(long)(2 * t + 1) * P + p] : 1e30f;
      const float m = (ok && mk) ? mk[(long)t * P + p] : 1.f;
      float v[CH];
      if (CH % 4 == 0) {
#pragma unroll
        for (int q4 = 0; q4 < CH / 4; ++q4) {
          const float4 f = ok ? *(const float4*)(colt + (long)p * cpg + 4 * q4) : make_float4(0.f, 0.f, 0.f, 0.f);
          v[4 * q4] = f.x; v[4 * q4 + 1] = f.y; v[4 * q4 + 2] = f.z; v[4 * q4 + 3] = f.w;
        }
      } else {
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) v[cc] = ok ? colt[(long)p * cpg + cc] : 0.f;
      }
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) reg[(2 + cc) * plane + i] = v[cc] * m;
    }
    __syncthreads();
    if (!pix_ok) continue;
    int ho0 = dcn_floor_div(y - R + hb + g.sh - 1, g.sh), ho1 = dcn_floor_div(y + R + hb, g.sh);
    int wo0 = dcn_floor_div(x - R + wb + g.sw - 1, g.sw), wo1 = dcn_floor_div(x + R + wb, g.sw);
    ho0 = ho0 < rho0 ? rho0 : ho0; wo0 = wo0 < rwo0 ? rwo0 : wo0;
    ho1 = ho1 > rho1 ? rho1 : ho1; wo1 = wo1 > rwo1 ? rwo1 : wo1;
    // quick reject first: the sample's row / column must fall in [y-1, y+1) x [x-1, x+1) (exact float compares, two
    // VALU each); only the ~4 of (2R+1)^2 candidates that pass pay for the full weight computation
    const float ym1 = (float)(y - 1), yp1 = (float)(y + 1), xm1 = (float)(x - 1), xp1 = (float)(x + 1);
    for (int ho = ho0; ho <= ho1; ++ho) {
      const float nomh = (float)(ho * g.sh - g.ph + ki * g.dh);
      const int ib = (ho - rho0) * rw - rwo0;
      for (int wo = wo0; wo <= wo1; ++wo) {
        const int i = ib + wo;
        const float oh = reg[i];
        const float hi = nomh + oh;
        if (!(hi >= ym1 && hi < yp1)) continue;  // also drops positions outside the output (offset 1e30)
        const float ow = reg[plane + i];
        const float wi = (float)(wo * g.sw - g.pw + kj * g.dw) + ow;
        if (!(wi >= xm1 && wi < xp1)) continue;
        if (!(fabsf(oh) <= Rf && fabsf(ow) <= Rf)) continue;  // outlier: scattered by dcn_col2im_outlier_kernel
        if (!(hi > -1.f && wi > -1.f && hi < g.h && wi < g.w)) continue;
        const int h_low = (int)floorf(hi), w_low = (int)floorf(wi);
        const int dy = y - h_low, dx = x - w_low;  // 0 or 1
        const float lh = hi - h_low, lw = wi - w_low;
        const float wgt = (dy ? lh : 1.f - lh) * (dx ? lw : 1.f - lw);
        if (wgt == 0.f) continue;
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) acc[cc] += wgt * reg[(2 + cc) * plane + i];
      }
    }
  }
  if (pix_ok) {
    float* gim = grad_im + ((long)b * g.c + c0) * HW + y * g.w + x;
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) gim[(long)cc * HW] += acc[cc];
  }
}

// the samples the gather kernel leaves out: an offset component beyond R (or not a number)
__global__ __launch_bounds__(256) void dcn_col2im_outlier_kernel(const float* col, const float* offset, const float* mask,
                                                                 float* grad_im, DcnGeom g, int b0, int step, int R) {
  const int P = g.ho * g.wo, K = g.kh * g.kw;
  const long total = (long)step * g.dg * K * P;
  const int cpg = g.c / g.dg;
  const float Rf = (float)R;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int p = (int)(idx % P);
    const int t = (int)((idx / P) % K);
    const int dgi = (int)((idx / ((long)P * K)) % g.dg);
    const int bl = (int)(idx / ((long)P * K * g.dg)), b = b0 + bl;
    const float* off = offset + ((long)b * g.dg + dgi) * 2 * K * P;
    const float oh = off[(long)(2 * t) * P + p], ow = off[(long)(2 * t + 1) * P + p];
    if (fabsf(oh) <= Rf && fabsf(ow) <= Rf) continue;
    const int wo = p % g.wo, ho = p / g.wo, ki = t / g.kw, kj = t - ki * g.kw;
    const float hi = ho * g.sh - g.ph + ki * g.dh + oh;
    const float wi = wo * g.sw - g.pw + kj * g.dw + ow;
    if (!(hi > -1.f && wi > -1.f && hi < g.h && wi < g.w)) continue;
    const int h_low = (int)floorf(hi), w_low = (int)floorf(wi);
    const float lh = hi - h_low, lw = wi - w_low;
    const float m = mask ? mask[((long)b * g.dg + dgi) * K * P + (long)t * P + p] : 1.f;
    for (int cc = 0; cc < cpg; ++cc) {
      const int c = dgi * cpg + cc;
      const float top = col[dcn_cg_index(g, c, t, bl, p)] * m;
      float* gim = grad_im + ((long)b * g.c + c) * g.h * g.w;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int yy = h_low + dy, xx = w_low + dx;
          if (yy >= 0 && yy <= g.h - 1 && xx >= 0 && xx <= g.w - 1) {
            const float wgt = (dy ? lh : 1.f - lh) * (dx ? lw : 1.f - lw);
            if (wgt != 0.f) atomicAdd(gim + yy * g.w + xx, wgt * top);
          }
        }
    }
  }
}

__device__ __forceinline__ float dcn_coord_weight(const float* im, int H, int W, float h, float w, int dir) {
  // d bilinear / d h (dir 0) or / d w (dir 1), per-corner bounds as the forward (kernel.cu:145-188)
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h - h_low, lw = w - w_low;
  float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
  if (h_low >= 0 && w_low >= 0) v1 = im[h_low * W + w_low];
  if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * W + w_high];
  if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * W + w_low];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * W + w_high];
  if (dir == 0) return -(1.f - lw) * v1 - lw * v2 + (1.f - lw) * v3 + lw * v4;
  return -(1.f - lh) * v1 + (1.f - lh) * v2 - lh * v3 + lh * v4;
}

// grad_offset[b][dg*2K + 2t + dir][p] = sum_{c in dg} col[c,t] * (mask) * d bilinear/d coord ; grad_mask likewise
__global__ __launch_bounds__(256) void dcn_col2im_coord_kernel(const float* col, const float* x, const float* offset,
                                                               const float* mask, float* grad_offset, float* grad_mask,
                                                               DcnGeom g, int b0, int step, int use_lds) {
  // blockIdx.y: tile of 256 positions; blockIdx.x: (image, deformable group, tap).  The tile's column gradient
  // ([256 positions][cpg], contiguous) is read with coalesced 16-byte loads and transposed through LDS (row stride
  // cpg + 1: conflict-free), so that each thread then walks its own position's channels.
  extern __shared__ float tile[];
  const int P = g.ho * g.wo, K = g.kh * g.kw;
  const int cpg = g.c / g.dg;
  const bool pairs = g.w >= 2 && (long)g.c * g.h * g.w * 4 < (1L << 31) - (1L << 20);
  (void)step;
  {
    int yb = blockIdx.x;
    const int t = yb % K; yb /= K;
    const int dgi = yb % g.dg, bl = yb / g.dg;
    const int b = b0 + bl;
    const int pt0 = blockIdx.y * 256, p = pt0 + threadIdx.x;
    const float* ctile = col + dcn_cg_index(g, dgi * cpg, t, bl, pt0);
    if (use_lds) {
      const int npos = P - pt0 < 256 ? P - pt0 : 256, nfl = npos * cpg;
      if ((cpg & 3) == 0) {
        for (int i = threadIdx.x * 4; i < nfl; i += 1024) {
          const float4 v = *(const float4*)(ctile + i);
          const int pl = i / cpg, cc = i - pl * cpg;
          float* d = tile + pl * (cpg + 1) + cc;
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
      } else {
        for (int i = threadIdx.x; i < nfl; i += 256) tile[(i / cpg) * (cpg + 1) + i % cpg] = ctile[i];
      }
      __syncthreads();
    }
    if (p >= P) return;
    const float* colp = use_lds ? tile + threadIdx.x * (cpg + 1) : ctile + (long)threadIdx.x * cpg;
    const int wo = p % g.wo, ho = p / g.wo;
    const int i = t / g.kw, j = t - i * g.kw;
    const float* off = offset + ((long)b * g.dg + dgi) * 2 * K * P;
    const float hi = ho * g.sh - g.ph + i * g.dh + off[(2 * t) * P + p];
    const float wi = wo * g.sw - g.pw + j * g.dw + off[(2 * t + 1) * P + p];
    const bool inside = hi > -1.f && wi > -1.f && hi < g.h && wi < g.w;
    const float mk = mask ? mask[((long)b * g.dg + dgi) * K * P + t * P + p] : 1.f;
    float gh = 0.f, gw = 0.f, gm = 0.f;
    if (inside && pairs) {
      // corners as 8-byte column pairs through a buffer resource over the image (see dcn_im2col_group_kernel); a corner
      // outside the image counts as 0, exactly the per-corner bounds of the reference (kernel.cu:145-188)
      const __amdgpu_buffer_rsrc_t rx =
          __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)b * g.c * g.h * g.w), 0, g.c * g.h * g.w * 4, 0x00020000);
      const float hf = floorf(hi), wf = floorf(wi);
      const int h_low = (int)hf, w_low = (int)wf;
      const float lh = hi - hf, lw = wi - wf;
      const bool c0 = w_low >= 0, c1 = w_low + 1 <= g.w - 1;
      const int xs = w_low < 0 ? 0 : (c1 ? w_low : g.w - 2);
      const int base = (h_low * g.w + xs) * 4;
      const int a0 = h_low >= 0 ? base : DCN_OOB, a1 = h_low + 1 <= g.h - 1 ? base + g.w * 4 : DCN_OOB;
      for (int cc = 0; cc < cpg; ++cc) {
        const int c = dgi * cpg + cc;
        const float cv = colp[cc];
        const int coff = c * g.h * g.w * 4;
        const dcn_u32x2 p0 = __builtin_amdgcn_raw_buffer_load_b64(rx, a0 + coff, 0, 0);
        const dcn_u32x2 p1 = __builtin_amdgcn_raw_buffer_load_b64(rx, a1 + coff, 0, 0);
        const float v1 = c0 ? __uint_as_float(c1 ? p0.x : p0.y) : 0.f, v2 = c1 ? __uint_as_float(c0 ? p0.y : p0.x) : 0.f;
        const float v3 = c0 ? __uint_as_float(c1 ? p1.x : p1.y) : 0.f, v4 = c1 ? __uint_as_float(c0 ? p1.y : p1.x) : 0.f;
        gh += cv * mk * (-(1.f - lw) * v1 - lw * v2 + (1.f - lw) * v3 + lw * v4);
        gw += cv * mk * (-(1.f - lh) * v1 + (1.f - lh) * v2 - lh * v3 + lh * v4);
        if (grad_mask) gm += cv * ((1.f - lh) * (1.f - lw) * v1 + (1.f - lh) * lw * v2 + lh * (1.f - lw) * v3 + lh * lw * v4);
      }
    } else if (inside) {
      for (int cc = 0; cc < cpg; ++cc) {
        const int c = dgi * cpg + cc;
        const float cv = colp[cc];
        const float* im = x + ((long)b * g.c + c) * g.h * g.w;
        gh += cv * mk * dcn_coord_weight(im, g.h, g.w, hi, wi, 0);
        gw += cv * mk * dcn_coord_weight(im, g.h, g.w, hi, wi, 1);
        if (grad_mask) gm += cv * dcn_bilinear(im, g.h, g.w, hi, wi);
      }
    }
    float* go = grad_offset + ((long)b * g.dg + dgi) * 2 * K * P;
    go[(2 * t) * P + p] = gh;
    go[(2 * t + 1) * P + p] = gw;
    if (grad_mask) grad_mask[((long)b * g.dg + dgi) * K * P + t * P + p] = gm;
  }
}

// ------------------------------------------------------------------------------------------------ SGEMM
struct MatView {
  float* p; long s_row, s_col; int split; long s_outer;  // column index n -> (n / split) * s_outer + (n % split) * s_col
  __device__ __forceinline__ long at(int r, int cidx) const {
    return (long)r * s_row + (long)(cidx / split) * s_outer + (long)(cidx % split) * s_col;
  }
};

// C[M][N] = alpha * A[M][K] * B[K][N] + beta * C
// gridDim.z > 1: split-K -- slice z contracts k in [z*kchunk, (z+1)*kchunk) and ADDS alpha * partial to C with fp32 atomics
// (only for beta == 1: the weight-gradient GEMM, M x N = 32 x 288 against K = 655 360, was 5 blocks of 41 000 iterations).
// cgl.use: C is the column gradient in its channel-innermost layout (row m = c_local * K + t, column n = bl * P + p)
struct CgLayout { int use, c_base; DcnGeom g; };
__global__ __launch_bounds__(256) void sgemm_kernel(MatView A, MatView B, MatView C, int M, int N, int K, float alpha,
                                                    float beta, int kchunk, CgLayout cgl) {
  __shared__ float As[16][64 + 4];
  __shared__ float Bs[16][64 + 4];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const bool split = gridDim.z > 1;
  const int kbeg = split ? blockIdx.z * kchunk : 0;
  const int kend = split ? (kbeg + kchunk < K ? kbeg + kchunk : K) : K;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = threadIdx.x + r * 256;  // 1024 elements per tile
      {
        const int kk = e & 15, mm = e >> 4;  // A tile: 64 rows x 16 k
        const int m = m0 + mm, k = k0 + kk;
        As[kk][mm] = (m < M && k < kend) ? A.p[A.at(m, k)] : 0.f;
      }
      {
        const int nn = e & 63, kk = e >> 6;  // B tile: 16 k x 64 cols
        const int n = n0 + nn, k = k0 + kk;
        Bs[kk][nn] = (n < N && k < kend) ? B.p[B.at(k, n)] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m < M && n < N) {
        float* dst = C.p + C.at(m, n);
        if (cgl.use) {
          const int KK = cgl.g.kh * cgl.g.kw, PP = cgl.g.ho * cgl.g.wo;
          dst = C.p + dcn_cg_index(cgl.g, cgl.c_base + m / KK, m % KK, n / PP, n % PP);
        }
        if (split) atomicAdd(dst, alpha * acc[i][j]);
        else *dst = alpha * acc[i][j] + (beta != 0.f ? beta * *dst : 0.f);
      }
    }
}

static void sgemm(hipStream_t s, MatView A, MatView B, MatView C, int M, int N, int K, float alpha, float beta,
                  const DcnGeom* cg_geom = nullptr, int c_base = 0) {
  CgLayout cgl{};
  if (cg_geom) { cgl.use = 1; cgl.c_base = c_base; cgl.g = *cg_geom; }
  dim3 grid((N + 63) / 64, (M + 63) / 64);
  int kchunk = K;
  // few output tiles against a long contraction (and an accumulating GEMM): spread K over the chip
  if (beta == 1.f && (long)grid.x * grid.y < 128 && K >= 8192) {
    int splits = (int)(1024 / ((long)grid.x * grid.y));
    kchunk = ((K + splits - 1) / splits + 15) & ~15;
    if (kchunk < 2048) kchunk = 2048;
    grid.z = (K + kchunk - 1) / kchunk;
  }
  hipLaunchKernelGGL(sgemm_kernel, grid, dim3(256), 0, s, A, B, C, M, N, K, alpha, beta, kchunk, cgl);
}

// ------------------------------------------------------------------------------------------------ skinny MFMA GEMMs
// The two GEMMs of the DCN backward have one long dimension (step * P = 655 360 positions) and two short ones
// (cog <= 64 output channels, cg * K = 288 column rows): HBM-bound (755 MB of columns written / read), which the
// 64x64x16 VALU tile kernel above does not reach (0.87 ms each).  Both run on v_mfma_f32_32x32x2_f32 (exact fp32).
typedef float dcn_f32x16 __attribute__((ext_vector_type(16)));
typedef float dcn_f32x4 __attribute__((ext_vector_type(4)));
typedef float dcn_f32x2 __attribute__((ext_vector_type(2)));

// columns[r][n] = sum_co W[co][r] * gO[b0 + n / P][co][n % P]      (r < ck = cg * K, n < step * P)
// A = W^T from LDS ([co][r], one ds_read_b32 per MFMA), B = gO straight from global memory (a half-wave reads 32
// consecutive positions of one channel = 128 contiguous bytes) held in registers for all row tiles; a wave owns 64
// positions and walks the ck / 32 row tiles; each accumulator row is a 128-byte contiguous store.
template <int KP>  // KP = ceil(cog / 2) k-steps held in registers
__global__ __launch_bounds__(256) void dcn_colgrad_mfma_kernel(const float* w, const float* go, float* cols, int cog,
                                                               int rows_per_block, int c_base, int cg, DcnGeom g, long N) {
  extern __shared__ float wl[];  // [2 * KP][ckp] (rows >= cog and columns >= ck are zero)
  // Rows are walked tap-major (r' = t * cg + c_local) so that a lane's four consecutive accumulator rows are four
  // consecutive channels of one tap: one 16-byte store into the channel-innermost column gradient.
  // blockIdx.y: chunk of rows (the weight slice of a chunk must fit LDS; the gO tile is re-read per chunk)
  const int K = g.kh * g.kw, P = g.ho * g.wo, ck_total = cg * K;
  const int r0 = blockIdx.y * rows_per_block;
  const int ck = ck_total - r0 < rows_per_block ? ck_total - r0 : rows_per_block;
  const int ckp = (ck + 31) / 32 * 32;
  for (int i = threadIdx.x; i < 2 * KP * ckp; i += 256) {
    const int co = i / ckp, r = i - co * ckp, rp = r0 + r;
    wl[i] = (co < cog && r < ck) ? w[(long)co * ck_total + (rp % cg) * K + rp / cg] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l32 = lane & 31;
  const long n0 = (blockIdx.x * 4L + wave) * 64;
  float bv[2][KP];
  bool nok[2];
  int nbl[2], np_[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const long nn = n0 + 32 * q + l32;
    nok[q] = nn < N;
    nbl[q] = nok[q] ? (int)(nn / P) : 0;
    np_[q] = nok[q] ? (int)(nn - (long)nbl[q] * P) : 0;
    const float* src = go + ((long)nbl[q] * g.co) * P + np_[q];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      const int co = 2 * kk + half;
      bv[q][kk] = (nok[q] && co < cog) ? src[(long)co * P] : 0.f;
    }
  }
  const bool vec = (cg & 3) == 0 && ((g.c / g.dg) & 3) == 0 && (c_base & 3) == 0;
  for (int rt = 0; rt < ckp / 32; ++rt) {
    dcn_f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      const float a = wl[(2 * kk + half) * ckp + rt * 32 + l32];
#pragma unroll
      for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[q][kk], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (!nok[q]) continue;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int row = rt * 32 + 8 * r4 + 4 * half;  // rows row .. row+3 = registers 4*r4 .. 4*r4+3
        if (row >= ck) continue;
        const int rp = r0 + row, t = rp / cg, cl = rp - t * cg;
        if (vec) {
          *(float4*)(cols + dcn_cg_index(g, c_base + cl, t, nbl[q], np_[q])) =
              make_float4(acc[q][4 * r4], acc[q][4 * r4 + 1], acc[q][4 * r4 + 2], acc[q][4 * r4 + 3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int rj = rp + j;
            if (row + j < ck) cols[dcn_cg_index(g, c_base + rj % cg, rj / cg, nbl[q], np_[q])] = acc[q][4 * r4 + j];
          }
        }
      }
    }
  }
}

// gradW[co][r] += scale * sum_n gO[b0 + n / P][co][n % P] * cols[r][n]     (co < cog <= 32, r < ck <= 288; P % 64 == 0)
// The contraction runs over positions, contiguous in memory for BOTH operands, while the MFMA wants lane l to hold
// (row l % 32, k = l / 32).  Summation order is free, so within a unit of 64 positions lane-half h takes positions
// 32h .. 32h+31 of its row as eight 16-byte loads: the first load of a 128-byte line misses, the seven issued right
// behind it hit L1, and every line crosses L2 once.  A wave keeps all ck / 32 column tiles as accumulators (<= 144
// registers), walks its share of the 64-position units and adds its partial result to gradW with fp32 atomics.
#define DCN_GW_TILES 9
__global__ __launch_bounds__(64) void dcn_gradw_mfma_kernel(const float* go, const float* cols, float* gw, int cog,
                                                            int ck_total, int co_total, int P, long N, float scale,
                                                            int units_per_wave) {
  const int lane = threadIdx.x, half = lane >> 5, l32 = lane & 31;
  // blockIdx.y: chunk of 9 column tiles (288 rows of the column matrix); the gO tile is re-read per chunk
  const int r0 = blockIdx.y * 32 * DCN_GW_TILES;
  const int ck = ck_total - r0 < 32 * DCN_GW_TILES ? ck_total - r0 : 32 * DCN_GW_TILES;
  cols += (long)r0 * N;
  gw += r0;
  const int ntiles = (ck + 31) / 32;
  dcn_f32x16 acc[DCN_GW_TILES];
#pragma unroll
  for (int t = 0; t < DCN_GW_TILES; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const long units = N / 64;
  const long u0 = (long)blockIdx.x * units_per_wave;
  for (long u = u0; u < u0 + units_per_wave && u < units; ++u) {
    const long n = u * 64 + 32 * half;  // this half's 32 positions (one image: P % 64 == 0)
    const long bl = n / P;
    const int p = (int)(n - bl * P);
    float4 a4[8];
    {
      const float* src = go + (bl * co_total + l32) * P + p;
#pragma unroll
      for (int j = 0; j < 8; ++j) a4[j] = l32 < cog ? *(const float4*)(src + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int t = 0; t < DCN_GW_TILES; ++t) {
      if (t >= ntiles) break;
      const int row = t * 32 + l32;
      float4 b4[8];
      const float* src = cols + (long)row * N + n;
#pragma unroll
      for (int j = 0; j < 8; ++j) b4[j] = row < ck ? *(const float4*)(src + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].x, b4[j].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].y, b4[j].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].z, b4[j].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].w, b4[j].w, acc[t], 0, 0, 0);
      }
    }
  }
  // D[m = co][n = r]: lane l, register i -> n = l % 32, m = 8 * (i / 4) + 4 * (l / 32) + i % 4
#pragma unroll
  for (int t = 0; t < DCN_GW_TILES; ++t) {
    if (t >= ntiles) break;
    const int r = t * 32 + l32;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = 8 * (i >> 2) + 4 * half + (i & 3);
      if (co < cog && r < ck) atomicAdd(gw + (long)co * ck_total + r, scale * acc[t][i]);
    }
  }
}

// dispatchers: true when the MFMA kernel took the GEMM
static bool dcn_colgrad_mfma(hipStream_t s, const float* w, const float* go, float* cols, int cog, int c_base, int cg,
                             const DcnGeom& g, long N) {
  if (cog > 64) return false;
  const int ck = cg * g.kh * g.kw;
  const int kp = (cog + 1) / 2;
  const int kpt = kp <= 8 ? 8 : kp <= 16 ? 16 : 32;
  int rpb = (48 * 1024 / (2 * kpt * (int)sizeof(float))) / 32 * 32;  // rows per block: weight slice <= 48 KB of LDS
  if (rpb > (ck + 31) / 32 * 32) rpb = (ck + 31) / 32 * 32;
  const size_t lds = (size_t)2 * kpt * rpb * sizeof(float);
  const dim3 grid((unsigned)((N + 255) / 256), (unsigned)((ck + rpb - 1) / rpb));
  if (kpt == 8) hipLaunchKernelGGL(dcn_colgrad_mfma_kernel<8>, grid, dim3(256), lds, s, w, go, cols, cog, rpb, c_base, cg, g, N);
  else if (kpt == 16) hipLaunchKernelGGL(dcn_colgrad_mfma_kernel<16>, grid, dim3(256), lds, s, w, go, cols, cog, rpb, c_base, cg, g, N);
  else hipLaunchKernelGGL(dcn_colgrad_mfma_kernel<32>, grid, dim3(256), lds, s, w, go, cols, cog, rpb, c_base, cg, g, N);
  return true;
}

static bool dcn_gradw_mfma(hipStream_t s, const float* go, const float* cols, float* gw, int cog, int ck, int co_total,
                           int P, long N, float scale) {
  if (cog > 32 || P % 64) return false;
  const long units = N / 64;
  int upw = (int)((units + 1023) / 1024);  // ~1024 waves: one per SIMD
  if (upw < 1) upw = 1;
  const dim3 blocks((unsigned)((units + upw - 1) / upw), (unsigned)((ck + 32 * DCN_GW_TILES - 1) / (32 * DCN_GW_TILES)));
  hipLaunchKernelGGL(dcn_gradw_mfma_kernel, blocks, dim3(64), 0, s, go, cols, gw, cog, ck, co_total, P, N, scale, upw);
  return true;
}

static inline MatView mv(const float* p, long s_row, long s_col, int split = 1 << 30, long s_outer = 0) {
  return MatView{const_cast<float*>(p), s_row, s_col, split, s_outer};
}

static inline int grid1d(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

__global__ void dcn_bias_kernel(float* out, const float* bias, int co, int P, long total, int add) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)((i / P) % co);
    out[i] = add ? out[i] + bias[c] : bias[c];
  }
}

// grad_bias[c] += sum over (n, p) of grad_output : one block per channel
__global__ __launch_bounds__(256) void dcn_bias_grad_kernel(const float* go, float* gb, int n, int co, int P) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  float acc = 0.f;
  for (long i = threadIdx.x; i < (long)n * P; i += 256) acc += go[((i / P) * co + c) * P + i % P];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) gb[c] += red[0];
}

// ------------------------------------------------------------------------------------------------ fused forward
// Deformable im2col fused with the GEMM: no `columns` round trip (the unfused forward writes and re-reads 1.5 GB of
// columns at [128,32,64,160]).  out[b][co][p] = sum_{c,t} W[co][c][t] * sample(x[b][c], p, t) on the fp32 matrix pipe:
// v_mfma_f32_32x32x2_f32 with A = W (32 co x 2 channels) and B = the gathered samples (2 channels x 32 positions).  In
// that instruction's B layout lane l holds position l%32 of channel l/32 -- so one wave-wide gather (4 corner loads per
// lane, lanes of a half-wave on consecutive positions of one channel plane) feeds one MFMA per 32 output channels.
// A wave owns 64 consecutive output positions (two 32-position MFMA column groups); sample coordinates and bilinear
// weights are computed once per (position, deformable group, tap) and reused by all channels of the group.  Weights are
// pre-transposed to [c][tap][co] (dcn_wt_kernel, into the caller's workspace) and staged per deformable group in LDS
// with a row stride that puts the two half-waves on disjoint banks.  Corner loads go through a buffer resource over the
// image, so an out-of-range corner is a voffset past num_records and reads 0 (the reference's per-corner bounds,
// deform_conv_cuda_kernel.cu:85-115).

__global__ void dcn_wt_kernel(const float* w, float* wt, int co, int c, int K, int cop) {
  const int total = c * K * cop;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int o = i % cop, ck = i / cop;
    wt[i] = o < co ? w[(long)o * c * K + ck] : 0.f;
  }
}

#define DCN_FWD_THREADS 512
template <int MT>
__global__ __launch_bounds__(DCN_FWD_THREADS) void dcn_fused_fwd_kernel(const float* x, const float* offset, const float* mask,
                                                            const float* wt, const float* bias, float* out, DcnGeom g,
                                                            int rs) {
  extern __shared__ float wl[];
  constexpr int COP = 32 * MT;
  const int P = g.ho * g.wo, K = g.kh * g.kw, HW = g.h * g.w;
  const int cpg = g.c / g.dg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * DCN_FWD_THREADS + wave * 64;
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)b * g.c * HW), 0, g.c * HW * 4, 0x00020000);
  const int pl = p0 + lane;
  const bool pl_ok = pl < P;
  int hin[2], win[2];
  bool pg_ok[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int pg = p0 + 32 * q + l32;
    const int ho = pg / g.wo, wo = pg - ho * g.wo;
    hin[q] = ho * g.sh - g.ph;
    win[q] = wo * g.sw - g.pw;
    pg_ok[q] = pg < P;
  }
  dcn_f32x16 acc[2][MT];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][m][r] = 0.f;
  const float* offb = offset + (long)b * g.dg * 2 * K * P;
  const float* mkb = mask ? mask + (long)b * g.dg * K * P : nullptr;
  const int stage_n = cpg * K * COP;
  for (int dgi = 0; dgi < g.dg; ++dgi) {
    __syncthreads();
    for (int i = threadIdx.x * 4; i < stage_n; i += DCN_FWD_THREADS * 4) {
      const float4 v = *(const float4*)(wt + (long)dgi * stage_n + i);
      const int cl = i / (K * COP), r = i - cl * (K * COP);
      *(float4*)&wl[cl * rs + r] = v;
    }
    __syncthreads();
    const float* offd = offb + (long)dgi * 2 * K * P;
    for (int t = 0; t < K; ++t) {
      const int ki = t / g.kw, kj = t - ki * g.kw;
      const float oh = pl_ok ? offd[(long)(2 * t) * P + pl] : 0.f;
      const float ow = pl_ok ? offd[(long)(2 * t + 1) * P + pl] : 0.f;
      const float mv_ = (mkb && pl_ok) ? mkb[((long)dgi * K + t) * P + pl] : 1.f;
      float w1[2], w2[2], w3[2], w4[2], mk[2];
      int a0[2], a1[2];
      bool lo_y[2], hi_x[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int src = (32 * q + l32) * 4;
        const float ohq = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(oh)));
        const float owq = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(ow)));
        mk[q] = mkb ? __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(mv_))) : 1.f;
        const float hi = hin[q] + ki * g.dh + ohq;
        const float wi = win[q] + kj * g.dw + owq;
        const bool in = pg_ok[q] && hi > -1.f && wi > -1.f && hi < g.h && wi < g.w;
        const float hf = floorf(hi), wf = floorf(wi);
        const int h_low = (int)hf, w_low = (int)wf;
        const float lh = hi - hf, lw = wi - wf, hh = 1.f - lh, hw = 1.f - lw;
        // one 8-byte load per row fetches both columns: the pair starts at xs = clamp(w_low, 0, W-2); at the left edge
        // (w_low = -1) the high column is the pair's .x, at the right edge (w_low = W-1) the low column is its .y, and
        // the column that falls outside the image gets weight 0
        const bool c0 = w_low >= 0, c1 = w_low + 1 <= g.w - 1;
        lo_y[q] = !c1;
        hi_x[q] = !c0;
        w1[q] = c0 ? hh * hw : 0.f; w2[q] = c1 ? hh * lw : 0.f; w3[q] = c0 ? lh * hw : 0.f; w4[q] = c1 ? lh * lw : 0.f;
        const int xs = w_low < 0 ? 0 : (c1 ? w_low : g.w - 2);
        const bool r0 = in && h_low >= 0, r1 = in && h_low + 1 <= g.h - 1;
        const int base = (h_low * g.w + xs + half * HW) * 4;
        a0[q] = r0 ? base : DCN_OOB;
        a1[q] = r1 ? base + g.w * 4 : DCN_OOB;
      }
      // four channel pairs per round: all 16 gathers are issued before the first blend, so a wave keeps 16 loads in
      // flight (one round at cpg = 8); the L2-hit latency of a dependent gather is what bounds this kernel otherwise
      for (int cq = 0; cq < cpg; cq += 8) {
        dcn_u32x2 pr[4][2][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool live = cq + 2 * u < cpg;
          const int coff = live ? (dgi * cpg + cq + 2 * u) * HW * 4 : 0;  // OOB markers stay past num_records after the add
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            pr[u][q][0] = __builtin_amdgcn_raw_buffer_load_b64(rx, live ? a0[q] + coff : DCN_OOB, 0, 0);
            pr[u][q][1] = __builtin_amdgcn_raw_buffer_load_b64(rx, live ? a1[q] + coff : DCN_OOB, 0, 0);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (cq + 2 * u >= cpg) break;
          float bv[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const dcn_u32x2 p0_ = pr[u][q][0], p1_ = pr[u][q][1];
            const float v1 = __uint_as_float(lo_y[q] ? p0_.y : p0_.x), v2 = __uint_as_float(hi_x[q] ? p0_.x : p0_.y);
            const float v3 = __uint_as_float(lo_y[q] ? p1_.y : p1_.x), v4 = __uint_as_float(hi_x[q] ? p1_.x : p1_.y);
            bv[q] = (w1[q] * v1 + w2[q] * v2 + w3[q] * v3 + w4[q] * v4) * mk[q];
          }
          const float* wrow = wl + (cq + 2 * u + half) * rs + t * COP + l32;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float a = wrow[m * 32];
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[q], acc[q][m], 0, 0, 0);
          }
        }
      }
    }
  }
  // D[m][n]: lane l, register r -> n = l%32 (position), m = 8*(r/4) + 4*(l/32) + r%4 (output channel within the tile)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int pg = p0 + 32 * q + l32;
    if (pg >= P) continue;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
        if (co < g.co) out[((long)b * g.co + co) * P + pg] = acc[q][m][r] + (bias ? bias[co] : 0.f);
      }
  }
}

// ------------------------------------------------------------------------------------------------ windowed forward
// The gather kernel above is bound by the texture-address path (about four lane addresses per clock per CU whatever
// the load width), not by HBM or the matrix pipe.  Learned offsets are small, so the samples of a tile of 512 consecutive
// output positions fall in a band of input rows: this kernel stages that band ("window": the rows the undeformed taps
// touch plus `halo` rows either side, full width, rows outside the image as zeros) for `nch` channels in LDS with
// coalesced 16-byte loads and gathers the bilinear corners from LDS (two ds_read2_b32 per sample).  A sample whose rows
// leave the window is fetched with the buffer loads of the gather kernel instead -- per lane, under a wave-uniform
// branch that is not taken when no lane of the wave missed -- so any offset field gives the reference's result; only
// the speed depends on the offsets being small.
// Both halves of a wave get the value the lower (q = 0) / upper (q = 1) half computed: v_permlane32_swap of a register
// with a copy of itself yields (lo, lo) and (hi, hi).
__device__ __forceinline__ void dcn_bcast_halves(unsigned v, unsigned (&o)[2]) {
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  o[0] = r[0];
  o[1] = r[1];
}
__device__ __forceinline__ void dcn_bcast_halves(float v, float (&o)[2]) {
  unsigned u[2];
  dcn_bcast_halves(__float_as_uint(v), u);
  o[0] = __uint_as_float(u[0]);
  o[1] = __uint_as_float(u[1]);
}

struct DcnWin {
  int rs, halo, wr, ws, nch;  // weight row stride, halo rows, window rows, per-channel window stride (floats), channels per stage
  int wrs;                    // padded-row variant: window row stride (W + 4), 0 = contiguous rows
};

template <int MT>
__global__ __launch_bounds__(DCN_FWD_THREADS) void dcn_win_fwd_kernel(const float* x, const float* offset, const float* mask,
                                                                      const float* wt, const float* bias, float* out,
                                                                      DcnGeom g, DcnWin wn) {
  extern __shared__ float lds[];
  constexpr int COP = 32 * MT;
  float* win = lds;                    // [nch][ws]
  float* wl = lds + wn.nch * wn.ws;    // [nch][rs]
  const int P = g.ho * g.wo, K = g.kh * g.kw, HW = g.h * g.w;
  const int cpg = g.c / g.dg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.y;
  const int pfirst = blockIdx.x * DCN_FWD_THREADS;
  const int p0 = pfirst + wave * 64;
  const int wy0 = (pfirst / g.wo) * g.sh - g.ph - wn.halo;  // first staged input row (may be negative)
  const float* xb = x + (long)b * g.c * HW;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, g.c * HW * 4, 0x00020000);
  const int pl = p0 + lane;
  const bool pl_ok = pl < P;
  const int ho_l = pl / g.wo, wo_l = pl - ho_l * g.wo;
  const int hin_l = ho_l * g.sh - g.ph, win_l = wo_l * g.sw - g.pw;
  dcn_f32x16 acc[2][MT];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][m][r] = 0.f;
  const float* offb = offset + (long)b * g.dg * 2 * K * P;
  const float* mkb = mask ? mask + (long)b * g.dg * K * P : nullptr;
  const int wcount = wn.wr * g.w;      // floats per staged channel
  const int f0 = wy0 * g.w;            // image-plane index of window element 0
  const bool vec4 = (g.w & 3) == 0;
  float oh_n = pl_ok ? offb[pl] : 0.f, ow_n = pl_ok ? offb[P + pl] : 0.f;
  float mv_n = (mkb && pl_ok) ? mkb[pl] : 1.f;
  for (int dgi = 0; dgi < g.dg; ++dgi) {
    for (int c0 = 0; c0 < cpg; c0 += wn.nch) {
      const int cabs = dgi * cpg + c0;
      __syncthreads();
      if (vec4) {
        // four 16-byte loads in flight per thread before the first LDS write (a stage costs ~2 memory latencies, not 8)
        const int q4 = wcount >> 2, passes = (q4 + DCN_FWD_THREADS - 1) / DCN_FWD_THREADS, items = wn.nch * passes;
        for (int it0 = 0; it0 < items; it0 += 4) {
          float4 v[4];
          int dst[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int it = it0 + u, ch = it / passes, e = ((it - ch * passes) * DCN_FWD_THREADS + threadIdx.x) * 4;
            const int f = f0 + e;
            const bool ok = it < items && e < wcount;
            dst[u] = ok ? ch * wn.ws + e : -1;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && f >= 0 && f < HW) v[u] = *(const float4*)(xb + (long)(cabs + ch) * HW + f);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (dst[u] >= 0) *(float4*)&win[dst[u]] = v[u];
        }
      } else {
        for (int i = threadIdx.x; i < wn.nch * wcount; i += DCN_FWD_THREADS) {
          const int ch = i / wcount, e = i - ch * wcount, f = f0 + e;
          win[ch * wn.ws + e] = (f >= 0 && f < HW) ? xb[(long)(cabs + ch) * HW + f] : 0.f;
        }
      }
      for (int i = threadIdx.x * 4; i < wn.nch * K * COP; i += DCN_FWD_THREADS * 4) {
        const float4 v = *(const float4*)(wt + (long)cabs * K * COP + i);
        const int cl = i / (K * COP), r = i - cl * (K * COP);
        *(float4*)&wl[cl * wn.rs + r] = v;
      }
      __syncthreads();
      for (int t = 0; t < K; ++t) {
        const int ki = t / g.kw, kj = t - ki * g.kw;
        const float oh = oh_n, ow = ow_n, mv_ = mv_n;
        {  // the next tap's offsets (and mask) are loaded now and consumed one tap later
          int tn = t + 1, dn = dgi;
          if (tn == K) { tn = 0; if (c0 + wn.nch >= cpg) ++dn; }
          if (dn < g.dg && pl_ok) {
            const float* on = offb + ((long)dn * 2 * K + 2 * tn) * P + pl;
            oh_n = on[0];
            ow_n = on[P];
            if (mkb) mv_n = mkb[((long)dn * K + tn) * P + pl];
          }
        }
        // each lane works out ONE position's sample (its own: p0 + lane); the two 32-position MFMA column groups pick
        // their values up with v_permlane32_swap: (lo, lo) and (hi, hi) of a register in one VALU instruction
        float w1[2], w2[2], w3[2], w4[2], mk[2];
        int la[2], ga0[2], ga1[2];
        bool lo_y[2], hi_x[2], miss[2];
        bool anymiss;
        {
          const float hi = hin_l + ki * g.dh + oh;
          const float wi = win_l + kj * g.dw + ow;
          const bool in = pl_ok && hi > -1.f && wi > -1.f && hi < g.h && wi < g.w;
          const float hf = floorf(hi), wf = floorf(wi);
          const int h_low = (int)hf, w_low = (int)wf;
          const float lh = hi - hf, lw = wi - wf, hh = 1.f - lh, hw = 1.f - lw;
          const bool c0_ = in && w_low >= 0, c1_ = in && w_low + 1 <= g.w - 1;
          const bool loy = w_low + 1 > g.w - 1, hix = w_low < 0;
          const int xs = hix ? 0 : (loy ? g.w - 2 : w_low);
          const int ry = h_low - wy0;
          const bool inwin = ry >= 0 && ry + 1 < wn.wr;  // both rows are staged (rows outside the image as zeros)
          const bool missl = in && !inwin;
          unsigned u[2];
          dcn_bcast_halves((in && inwin) ? (unsigned)(ry * g.w + xs) * 4u : 0u, u);
          la[0] = (int)u[0] + half * wn.ws * 4; la[1] = (int)u[1] + half * wn.ws * 4;
          dcn_bcast_halves(c0_ ? hh * hw : 0.f, w1);
          dcn_bcast_halves(c1_ ? hh * lw : 0.f, w2);
          dcn_bcast_halves(c0_ ? lh * hw : 0.f, w3);
          dcn_bcast_halves(c1_ ? lh * lw : 0.f, w4);
          dcn_bcast_halves((loy ? 1u : 0u) | (hix ? 2u : 0u) | (missl ? 4u : 0u), u);
#pragma unroll
          for (int q = 0; q < 2; ++q) { lo_y[q] = u[q] & 1u; hi_x[q] = u[q] & 2u; miss[q] = u[q] & 4u; }
          mk[0] = 1.f; mk[1] = 1.f;
          if (mkb) dcn_bcast_halves(mv_, mk);
          anymiss = __builtin_amdgcn_ballot_w64(missl) != 0;
          ga0[0] = ga0[1] = ga1[0] = ga1[1] = DCN_OOB;
          if (anymiss) {  // global pair addresses of the lanes whose sample left the window (others stay out of range)
            const int gbase = (h_low * g.w + xs) * 4;
            dcn_bcast_halves((missl && h_low >= 0) ? (unsigned)gbase : (unsigned)DCN_OOB, u);
            ga0[0] = (int)u[0] + half * HW * 4; ga0[1] = (int)u[1] + half * HW * 4;
            dcn_bcast_halves((missl && h_low + 1 <= g.h - 1) ? (unsigned)(gbase + g.w * 4) : (unsigned)DCN_OOB, u);
            ga1[0] = (int)u[0] + half * HW * 4; ga1[1] = (int)u[1] + half * HW * 4;
          }
        }
        for (int cq = 0; cq < wn.nch; cq += 2) {
          float bv[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float* r0 = (const float*)((const char*)win + la[q]) + cq * wn.ws;
            const float* r1 = r0 + g.w;
            float p0x = r0[0], p0y = r0[1], p1x = r1[0], p1y = r1[1];
            if (anymiss) {
              const int coff = (cabs + cq) * HW * 4;
              const dcn_u32x2 b0 = __builtin_amdgcn_raw_buffer_load_b64(rx, ga0[q] + coff, 0, 0);
              const dcn_u32x2 b1 = __builtin_amdgcn_raw_buffer_load_b64(rx, ga1[q] + coff, 0, 0);
              if (miss[q]) {
                p0x = __uint_as_float(b0.x); p0y = __uint_as_float(b0.y);
                p1x = __uint_as_float(b1.x); p1y = __uint_as_float(b1.y);
              }
            }
            const float v1 = lo_y[q] ? p0y : p0x, v2 = hi_x[q] ? p0x : p0y;
            const float v3 = lo_y[q] ? p1y : p1x, v4 = hi_x[q] ? p1x : p1y;
            bv[q] = (w1[q] * v1 + w2[q] * v2 + w3[q] * v3 + w4[q] * v4) * mk[q];
          }
          const float* wrow = wl + (cq + half) * wn.rs + t * COP + l32;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float a = wrow[m * 32];
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[q], acc[q][m], 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int pg = p0 + 32 * q + l32;
    if (pg >= P) continue;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
        if (co < g.co) out[((long)b * g.co + co) * P + pg] = acc[q][m][r] + (bias ? bias[co] : 0.f);
      }
  }
}

// Padded-row variant of the window kernel (W % 4 == 0): window row ry starts at ry * wrs with wrs = W + 4; floats 0..3 of
// a row are zero and column x sits at 4 + x, so the column pair (w_low, w_low + 1) is addressable for w_low in
// [-1, W-1] (x = -1 is the row's own pad, x = W the next row's pad or the 4-float tail).  With rows outside the image
// staged as zeros a sample inside the window needs no per-corner logic at all: two ds_read2_b32 and four FMAs; an
// invalid sample points at the zero pad of row 0 with zero weights.  Samples that leave the window contribute nothing
// in the main loop and are added by a second pass over the channel pairs (global 8-byte pair loads) that only runs for
// waves that have such a lane.  The mask (v2) is folded into the four bilinear weights.
template <int MT>
__global__ __launch_bounds__(DCN_FWD_THREADS) void dcn_winp_fwd_kernel(const float* x, const float* offset, const float* mask,
                                                                       const float* wt, const float* bias, float* out,
                                                                       DcnGeom g, DcnWin wn) {
  extern __shared__ float lds[];
  constexpr int COP = 32 * MT;
  float* win = lds;                    // [nch][ws]
  float* wl = lds + wn.nch * wn.ws;    // [nch][rs]
  const int P = g.ho * g.wo, K = g.kh * g.kw, HW = g.h * g.w;
  const int cpg = g.c / g.dg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.y;
  const int pfirst = blockIdx.x * DCN_FWD_THREADS;
  const int p0 = pfirst + wave * 64;
  const int wy0 = (pfirst / g.wo) * g.sh - g.ph - wn.halo;  // first staged input row (may be negative)
  const float* xb = x + (long)b * g.c * HW;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, g.c * HW * 4, 0x00020000);
  const int pl = p0 + lane;
  const bool pl_ok = pl < P;
  const int ho_l = pl / g.wo, wo_l = pl - ho_l * g.wo;
  const int hin_l = ho_l * g.sh - g.ph, win_l = wo_l * g.sw - g.pw;
  dcn_f32x16 acc[2][MT];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][m][r] = 0.f;
  const float* offb = offset + (long)b * g.dg * 2 * K * P;
  const float* mkb = mask ? mask + (long)b * g.dg * K * P : nullptr;
  // staging: thread -> (row group st_r, 16-byte slot st_j of a row) once; slot 0 is the row's zero pad
  const int slots = wn.wrs >> 2, rpp = DCN_FWD_THREADS / slots;
  const int st_j = threadIdx.x % slots, st_r = threadIdx.x / slots;
  const int nrows = wn.nch * wn.wr;
  const int half_off = half * wn.ws * 4, wrs4 = wn.wrs * 4;
  float oh_n = pl_ok ? offb[pl] : 0.f, ow_n = pl_ok ? offb[P + pl] : 0.f;
  float mv_n = (mkb && pl_ok) ? mkb[pl] : 1.f;
  for (int dgi = 0; dgi < g.dg; ++dgi) {
    for (int c0 = 0; c0 < cpg; c0 += wn.nch) {
      const int cabs = dgi * cpg + c0;
      __syncthreads();
      if (st_r < rpp) {
        for (int rb = st_r; rb < nrows; rb += 4 * rpp) {  // four rows' loads in flight before the first LDS write
          float4 v[4];
          int dst[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int rr = rb + u * rpp, ch = rr / wn.wr, ry = rr - ch * wn.wr, y = wy0 + ry;
            dst[u] = rr < nrows ? ch * wn.ws + ry * wn.wrs + st_j * 4 : -1;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rr < nrows && st_j > 0 && y >= 0 && y < g.h)
              v[u] = *(const float4*)(xb + (long)(cabs + ch) * HW + y * g.w + (st_j - 1) * 4);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (dst[u] >= 0) *(float4*)&win[dst[u]] = v[u];
        }
      }
      if (threadIdx.x < wn.nch * 4) win[(threadIdx.x >> 2) * wn.ws + wn.wr * wn.wrs + (threadIdx.x & 3)] = 0.f;  // tail pad
      for (int i = threadIdx.x * 4; i < wn.nch * K * COP; i += DCN_FWD_THREADS * 4) {
        const float4 v = *(const float4*)(wt + (long)cabs * K * COP + i);
        const int cl = i / (K * COP), r = i - cl * (K * COP);
        *(float4*)&wl[cl * wn.rs + r] = v;
      }
      __syncthreads();
      for (int t = 0; t < K; ++t) {
        const int ki = t / g.kw, kj = t - ki * g.kw;
        const float oh = oh_n, ow = ow_n, mv_ = mv_n;
        {  // the next tap's offsets (and mask) are loaded now and consumed one tap later
          int tn = t + 1, dn = dgi;
          if (tn == K) { tn = 0; if (c0 + wn.nch >= cpg) ++dn; }
          if (dn < g.dg && pl_ok) {
            const float* on = offb + ((long)dn * 2 * K + 2 * tn) * P + pl;
            oh_n = on[0];
            ow_n = on[P];
            if (mkb) mv_n = mkb[((long)dn * K + tn) * P + pl];
          }
        }
        // this lane's own position; the two 32-position column groups pick the values up with v_permlane32_swap
        const float hi = hin_l + ki * g.dh + oh, wi = win_l + kj * g.dw + ow;
        const bool in = pl_ok && hi > -1.f && wi > -1.f && hi < g.h && wi < g.w;
        const float hf = floorf(hi), wf = floorf(wi);
        const int h_low = (int)hf, w_low = (int)wf;
        const float lh = hi - hf, lw = wi - wf, hh = 1.f - lh, hw = 1.f - lw;
        const int ry = h_low - wy0;
        const bool inwin = ry >= 0 && ry + 1 < wn.wr;  // both rows are staged (rows outside the image as zeros)
        const bool missl = in && !inwin;
        const float wa = in ? hh * hw * mv_ : 0.f, wb = in ? hh * lw * mv_ : 0.f;
        const float wc = in ? lh * hw * mv_ : 0.f, wd = in ? lh * lw * mv_ : 0.f;
        float w1[2], w2[2], w3[2], w4[2];
        unsigned la[2];
        dcn_bcast_halves((in && inwin) ? (unsigned)(ry * wn.wrs + 4 + w_low) * 4u : 0u, la);
        dcn_bcast_halves(wa, w1);
        dcn_bcast_halves(wb, w2);
        dcn_bcast_halves(wc, w3);
        dcn_bcast_halves(wd, w4);
        const float* wrow = wl + half * wn.rs + t * COP + l32;
        const char* wb0 = (const char*)win + half_off;
        for (int cq = 0; cq < wn.nch; cq += 2) {
          float bv[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float* r0 = (const float*)(wb0 + la[q]) + cq * wn.ws;
            const float* r1 = (const float*)((const char*)r0 + wrs4);
            bv[q] = w1[q] * r0[0] + w2[q] * r0[1] + w3[q] * r1[0] + w4[q] * r1[1];
          }
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float a = wrow[cq * wn.rs + m * 32];
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[q], acc[q][m], 0, 0, 0);
          }
        }
        if (__builtin_amdgcn_ballot_w64(missl) != 0) {
          // second pass for the lanes whose sample left the window: their corners from global memory as 8-byte pairs
          // starting at clamp(w_low, 0, W-2), the column outside the image with weight 0; every other lane adds 0
          const bool c0_ = w_low >= 0, c1_ = w_low + 1 <= g.w - 1;
          const int xs = !c0_ ? 0 : (c1_ ? w_low : g.w - 2);
          const int gbase = (h_low * g.w + xs) * 4;
          unsigned ga0[2], ga1[2], fl[2];
          float m1[2], m2[2], m3[2], m4[2];
          dcn_bcast_halves((missl && h_low >= 0) ? (unsigned)gbase : (unsigned)DCN_OOB, ga0);
          dcn_bcast_halves((missl && h_low + 1 <= g.h - 1) ? (unsigned)(gbase + g.w * 4) : (unsigned)DCN_OOB, ga1);
          dcn_bcast_halves((c1_ ? 0u : 1u) | (c0_ ? 0u : 2u), fl);
          dcn_bcast_halves((missl && c0_) ? wa : 0.f, m1);
          dcn_bcast_halves((missl && c1_) ? wb : 0.f, m2);
          dcn_bcast_halves((missl && c0_) ? wc : 0.f, m3);
          dcn_bcast_halves((missl && c1_) ? wd : 0.f, m4);
          for (int cq = 0; cq < wn.nch; cq += 2) {
            float bv[2];
            const int coff = (cabs + cq + half) * HW * 4;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const dcn_u32x2 b0 = __builtin_amdgcn_raw_buffer_load_b64(rx, (int)ga0[q] + coff, 0, 0);
              const dcn_u32x2 b1 = __builtin_amdgcn_raw_buffer_load_b64(rx, (int)ga1[q] + coff, 0, 0);
              const bool loy = fl[q] & 1u, hix = fl[q] & 2u;
              const float v1 = __uint_as_float(loy ? b0.y : b0.x), v2 = __uint_as_float(hix ? b0.x : b0.y);
              const float v3 = __uint_as_float(loy ? b1.y : b1.x), v4 = __uint_as_float(hix ? b1.x : b1.y);
              bv[q] = m1[q] * v1 + m2[q] * v2 + m3[q] * v3 + m4[q] * v4;
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const float a = wrow[cq * wn.rs + m * 32];
#pragma unroll
              for (int q = 0; q < 2; ++q) acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[q], acc[q][m], 0, 0, 0);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int pg = p0 + 32 * q + l32;
    if (pg >= P) continue;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
        if (co < g.co) out[((long)b * g.co + co) * P + pg] = acc[q][m][r] + (bias ? bias[co] : 0.f);
      }
  }
}

static int dcn_fused_rs(int K, int cop) {
  int rs = K * cop;            // a multiple of 32: the half-waves (channel cq, cq+1) must sit 32 banks apart
  if (rs % 64 == 0) rs += 32;
  return rs;
}

// The fused kernel covers the configurations the model uses (group 1, an even number of channels per deformable group,
// <= 64 output channels); everything else takes the im2col + GEMM path below.  RTP_DCN_UNFUSED=1 forces that path (A/B).
static bool dcn_fused_ok(const DcnGeom& g, int step) {
  const char* e = getenv("RTP_DCN_UNFUSED");
  const bool off = e && atoi(e);
  if (off || g.group != 1 || g.co > 64) return false;
  const int cpg = g.c / g.dg, K = g.kh * g.kw, cop = g.co > 32 ? 64 : 32;
  if (cpg % 2 || g.w < 2) return false;
  if ((long)g.c * g.h * g.w * 4 >= (1L << 31) - (1L << 20)) return false;
  if ((long)cpg * dcn_fused_rs(K, cop) * 4 > 64 * 1024) return false;
  if ((long)step * g.ho * g.wo < cop) return false;  // the transposed weights borrow the columns workspace
  return true;
}

// Window plan for the tile size of the kernel; nch = 0 when no even divisor of cpg fits the LDS budget.
static DcnWin dcn_win_plan(const DcnGeom& g, int cop) {
  const int K = g.kh * g.kw, cpg = g.c / g.dg;
  const char* e = getenv("RTP_DCN_HALO");
  DcnWin wn;
  wn.rs = dcn_fused_rs(K, cop);
  wn.halo = e ? atoi(e) : 2;
  if (wn.halo < 0) wn.halo = 0;
  const int rows = (DCN_FWD_THREADS + g.wo - 2) / g.wo + 1;  // output rows a tile of consecutive positions can touch
  wn.wr = (rows - 1) * g.sh + (g.kh - 1) * g.dh + 2 + 2 * wn.halo;
  wn.wrs = ((g.w & 3) == 0 && g.w + 4 <= 4 * DCN_FWD_THREADS) ? g.w + 4 : 0;
  wn.ws = wn.wrs ? wn.wr * wn.wrs + 4 : wn.wr * g.w;
  wn.ws = (wn.ws + 31) / 32 * 32;
  if (wn.ws % 64 == 0) wn.ws += 32;  // channel cq+1 (upper half-wave) 32 banks away from channel cq
  wn.nch = 0;
  const char* nw = getenv("RTP_DCN_NOWIN");
  if (nw && atoi(nw)) return wn;
  for (int budget : {78 * 1024, 156 * 1024}) {  // two blocks per CU, else one
    for (int n = cpg; n >= 2; --n) {
      if (cpg % n || n % 2) continue;
      if ((long)n * (wn.ws + wn.rs) * 4 <= budget) { wn.nch = n; return wn; }
    }
  }
  return wn;
}

static void dcn_forward_fused(const float* input, const float* weight, const float* bias, const float* offset,
                              const float* mask, float* output, float* ws, const DcnGeom& g, hipStream_t s) {
  const int P = g.ho * g.wo, K = g.kh * g.kw, cpg = g.c / g.dg;
  const int cop = g.co > 32 ? 64 : 32, rs = dcn_fused_rs(K, cop);
  hipLaunchKernelGGL(dcn_wt_kernel, dim3((g.c * K * cop + 255) / 256), dim3(256), 0, s, weight, ws, g.co, g.c, K, cop);
  const dim3 grid((P + DCN_FWD_THREADS - 1) / DCN_FWD_THREADS, g.n);
  const DcnWin wn = dcn_win_plan(g, cop);
  if (wn.nch) {
    const size_t lds = (size_t)wn.nch * (wn.ws + wn.rs) * sizeof(float);
    if (wn.wrs) {
      if (cop == 32)
        hipLaunchKernelGGL(dcn_winp_fwd_kernel<1>, grid, dim3(DCN_FWD_THREADS), lds, s, input, offset, mask, ws, bias, output, g, wn);
      else
        hipLaunchKernelGGL(dcn_winp_fwd_kernel<2>, grid, dim3(DCN_FWD_THREADS), lds, s, input, offset, mask, ws, bias, output, g, wn);
    } else if (cop == 32)
      hipLaunchKernelGGL(dcn_win_fwd_kernel<1>, grid, dim3(DCN_FWD_THREADS), lds, s, input, offset, mask, ws, bias, output, g, wn);
    else
      hipLaunchKernelGGL(dcn_win_fwd_kernel<2>, grid, dim3(DCN_FWD_THREADS), lds, s, input, offset, mask, ws, bias, output, g, wn);
    return;
  }
  const size_t lds = (size_t)cpg * rs * sizeof(float);
  if (cop == 32)
    hipLaunchKernelGGL(dcn_fused_fwd_kernel<1>, grid, dim3(DCN_FWD_THREADS), lds, s, input, offset, mask, ws, bias, output, g, rs);
  else
    hipLaunchKernelGGL(dcn_fused_fwd_kernel<2>, grid, dim3(DCN_FWD_THREADS), lds, s, input, offset, mask, ws, bias, output, g, rs);
}

static void dcn_im2col(hipStream_t s, const float* input, const float* offset, const float* mask, float* ws, const DcnGeom& g,
                       int b0, int step) {
  const int P = g.ho * g.wo;
  if (g.w >= 2 && (long)g.c * g.h * g.w * 4 < (1L << 31) - (1L << 20))
    hipLaunchKernelGGL(dcn_im2col_group_kernel, dim3(grid1d((long)step * g.dg * P)), dim3(256), 0, s, input, offset, mask, ws,
                       g, b0, step);
  else
    hipLaunchKernelGGL(dcn_im2col_kernel, dim3(grid1d((long)g.c * step * P)), dim3(256), 0, s, input, offset, mask, ws, g, b0,
                       step);
}

static int dcn_check(DcnGeom& g, int im2col_step) {
  if (g.n < 1 || g.c < 1 || g.co < 1 || g.kh < 1 || g.kw < 1 || g.sh < 1 || g.sw < 1 || g.dh < 1 || g.dw < 1)
    return RTP_ERR_SHAPE;  // shape_check, deform_conv_cuda.cpp:62-150
  if (g.group < 1 || g.dg < 1 || g.c % g.group || g.co % g.group || g.c % g.dg) return RTP_ERR_SHAPE;
  dcn_out_size(g);
  if (g.ho < 1 || g.wo < 1) return RTP_ERR_SHAPE;
  if (im2col_step < 1 || g.n % im2col_step) return RTP_ERR_SHAPE;  // 'im2col step must divide batchsize'
  return RTP_OK;
}

extern "C" long rtp_dcn_workspace_bytes(int n, int c, int h, int w, int co, int kh, int kw, int ho, int wo) {
  (void)h; (void)w; (void)co;
  return (long)c * kh * kw * n * ho * wo * (long)sizeof(float);  // columns for one chunk of up to n images
}

static int dcn_forward(const float* input, const float* weight, const float* bias, const float* offset,
                       const float* mask, float* output, float* ws, DcnGeom g, int step, hipStream_t s) {
  const int P = g.ho * g.wo, K = g.kh * g.kw;
  const int cg = g.c / g.group, cog = g.co / g.group;
  RtpProfScope prof(RTP_FAM_DCN, s);
  if (dcn_fused_ok(g, step)) {
    dcn_forward_fused(input, weight, bias, offset, mask, output, ws, g, s);
    RTP_CHECK_LAUNCH();
    return RTP_OK;
  }
  for (int b0 = 0; b0 < g.n; b0 += step) {
    dcn_im2col(s, input, offset, mask, ws, g, b0, step);
    for (int gi = 0; gi < g.group; ++gi) {
      // out[b0 + n/P][gi*cog + m][n%P] = W[gi][m][:] . columns[gi][:][n]
      sgemm(s, mv(weight + (long)gi * cog * cg * K, (long)cg * K, 1),
            mv(ws + (long)gi * cg * K * step * P, (long)step * P, 1),
            mv(output + ((long)b0 * g.co + gi * cog) * P, P, 1, P, (long)g.co * P), cog, step * P, cg * K, 1.f, 0.f);
    }
  }
  if (bias) {
    const long total = (long)g.n * g.co * P;
    hipLaunchKernelGGL(dcn_bias_kernel, dim3(grid1d(total)), dim3(256), 0, s, output, bias, g.co, P, total, 1);
  }
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

static bool dcn_bwd_fused_ok(const DcnGeom& g, int step);
static int dcn_backward_fused(const float* input, const float* offset, const float* gradOutput, float* gradInput,
                              float* gradOffset, const float* weight, float* gradWeight, float* ws, DcnGeom g, float scale,
                              hipStream_t s, int overwrite = 0);

static int dcn_backward_input(const float* input, const float* offset, const float* mask, const float* gradOutput,
                              float* gradInput, float* gradOffset, float* gradMask, const float* weight, float* ws,
                              DcnGeom g, int step, hipStream_t s) {
  const int P = g.ho * g.wo, K = g.kh * g.kw;
  const int cg = g.c / g.group, cog = g.co / g.group;
  if (!mask && !gradMask && dcn_bwd_fused_ok(g, step))   // one pass, no column gradient in HBM
    return dcn_backward_fused(input, offset, gradOutput, gradInput, gradOffset, weight, nullptr, ws, g, 1.f, s);
  RtpProfScope prof(RTP_FAM_DCN, s);
  for (int b0 = 0; b0 < g.n; b0 += step) {
    for (int gi = 0; gi < g.group; ++gi) {
      // column gradient of group gi = W[gi]^T (cg*K x cog) . gradOut chunk (cog x step*P), written channel-innermost
      // into the whole workspace (every group addresses its own channels through dcn_cg_index)
      if (!dcn_colgrad_mfma(s, weight + (long)gi * cog * cg * K, gradOutput + ((long)b0 * g.co + gi * cog) * P, ws, cog,
                            gi * cg, cg, g, (long)step * P))
        sgemm(s, mv(weight + (long)gi * cog * cg * K, 1, (long)cg * K),
              mv(gradOutput + ((long)b0 * g.co + gi * cog) * P, P, 1, P, (long)g.co * P), mv(ws, 0, 0), cg * K, step * P,
              cog, 1.f, 0.f, &g, gi * cg);
    }
    {
      const int cpg = g.c / g.dg;
      const size_t lds = (size_t)256 * (cpg + 1) * sizeof(float);
      const int use_lds = lds <= 64 * 1024;
      hipLaunchKernelGGL(dcn_col2im_coord_kernel, dim3(step * g.dg * K, (P + 255) / 256), dim3(256), use_lds ? lds : 0, s, ws,
                         input, offset, mask, gradOffset, gradMask, g, b0, step, use_lds);
    }
    const char* re = getenv("RTP_DCN_GATHER_R");
    const int R = re ? atoi(re) : 2;  // offsets up to R pixels take the atomic-free gather; 0: scatter everything
    if (R > 0 && R <= 8) {
      const int cpg = g.c / g.dg;
      const int ch = cpg % 16 == 0 ? 16 : cpg % 8 == 0 ? 8 : cpg % 4 == 0 ? 4 : cpg % 2 == 0 ? 2 : 1;
      const int tiles = ((g.h + DCN_GT_Y - 1) / DCN_GT_Y) * ((g.w + DCN_GT_X - 1) / DCN_GT_X);
      const int blocks = step * g.dg * (cpg / ch) * tiles;
      const int rh_max = (DCN_GT_Y - 1 + 2 * R) / g.sh + 2, rw_max = (DCN_GT_X - 1 + 2 * R) / g.sw + 2;
      const size_t lds = (size_t)(2 + ch) * rh_max * rw_max * sizeof(float);
#define DCN_GATHER(CH_)                                                                                              \
  hipLaunchKernelGGL(dcn_col2im_gather_kernel<CH_>, dim3(blocks), dim3(256), lds, s, ws, offset, mask, gradInput, g, b0, \
                     step, R, rh_max, rw_max)
      switch (ch) {
        case 16: DCN_GATHER(16); break;
        case 8: DCN_GATHER(8); break;
        case 4: DCN_GATHER(4); break;
        case 2: DCN_GATHER(2); break;
        default: DCN_GATHER(1); break;
      }
      hipLaunchKernelGGL(dcn_col2im_outlier_kernel, dim3(grid1d((long)step * g.dg * K * P)), dim3(256), 0, s, ws, offset, mask,
                         gradInput, g, b0, step, R);
    } else if ((size_t)g.h * g.w * sizeof(float) <= 60 * 1024)
      hipLaunchKernelGGL(dcn_col2im_plane_kernel, dim3(step * g.c), dim3(256), sizeof(float) * g.h * g.w, s, ws, offset, mask,
                         gradInput, g, b0, step);
    else
      hipLaunchKernelGGL(dcn_col2im_kernel, dim3(grid1d((long)g.c * K * step * P)), dim3(256), 0, s, ws, offset, mask,
                         gradInput, g, b0, step);
  }
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}


// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradient WITHOUT the column matrix (round 3): gradW[co][c*K + t] += scale * sum_p gO[co][p] * col[c*K + t][p], with the
// deformable samples produced on chip.  The contraction runs over POSITIONS, but a coalesced gather has its lanes on positions
// (64 consecutive output pixels of one channel plane hit two or three cache lines; lanes on (channel, tap) columns hit 64), so a
// wave -- one deformable group of 8 channels -- gathers its 72 columns x 64 positions the im2col way (sample set-up once per
// (position, tap), corner pairs as 8-byte buffer loads) into a wave-private LDS tile [72][64 + 4] and reads it back TRANSPOSED as
// the B operand of v_mfma_f32_32x32x2_f32 (lane = column, four consecutive positions per ds_read_b128; the row pad of 4 floats
// makes the 16 rows of a lane group hit 16 different 4-bank groups).  A = gO rows straight from global memory, lane-half h taking
// positions 32h .. 32h + 31 of its row as eight 16-byte loads (the summation order over positions is free).  3 column tiles x 16
// accumulator registers per wave, kept across all of the wave's tiles, added to gradW with fp32 atomics at the end (as
// dcn_gradw_mfma_kernel does).  No 755-MB columns write + read per 64-image chunk: im2col 0.45 + GEMM 0.38 ms -> one launch.
// Geometry: one group, 8 channels per deformable group, 3 x 3 taps, <= 32 output channels, Ho * Wo a multiple of 64.
#define DCN_GF_ROW 68   // floats per LDS row: 64 positions + 4 pad
__global__ __launch_bounds__(256, 2) void dcn_gradw_fused_kernel(const float* x, const float* offset, const float* mask, const float* go,
                                                                 float* gw, DcnGeom g, float scale, int tiles_per_block) {
  extern __shared__ __attribute__((aligned(16))) float gf_lds[];
  const int lane = threadIdx.x & 63, dgi = threadIdx.x >> 6;   // wave = deformable group (g.dg == 4 waves)
  const int half = lane >> 5, l32 = lane & 31;
  float* colL = gf_lds + dgi * (72 * DCN_GF_ROW);
  const int P = g.ho * g.wo, K = 9, HW = g.h * g.w;
  const long tiles = (long)g.n * (P / 64);
  dcn_f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const long t0 = (long)blockIdx.x * tiles_per_block;
  for (long tile = t0; tile < t0 + tiles_per_block && tile < tiles; ++tile) {
    const int b = (int)(tile / (P / 64)), p0 = (int)(tile - (long)b * (P / 64)) * 64;
    // ---- gather: lane = position p0 + lane, this wave's group: 9 taps x 8 channels -> colL[cc * 9 + t][lane]
    {
      const int p = p0 + lane;
      const int wo = p % g.wo, ho = p / g.wo;
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)b * g.c * HW), 0, g.c * HW * 4, 0x00020000);
      const float* off = offset + ((long)b * g.dg + dgi) * 2 * K * P + p;
      const float* mk = mask ? mask + ((long)b * g.dg + dgi) * K * P + p : nullptr;
      const int h_in = ho * g.sh - g.ph, w_in = wo * g.sw - g.pw;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int i = t / 3, j = t - i * 3;
        const float hi = h_in + i * g.dh + off[(long)(2 * t) * P];
        const float wi = w_in + j * g.dw + off[(long)(2 * t + 1) * P];
        const float m = mk ? mk[(long)t * P] : 1.f;
        const bool in = hi > -1.f && wi > -1.f && hi < g.h && wi < g.w;
        const float hf = floorf(hi), wf = floorf(wi);
        const int h_low = (int)hf, w_low = (int)wf;
        const float lh = hi - hf, lw = wi - wf, hh = 1.f - lh, hw = 1.f - lw;
        const bool c0 = w_low >= 0, c1 = w_low + 1 <= g.w - 1;
        const float w1 = c0 ? hh * hw : 0.f, w2 = c1 ? hh * lw : 0.f, w3 = c0 ? lh * hw : 0.f, w4 = c1 ? lh * lw : 0.f;
        const int xs = w_low < 0 ? 0 : (c1 ? w_low : g.w - 2);
        const int base = (h_low * g.w + xs) * 4;
        const int a0 = (in && h_low >= 0) ? base : DCN_OOB;
        const int a1 = (in && h_low + 1 <= g.h - 1) ? base + g.w * 4 : DCN_OOB;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const int coff = (dgi * 8 + cc) * HW * 4;
          const dcn_u32x2 q0 = __builtin_amdgcn_raw_buffer_load_b64(rx, a0 + coff, 0, 0);
          const dcn_u32x2 q1 = __builtin_amdgcn_raw_buffer_load_b64(rx, a1 + coff, 0, 0);
          const float v1 = __uint_as_float(c1 ? q0.x : q0.y), v2 = __uint_as_float(c0 ? q0.y : q0.x);
          const float v3 = __uint_as_float(c1 ? q1.x : q1.y), v4 = __uint_as_float(c0 ? q1.y : q1.x);
          float v = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
          if (mk) v *= m;
          colL[(cc * 9 + t) * DCN_GF_ROW + lane] = v;
        }
      }
    }
    // ---- gO rows (A operand): lane (co = l32, half) holds positions 32 half .. 32 half + 31 of its row
    float4 a4[8];
    {
      const float* src = go + ((long)b * g.co + l32) * P + p0 + 32 * half;
#pragma unroll
      for (int j = 0; j < 8; ++j) a4[j] = l32 < g.co ? *(const float4*)(src + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // (the tile is this wave's own: the LDS pipe keeps a wave's writes and reads in order, no barrier)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int row = t * 32 + l32;
      const float* src = colL + (row < 72 ? row : 0) * DCN_GF_ROW + 32 * half;
      float4 b4[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        b4[j] = *(const float4*)(src + 4 * j);
        if (row >= 72) b4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].x, b4[j].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].y, b4[j].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].z, b4[j].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].w, b4[j].w, acc[t], 0, 0, 0);
      }
    }
  }
  // D[m = co][n = column]: lane l, register i -> n = l % 32, m = 8 * (i / 4) + 4 * (l / 32) + i % 4; column -> gradW column
  // (dgi * 8 + cc) * 9 + t = dgi * 72 + (cc * 9 + t)
  const int ck_total = g.c * K;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int r = t * 32 + l32;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = 8 * (i >> 2) + 4 * half + (i & 3);
      if (co < g.co && r < 72) atomicAdd(gw + (long)co * ck_total + dgi * 72 + r, scale * acc[t][i]);
    }
  }
}

static bool dcn_gradw_fused_ok(const DcnGeom& g) {
  return g.group == 1 && g.dg == 4 && g.c == 32 && g.kh == 3 && g.kw == 3 && g.co <= 32 && (g.ho * g.wo) % 64 == 0 && g.w >= 2;
}

static int dcn_gradw_fused_launch(const float* input, const float* offset, const float* mask, const float* gradOutput,
                                   float* gradWeight, const DcnGeom& g, float scale, hipStream_t s) {
  const long tiles = (long)g.n * (g.ho * g.wo / 64);
  int tpb = (int)((tiles + 511) / 512);   // two 4-wave blocks per CU
  if (tpb < 1) tpb = 1;
  const size_t lds = sizeof(float) * 4 * 72 * DCN_GF_ROW;
  static bool attr[RTP_MAX_DEVICES] = {};
  if (rtp_once_per_device(attr)) {
    if (hipFuncSetAttribute((const void*)dcn_gradw_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      attr[rtp_device_index()] = false;
      return RTP_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL(dcn_gradw_fused_kernel, dim3((unsigned)((tiles + tpb - 1) / tpb)), dim3(256), lds, s, input, offset, mask,
                     gradOutput, gradWeight, g, scale, tpb);
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// The WHOLE backward in one pass over the output rows (round 3): column gradient, coordinate gradient, input gradient and
// (WITH_GW) weight gradient without any column matrix in HBM.  3 x 3 taps, stride 1, dilation 1, padding 1 (Ho = H, Wo = W),
// 8 channels per deformable group, <= 32 output channels, no mask.
//
// A wave owns one deformable group of a strip of 64 consecutive output columns of one image and marches down its rows; lane =
// column.  Per row:
//   (1) column gradient G[(c, tap)][column] = W^T . gO for the wave's 72 column rows x 64 positions on v_mfma_f32_32x32x2_f32
//       (A = the group's weights, held in registers for the whole march; B = gO of the row, lane = position) -> wave-private LDS
//       tile [72][64 + 4], so that each lane reads its own position's values back;
//   (2) per tap: the sample is set up once (as dcn_im2col_group_kernel does) and each of the 8 channels gathers its corner pairs
//       ONCE for three consumers: the column value (written over G in the same LDS slot: the B operand of the weight gradient),
//       the coordinate gradient (bottom - top and right - left differences of the same corners) and -- no gather, no atomics --
//   (3) the INPUT gradient: a sample whose offsets lie in [-2, 2) lands in a 5 x 5 neighbourhood of its undeformed tap position,
//       so its four bilinear weights are two separable 5-vectors (two non-zeros each) and the scatter becomes 25 FMAs into a
//       STATICALLY indexed register patch; the three taps of a tap row share a 5 x 7 patch, which is reduced across lanes with six
//       DPP wave shifts per row (lane x receives what lanes x-6 .. x contributed to output column x0 + x - 3) and added to a ring of
//       7 output-row accumulators per channel; the ring's oldest row is complete after every position row and is stored (each
//       grad_input element is written by exactly one wave: bit-reproducible).  Strips advance by 56 columns (7 lanes of overlap
//       recompute the neighbours' column gradients, nothing else); row segments carry 3 halo rows either side for the same reason;
//   (4) weight gradient: LDS tile read back transposed (lane = column row), A = gO rows as 16-byte loads, accumulators kept for
//       the whole march, fp32 atomics at the end.
// Samples with an offset component outside [-2, 2) -- rare at any sane offset scale -- are left to dcn_bwd_outlier_rows_kernel:
// the wave appends (image, group, row, strip) to a list in the workspace, and that kernel recomputes the few column gradients
// it needs (32 MACs per channel) and scatters them with global atomics after this kernel has finished.  Replaces, per 64-image
// chunk at [.,32,64,160]: colgrad 0.23 + coord 0.45 + gather 0.78 + outliers 0.09 + im2col 0.45 + weight GEMM 0.38 ms and
// 3 x 755 MB of column traffic.
#define DCN_FB_ROW 68   // floats per LDS row
#define DCN_FB_XO 56    // output columns / owned positions per strip
#define DCN_FB_PF 2      // corner pairs are requested this many channels ahead (3: register spills, no gain)
typedef __bf16 dcn_bf16x8 __attribute__((ext_vector_type(8)));
// x = hi + lo + O(2^-17 x) with hi, lo in bf16: an fp32 product on the bf16 matrix core as hi*hi + hi*lo + lo*hi (the dropped
// lo*lo term is 2^-16 relative; tools/ubench/mfma_bf16x3: 3.9e-6 norm-wise at K = 32, a third of the fp32 MFMA's cycles)
__device__ __forceinline__ void dcn_split8(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7,
                                           dcn_bf16x8& hi, dcn_bf16x8& lo) {
  const float x[8] = {x0, x1, x2, x3, x4, x5, x6, x7};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)x[e];
    hi[e] = h;
    lo[e] = (__bf16)(x[e] - (float)h);
  }
}
#define DCN_MFMA_X3(AH_, AL_, BH_, BL_, D_)                                    \
  D_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH_, BH_, D_, 0, 0, 0);         \
  D_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH_, BL_, D_, 0, 0, 0);         \
  D_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL_, BH_, D_, 0, 0, 0);
__device__ __forceinline__ float dcn_wave_shr1(float v) {   // lane x <- lane x - 1, lane 0 <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}

// X3: the two matrix products on v_mfma_f32_32x32x16_bf16 with (hi, lo) splits instead of v_mfma_f32_32x32x2_f32
template <bool WITH_GW, bool X3>   // one block per CU: 512 registers per lane (two waves per SIMD at 256 spilled and ran 2x slower)
__global__ __launch_bounds__(256, 1) void dcn_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                               const float* __restrict__ go, const float* __restrict__ w,
                                                               float* gin, float* goff, float* gw, unsigned* rec, DcnGeom g,
                                                               float scale, int strips, int segs, int seg_rows, int overwrite) {
  extern __shared__ __attribute__((aligned(16))) float fb_lds[];
  const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  const int dgi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: the buffer resources below depend on it
  float* T = fb_lds + dgi * (72 * DCN_FB_ROW);
  const int H = g.h, W = g.w, HW = H * W;
  int bid = blockIdx.x;
  const int k = bid % strips; bid /= strips;
  const int sg = bid % segs;
  const int b = bid / segs;
  const int x0 = DCN_FB_XO * k - 4;
  const int wo = x0 + lane;
  const bool pos_x = wo >= 0 && wo < W;
  const bool own_x = lane >= 4 && lane < 4 + DCN_FB_XO && wo < W;
  const int oc = wo - 3;
  const bool st_x = lane >= 7 && lane < 7 + DCN_FB_XO && oc < W;
  const int r0 = sg * seg_rows, r1 = min(H, r0 + seg_rows);
  const int hs = max(r0 - 3, 0), he = min(r1 + 3, H);
  // every global stream goes through a buffer resource with 32-bit offsets (per-lane part + uniform part): 64-bit pointers per
  // stream would cost two registers each for ~80 streams; a lane that must not load / store passes an out-of-range offset
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)b * g.c * HW), 0, g.c * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rgo = __builtin_amdgcn_make_buffer_rsrc((void*)(go + (long)b * g.co * HW), 0, g.co * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rof =
      __builtin_amdgcn_make_buffer_rsrc((void*)(offset + ((long)(b * 4 + dgi) * 18) * HW), 0, 18 * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rgf =
      __builtin_amdgcn_make_buffer_rsrc((void*)(goff + ((long)(b * 4 + dgi) * 18) * HW), 0, 18 * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rgi =
      __builtin_amdgcn_make_buffer_rsrc((void*)(gin + ((long)b * g.c + dgi * 8) * HW), 0, 8 * HW * 4, 0x00020000);

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, g.co * g.c * 9 * 4, 0x00020000);
  // ring of 7 output-row accumulators per channel, in LDS behind the tiles ([channel][slot][lane], slot = output row mod 7): in
  // registers it cost 56 VGPRs and 48 moves per row to rotate
  float* R = fb_lds + 4 * (72 * DCN_FB_ROW) + dgi * (8 * 7 * 64) + lane;
#pragma unroll
  for (int q = 0; q < 56; ++q) R[q * 64] = 0.f;
  dcn_f32x16 gacc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) gacc[t][r] = 0.f;

  // Software pipeline of the march (one wave per SIMD: nothing else hides latency; a wave cannot run its own MFMAs under its own
  // vector instructions either -- tools/ubench/issue_rates: 2 MFMA + 32 FMA interleaved take the sum of both -- so the column
  // gradient is simply computed at the start of the row).  Column rows are ordered tap row first (row 24 i + 3 cc + j for tap
  // (i, j) of channel cc), so a tap row is one MFMA tile (24 of 32 rows used).  gO operands and offsets are loaded one row ahead
  // into the registers their predecessors leave, a tap row's addresses and first corner pairs during the previous tap row, corner
  // pairs two channels ahead, the old grad_input row and the weight gradient's gO rows before the taps; the weights stay in
  // registers.
  // operand element idx of a lane-half covers output channel co = 2 idx + half (fp32 MFMA: k = half) or 16 (idx / 8) + 8 half +
  // idx % 8 (bf16 MFMA, k = 8 half + idx % 8 of k step idx / 8)
#define DCN_FB_CO_STEP(IDX_) ((X3 ? 16 * ((IDX_) / 8) + (IDX_) % 8 : 2 * (IDX_)))
  const int co_half = half * (X3 ? 8 : 1);
  float wa[3][16];    // A operand of tile i: W[co][group column of local row l32]
  dcn_bf16x8 waH[3][2], waL[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int vw = l32 < 24 ? (co_half * (g.c * 9) + dgi * 72 + (l32 / 3) * 9 + 3 * i + l32 % 3) * 4 : DCN_OOB;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      wa[i][kk] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, vw + DCN_FB_CO_STEP(kk) * (g.c * 9 * 4), 0, 0));
    if (X3) {
#pragma unroll
      for (int st = 0; st < 2; ++st)
        dcn_split8(wa[i][8 * st], wa[i][8 * st + 1], wa[i][8 * st + 2], wa[i][8 * st + 3], wa[i][8 * st + 4], wa[i][8 * st + 5],
                   wa[i][8 * st + 6], wa[i][8 * st + 7], waH[i][st], waL[i][st]);
    }
  }
  float bvA[2][16];   // B operand gO[co][position 32 nt + l32] of the row
  float ofs[18];                  // the position's 18 offsets of the row
#define DCN_FB_LOAD_BV(DST_, ROW_)                                                                                             \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                                           \
    const int won = x0 + 32 * nt + l32;                                                                                        \
    const int vo = (won >= 0 && won < W && (ROW_) < he) ? (co_half * HW + (ROW_) * W + won) * 4 : DCN_OOB;                     \
    _Pragma("unroll") for (int kk = 0; kk < 16; ++kk)                                                                          \
        DST_[nt][kk] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rgo, vo + DCN_FB_CO_STEP(kk) * (HW * 4), 0, 0));  \
  }
  DCN_FB_LOAD_BV(bvA, hs)
  {
    const int vof = pos_x ? (hs * W + wo) * 4 : DCN_OOB;
#pragma unroll
    for (int q = 0; q < 18; ++q) ofs[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rof, vof, q * HW * 4, 0));
  }
  // Gather addresses of a tap row and the corner pairs of its first two channels, requested while the PREVIOUS tap row's last
  // channels are in work (its offsets are in the registers already): a tap row used to start by waiting for its first gathers
  int na0[3], na1[3];
  dcn_u32x2 nqa[DCN_FB_PF][3], nqb[DCN_FB_PF][3];
#define DCN_FB_NEXT_TAP_ROW(I_, HO_, OWN_)                                                                                     \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                                              \
    const float hi = (float)((HO_) - 1 + (I_)) + ofs[2 * (3 * (I_) + j)], wi = (float)(wo - 1 + j) + ofs[2 * (3 * (I_) + j) + 1]; \
    const bool in = hi > -1.f && wi > -1.f && hi < (float)H && wi < (float)W;                                                  \
    const int h_low = (int)floorf(hi), w_low = (int)floorf(wi);                                                                \
    const int xs = w_low < 0 ? 0 : (w_low + 1 <= W - 1 ? w_low : W - 2);                                                       \
    const int base = (h_low * W + xs) * 4;                                                                                     \
    const bool act = (OWN_) && in;                                                                                             \
    na0[j] = (act && h_low >= 0) ? base : DCN_OOB;                                                                             \
    na1[j] = (act && h_low + 1 <= H - 1) ? base + W * 4 : DCN_OOB;                                                             \
    _Pragma("unroll") for (int ch = 0; ch < DCN_FB_PF; ++ch) {                                                                 \
      nqa[ch][j] = __builtin_amdgcn_raw_buffer_load_b64(rx, na0[j] + (dgi * 8 + ch) * HW * 4, 0, 0);                           \
      nqb[ch][j] = __builtin_amdgcn_raw_buffer_load_b64(rx, na1[j] + (dgi * 8 + ch) * HW * 4, 0, 0);                           \
    }                                                                                                                          \
  }
  DCN_FB_NEXT_TAP_ROW(0, hs, (hs >= r0 && hs < r1 && own_x))

  for (int ho = hs; ho < he; ++ho) {
    const bool own_row = ho >= r0 && ho < r1;   // wave-uniform: halo rows only feed the input-gradient ring
    const bool own_next = ho + 1 >= r0 && ho + 1 < r1 && ho + 1 < he && own_x;
    // gO rows of the weight gradient (A operand, lane = output channel): requested now, used after the taps
    dcn_f32x4 a4[8];
    if (WITH_GW) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // positions 32 half + 4 j .. (fp32 MFMA: k = half, any order of k steps) or 16 (j / 2) + 8 half + 4 (j % 2) .. (bf16 MFMA:
        // k = 8 half + e of k step j / 2); 4-aligned group, W % 4 == 0: wholly inside or wholly outside the row
        const int w4 = x0 + (X3 ? 16 * (j >> 1) + 8 * half + 4 * (j & 1) : 32 * half + 4 * j);
        const int va = (own_row && w4 >= 0 && w4 < W) ? (l32 * HW + ho * W + w4) * 4 : DCN_OOB;
        a4[j] = __builtin_bit_cast(dcn_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rgo, va, 0, 0));
      }
    }
    // ---- (1) the row's column gradient, tile i -> T[24 i + (cc, j)][lane]; D[m][n]: lane l, register r -> n = l % 32,
    // m = 8 (r / 4) + 4 (l / 32) + r % 4 (rows 24 .. 31 do not exist)
    dcn_bf16x8 bH[2][2], bL[2][2];
    if (X3) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int st = 0; st < 2; ++st)
          dcn_split8(bvA[nt][8 * st], bvA[nt][8 * st + 1], bvA[nt][8 * st + 2], bvA[nt][8 * st + 3], bvA[nt][8 * st + 4],
                     bvA[nt][8 * st + 5], bvA[nt][8 * st + 6], bvA[nt][8 * st + 7], bH[nt][st], bL[nt][st]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      dcn_f32x16 d[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[nt][r] = 0.f;
      if (X3) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) { DCN_MFMA_X3(waH[i][st], waL[i][st], bH[nt][st], bL[nt][st], d[nt]) }
      } else {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[i][kk], bvA[nt][kk], d[nt], 0, 0, 0);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 12; ++r) T[(24 * i + 8 * (r >> 2) + 4 * half + (r & 3)) * DCN_FB_ROW + 32 * nt + l32] = d[nt][r];
    }
    __builtin_amdgcn_sched_barrier(0);
    DCN_FB_LOAD_BV(bvA, ho + 1)   // consumed: the registers take the next row's
    int rs[7];   // ring slot (x 64 floats) of output row ho - 3 + q
#pragma unroll
    for (int q = 0; q < 7; ++q) rs[q] = ((ho + 4 + q) % 7) * 64;
    // the grad_input row this position row completes (accumulate contract: read-modify-write), fetched now, stored after the taps
    const int orow = ho - 3;
    const int vg = (orow >= r0 && orow < r1 && st_x) ? (orow * W + oc) * 4 : DCN_OOB;
    float gold[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc)   // overwrite mode: nothing to add to (an out-of-range offset reads 0 without touching memory)
      gold[cc] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rgi, overwrite ? DCN_OOB : vg, cc * HW * 4, 0));
    const bool own = own_row && own_x;
    const int vof1 = (pos_x && ho + 1 < he) ? ((ho + 1) * W + wo) * 4 : DCN_OOB, vgf = own ? (ho * W + wo) * 4 : DCN_OOB;
    bool outl = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      // ---- (2) + (3): the tap row
      int a0[3], a1[3];
      float lh[3], wh[3][5], ww[3][5], gh[3], gwc[3];
      dcn_f32x2 wxy[3], exy[3];   // what the loaded column pair contributes to the row's sample / to d sample / d w
      bool wide = false;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int t = 3 * i + j;
        const float oh = ofs[2 * t], ow = ofs[2 * t + 1];
        const int nh = ho - 1 + i, nw = wo - 1 + j;
        const float hi = (float)nh + oh, wi = (float)nw + ow;
        const bool in = hi > -1.f && wi > -1.f && hi < (float)H && wi < (float)W;
        const float hf = floorf(hi), wf = floorf(wi);
        const int h_low = (int)hf, w_low = (int)wf;
        lh[j] = hi - hf;
        const float lw = wi - wf;
        const float hh = 1.f - lh[j], hw = 1.f - lw;
        const int ia = h_low - nh + 2, ib = w_low - nw + 2;   // patch row / column of the top-left corner
        const bool inpatch = pos_x && (unsigned)ia < 4u && (unsigned)ib < 4u;
        outl = outl || (own && in && !inpatch);
        wide = wide || (inpatch && ((unsigned)(ia - 1) > 1u || (unsigned)(ib - 1) > 1u));   // leaves the inner 3 x 3 footprint
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          wh[j][q] = inpatch ? ((q == ia ? hh : 0.f) + (q == ia + 1 ? lh[j] : 0.f)) : 0.f;
          ww[j][q] = (q == ib ? hw : 0.f) + (q == ib + 1 ? lw : 0.f);
        }
        // the 8-byte load covers columns (xs, xs + 1), clamped into the row: at the left / right edge one of the two is the live
        // corner and the other one does not exist; per-tap weights on the pair replace per-channel selects
        const bool c0 = w_low >= 0, c1 = w_low + 1 <= W - 1;
        const int xs = w_low < 0 ? 0 : (c1 ? w_low : W - 2);
        wxy[j] = c0 ? (c1 ? dcn_f32x2{hw, lw} : dcn_f32x2{0.f, hw}) : dcn_f32x2{lw, 0.f};
        exy[j] = c0 ? (c1 ? dcn_f32x2{-1.f, 1.f} : dcn_f32x2{0.f, -1.f}) : dcn_f32x2{1.f, 0.f};
        a0[j] = na0[j]; a1[j] = na1[j];   // computed (and the first two channels requested) during the previous tap row
        gh[j] = 0.f; gwc[j] = 0.f;
      }
      // corner pairs of channels cc + 1 .. cc + DCN_FB_PF are in flight while channel cc is consumed (halo rows / lanes that do not own
      // their position pass out-of-range addresses: zeros, no memory access)
      dcn_u32x2 qa[DCN_FB_PF + 1][3], qb[DCN_FB_PF + 1][3];   // [stage = channel % (DCN_FB_PF + 1)][tap]
#define DCN_FB_GATHER(CH_)                                                                                                     \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                                              \
    qa[(CH_) % (DCN_FB_PF + 1)][j] = __builtin_amdgcn_raw_buffer_load_b64(rx, a0[j] + (dgi * 8 + (CH_)) * HW * 4, 0, 0);      \
    qb[(CH_) % (DCN_FB_PF + 1)][j] = __builtin_amdgcn_raw_buffer_load_b64(rx, a1[j] + (dgi * 8 + (CH_)) * HW * 4, 0, 0);      \
  }
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int ch = 0; ch < DCN_FB_PF; ++ch) { qa[ch][j] = nqa[ch][j]; qb[ch][j] = nqb[ch][j]; }
      // this tap row's offsets are consumed: their registers take the next row's
#pragma unroll
      for (int q = 6 * i; q < 6 * i + 6; ++q) ofs[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rof, vof1, q * HW * 4, 0));
      // what every tap of every channel does whatever the patch: column gradient value, the gathered pair -> sample, coordinate
      // gradient
#define DCN_FB_TAP(J_)                                                                                                         \
  const int m = 24 * i + 3 * cc + (J_);                                                                                        \
  const float gv = T[m * DCN_FB_ROW + lane];                                                                                   \
  {                                                                                                                            \
    const dcn_u32x2 q0 = qa[cc % (DCN_FB_PF + 1)][J_], q1 = qb[cc % (DCN_FB_PF + 1)][J_];                                      \
    const dcn_f32x2 px = {__uint_as_float(q0.x), __uint_as_float(q1.x)}, py = {__uint_as_float(q0.y), __uint_as_float(q1.y)}; \
    const dcn_f32x2 tb = wxy[J_].x * px + wxy[J_].y * py; /* (top, bottom) row samples */                                      \
    const dcn_f32x2 dd = exy[J_].x * px + exy[J_].y * py; /* their d / d w */                                                  \
    const float dv = tb.y - tb.x;                         /* d sample / d h */                                                 \
    if (WITH_GW) T[m * DCN_FB_ROW + lane] = tb.x + lh[J_] * dv; /* the sample (0 for positions this wave does not own) */      \
    gh[J_] += gv * dv;                                                                                                         \
    gwc[J_] += gv * (dd.x + lh[J_] * (dd.y - dd.x));                                                                           \
  }
      // pin the order of a channel iteration: without this the compiler sinks all 48 gathers of the tap row below the patch
      // arithmetic (and spills the 24 column-gradient values they are multiplied with)
#define DCN_FB_PIN()                                                                                                           \
  asm volatile("" : "+v"(gh[0]), "+v"(gh[1]), "+v"(gh[2]), "+v"(gwc[0]), "+v"(gwc[1]), "+v"(gwc[2]));                          \
  __builtin_amdgcn_sched_barrier(0);
      if (__builtin_amdgcn_ballot_w64(wide) == 0) {
        // every sample of the tap row within one pixel of its undeformed position (wave-uniform; always so while the offset
        // conv is near its zero initialisation): 3 x 3 footprints, a 3 x 5 patch (rows 1 .. 3, columns 1 .. 5 of the general
        // one, as pairs (1,2) (3,4) 5), five shifts per row
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          if (cc < 8 - DCN_FB_PF) DCN_FB_GATHER(cc + DCN_FB_PF)
          if (cc == 8 - DCN_FB_PF) { if (i < 2) { DCN_FB_NEXT_TAP_ROW(i + 1, ho, own) } else { DCN_FB_NEXT_TAP_ROW(0, ho + 1, own_next) } }
          float rv[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) rv[a] = R[cc * 448 + rs[i + 1 + a]];
          dcn_f32x2 N12[3], N34[3];
          float N5[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) { N12[a] = dcn_f32x2{0.f, 0.f}; N34[a] = dcn_f32x2{0.f, 0.f}; N5[a] = 0.f; }
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            DCN_FB_TAP(j)
            const float s1 = ww[j][1] * gv, s2 = ww[j][2] * gv, s3 = ww[j][3] * gv;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              const float h = wh[j][a + 1];
              if (j == 0) { N12[a] += h * dcn_f32x2{s1, s2}; N34[a].x += h * s3; }
              if (j == 1) { N12[a].y += h * s1; N34[a] += h * dcn_f32x2{s2, s3}; }
              if (j == 2) { N34[a] += h * dcn_f32x2{s1, s2}; N5[a] += h * s3; }
            }
          }
          float acc[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) acc[a] = N5[a];
#define DCN_FB_SHIFT_ADD3(SRC_)                                                                                                \
  _Pragma("unroll") for (int a = 0; a < 3; ++a) acc[a] = dcn_wave_shr1(acc[a]) + SRC_;                                         \
  asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
          DCN_FB_SHIFT_ADD3(N34[a].y)
          DCN_FB_SHIFT_ADD3(N34[a].x)
          DCN_FB_SHIFT_ADD3(N12[a].y)
          DCN_FB_SHIFT_ADD3(N12[a].x)
          DCN_FB_SHIFT_ADD3(0.f)   // column 0 of the general patch is empty
#pragma unroll
          for (int a = 0; a < 3; ++a) R[cc * 448 + rs[i + 1 + a]] = rv[a] + acc[a];
          DCN_FB_PIN()
        }
      } else {
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          if (cc < 8 - DCN_FB_PF) DCN_FB_GATHER(cc + DCN_FB_PF)
          if (cc == 8 - DCN_FB_PF) { if (i < 2) { DCN_FB_NEXT_TAP_ROW(i + 1, ho, own) } else { DCN_FB_NEXT_TAP_ROW(0, ho + 1, own_next) } }
          float rv[5];
#pragma unroll
          for (int a = 0; a < 5; ++a) rv[a] = R[cc * 448 + rs[i + a]];
          // 5 x 7 patch of the tap row as packed pairs (v_pk_fma_f32): columns (0,1) (2,3) (4,5) 6
          dcn_f32x2 L01[5], L23[5], L45[5];
          float L6[5];
#pragma unroll
          for (int a = 0; a < 5; ++a) { L01[a] = dcn_f32x2{0.f, 0.f}; L23[a] = dcn_f32x2{0.f, 0.f}; L45[a] = dcn_f32x2{0.f, 0.f}; L6[a] = 0.f; }
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            DCN_FB_TAP(j)
            const float s0 = ww[j][0] * gv, s1 = ww[j][1] * gv, s2 = ww[j][2] * gv, s3 = ww[j][3] * gv, s4 = ww[j][4] * gv;
#pragma unroll
            for (int a = 0; a < 5; ++a) {
              const float h = wh[j][a];
              if (j == 0) { L01[a] += h * dcn_f32x2{s0, s1}; L23[a] += h * dcn_f32x2{s2, s3}; L45[a].x += h * s4; }
              if (j == 1) { L01[a].y += h * s0; L23[a] += h * dcn_f32x2{s1, s2}; L45[a] += h * dcn_f32x2{s3, s4}; }
              if (j == 2) { L23[a] += h * dcn_f32x2{s0, s1}; L45[a] += h * dcn_f32x2{s2, s3}; L6[a] += h * s4; }
            }
          }
          // x reduction, the five patch rows' chains interleaved: a DPP read of a register the previous instruction wrote costs
          // two wait states (the chain-by-chain order stalled 8 cycles per shift)
          float acc[5];
#pragma unroll
          for (int a = 0; a < 5; ++a) acc[a] = L6[a];
#define DCN_FB_SHIFT_ADD(SRC_)                                                                                                 \
  _Pragma("unroll") for (int a = 0; a < 5; ++a) acc[a] = dcn_wave_shr1(acc[a]) + SRC_;                                         \
  asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]));
          DCN_FB_SHIFT_ADD(L45[a].y)
          DCN_FB_SHIFT_ADD(L45[a].x)
          DCN_FB_SHIFT_ADD(L23[a].y)
          DCN_FB_SHIFT_ADD(L23[a].x)
          DCN_FB_SHIFT_ADD(L01[a].y)
          DCN_FB_SHIFT_ADD(L01[a].x)
#pragma unroll
          for (int a = 0; a < 5; ++a) R[cc * 448 + rs[i + a]] = rv[a] + acc[a];
          DCN_FB_PIN()
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gh[j]), rgf, vgf, (2 * (3 * i + j)) * HW * 4, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gwc[j]), rgf, vgf, (2 * (3 * i + j) + 1) * HW * 4, 0);
      }
    }
    if (own_row) {
      if (__builtin_amdgcn_ballot_w64(outl) != 0 && lane == 0) {
        const unsigned idx = atomicAdd(rec, 1u);
        rec[2 + 2 * idx] = (unsigned)b;
        rec[3 + 2 * idx] = ((unsigned)ho << 16) | ((unsigned)k << 8) | (unsigned)dgi;
      }
      // ---- (4) weight gradient of the row: gacc[t][co][local column row] += gO[co][pos] * sample[24 t + local row][pos]
      if (WITH_GW && X3) {
        dcn_bf16x8 aH[4], aL[4];
#pragma unroll
        for (int st = 0; st < 4; ++st)
          dcn_split8(a4[2 * st].x, a4[2 * st].y, a4[2 * st].z, a4[2 * st].w, a4[2 * st + 1].x, a4[2 * st + 1].y, a4[2 * st + 1].z,
                     a4[2 * st + 1].w, aH[st], aL[st]);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const float* bs = T + (24 * t + (l32 < 24 ? l32 : 0)) * DCN_FB_ROW + 8 * half;
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            float4 b0 = *(const float4*)(bs + 16 * st), b1 = *(const float4*)(bs + 16 * st + 4);
            if (l32 >= 24) { b0 = make_float4(0.f, 0.f, 0.f, 0.f); b1 = b0; }
            dcn_bf16x8 sH, sL;
            dcn_split8(b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, sH, sL);
            DCN_MFMA_X3(aH[st], aL[st], sH, sL, gacc[t])
          }
        }
      }
      if (WITH_GW && !X3) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const float* bs = T + (24 * t + (l32 < 24 ? l32 : 0)) * DCN_FB_ROW + 32 * half;
          float4 b4[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            b4[j] = *(const float4*)(bs + 4 * j);
            if (l32 >= 24) b4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            gacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].x, b4[j].x, gacc[t], 0, 0, 0);
            gacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].y, b4[j].y, gacc[t], 0, 0, 0);
            gacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].z, b4[j].z, gacc[t], 0, 0, 0);
            gacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].w, b4[j].w, gacc[t], 0, 0, 0);
          }
        }
      }
    }
    // ---- output row ho - 3 is complete
    {
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gold[cc] + R[cc * 448 + rs[0]]), rgi, vg, cc * HW * 4, 0);
        R[cc * 448 + rs[0]] = 0.f;   // the slot is output row ho + 4 from the next position row on
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 3; ++e) {   // the rows below the last position row (bottom of the image)
    const int orow = he - 3 + e;
    if (orow >= r0 && orow < r1) {
      const int vg = st_x ? (orow * W + oc) * 4 : DCN_OOB;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const float old = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rgi, overwrite ? DCN_OOB : vg, cc * HW * 4, 0));
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(old + R[cc * 448 + ((orow + 7) % 7) * 64]), rgi, vg, cc * HW * 4, 0);
      }
    }
  }
  if (WITH_GW) {
    const int ck_total = g.c * 9;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int col = dgi * 72 + (l32 / 3) * 9 + 3 * t + l32 % 3;   // local row l32 = 3 cc + j of tap row t
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = 8 * (i >> 2) + 4 * half + (i & 3);
        if (co < g.co && l32 < 24) atomicAdd(gw + (long)co * ck_total + col, scale * gacc[t][i]);
      }
    }
  }
}

// The rows dcn_bwd_fused_kernel listed: samples with an offset component outside [-2, 2).  One wave per listed (image, group,
// row, strip); the column gradient of such a sample is recomputed from W and gO (co MACs per channel) and scattered to its live
// corners with fp32 atomics, as dcn_col2im_outlier_kernel does.
__global__ __launch_bounds__(64) void dcn_bwd_outlier_rows_kernel(const float* offset, const float* go, const float* w,
                                                                  float* gin, const unsigned* rec, DcnGeom g) {
  const unsigned count = rec[0];
  const int lane = threadIdx.x, H = g.h, W = g.w, HW = H * W;
  const int cc = lane >> 3, part = lane & 7;   // while a sample is served: lane = (channel of the group, eighth of the co sum)
  for (unsigned r = blockIdx.x; r < count; r += gridDim.x) {
    const int b = (int)rec[2 + 2 * r];
    const unsigned v = rec[3 + 2 * r];
    const int ho = (int)(v >> 16), k = (int)((v >> 8) & 255u), dgi = (int)(v & 255u);
    const int wo = DCN_FB_XO * k - 4 + lane;
    const bool own = lane >= 4 && lane < 4 + DCN_FB_XO && wo < W;
    const float* offp = offset + ((long)(b * g.dg + dgi) * 18) * HW + (long)ho * W + wo;
    const float* gop = go + ((long)b * g.co * H + ho) * W;
    float* gim = gin + ((long)b * g.c + dgi * 8 + cc) * HW;
    for (int t = 0; t < 9; ++t) {
      const int i = t / 3, j = t - 3 * i;
      const float oh = own ? offp[(long)(2 * t) * HW] : 0.f, ow = own ? offp[(long)(2 * t + 1) * HW] : 0.f;
      const int nh = ho - 1 + i, nw = wo - 1 + j;
      const float hi = (float)nh + oh, wi = (float)nw + ow;
      const float hf = floorf(hi), wf = floorf(wi);
      const int h_low = (int)hf, w_low = (int)wf;
      const int ia = h_low - nh + 2, ib = w_low - nw + 2;
      const bool flag = own && hi > -1.f && wi > -1.f && hi < (float)H && wi < (float)W && !((unsigned)ia < 4u && (unsigned)ib < 4u);
      const float lh = hi - hf, lw = wi - wf;
      unsigned long long todo = __builtin_amdgcn_ballot_w64(flag);
      while (todo) {   // wave-uniform loop over the row's outlier samples of this tap; the whole wave serves one sample
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int s_h = __shfl(h_low, src), s_w = __shfl(w_low, src), s_wo = __shfl(wo, src);
        const float s_lh = __shfl(lh, src), s_lw = __shfl(lw, src);
        float cv = 0.f;
        for (int co = part; co < g.co; co += 8)
          cv += w[(long)co * (g.c * 9) + (dgi * 8 + cc) * 9 + t] * gop[(long)co * HW + s_wo];
        cv += __shfl_xor(cv, 1); cv += __shfl_xor(cv, 2); cv += __shfl_xor(cv, 4);
        if (part < 4) {   // one live corner each
          const int dy = part >> 1, dx = part & 1;
          const int yy = s_h + dy, xx = s_w + dx;
          const float wgt = (dy ? s_lh : 1.f - s_lh) * (dx ? s_lw : 1.f - s_lw);
          if (yy >= 0 && yy <= H - 1 && xx >= 0 && xx <= W - 1 && wgt != 0.f) atomicAdd(gim + yy * W + xx, wgt * cv);
        }
      }
    }
  }
}

static bool dcn_bwd_fused_ok(const DcnGeom& g, int step) {
  const char* e = getenv("RTP_DCN_NO_FUSED_BWD");
  if (e && atoi(e)) return false;
  if (!(g.group == 1 && g.dg == 4 && g.c == 32 && g.kh == 3 && g.kw == 3 && g.sh == 1 && g.sw == 1 && g.dh == 1 && g.dw == 1 &&
        g.ph == 1 && g.pw == 1 && g.co <= 32 && g.w % 4 == 0 && g.w >= 4 && g.h < 65536))
    return false;
  const long strips = (g.w + DCN_FB_XO - 1) / DCN_FB_XO;
  if (strips > 255 || (long)g.c * g.h * g.w * 4 >= (1L << 31) - (1L << 20)) return false;
  // the row list lives in the caller's workspace (sized for `step` images of columns, rtp_dcn_workspace_bytes)
  const long rec_bytes = 8 + 8 * (long)g.n * g.dg * strips * g.h;
  return rec_bytes <= (long)g.c * 9 * step * g.h * g.w * 4;
}

// gradWeight == nullptr: input / offset gradients only
static int dcn_backward_fused(const float* input, const float* offset, const float* gradOutput, float* gradInput,
                              float* gradOffset, const float* weight, float* gradWeight, float* ws, DcnGeom g, float scale,
                              hipStream_t s, int overwrite) {
  RtpProfScope prof(RTP_FAM_DCN, s);
  // overwrite: grad_input / grad_offset are stored by this pass (every element exactly once), only the weight gradient -- summed
  // with atomics -- needs its zeros
  if (overwrite && gradWeight && hipMemsetAsync(gradWeight, 0, sizeof(float) * g.co * g.c * 9, s) != hipSuccess) return RTP_ERR_LAUNCH;
  const int strips = (g.w + DCN_FB_XO - 1) / DCN_FB_XO;
  // row segments: one block per CU at a time (512 registers per lane), so the march takes ceil(blocks / CUs) rounds of
  // seg_rows + halo rows each; the segment count with the fewest rows on the critical path wins (128 images x 3 strips on 256 CUs:
  // two segments -> 3 rounds x 35 rows instead of 2 x 64)
  int segs = 1;
  const char* e = getenv("RTP_DCN_SEGS");
  if (e && atoi(e) > 0) segs = atoi(e);
  else {
    static int cus_dev[RTP_MAX_DEVICES] = {};   // CU count, per device
    const int dev = rtp_device_index();
    if (!cus_dev[dev]) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus_dev[dev] = prop.multiProcessorCount;
      if (cus_dev[dev] < 1) cus_dev[dev] = 256;
    }
    const int cus = cus_dev[dev];
    long best = -1;
    for (int sgs = 1; sgs <= 8 && sgs <= g.h; ++sgs) {
      const int rows = (g.h + sgs - 1) / sgs;
      const long rounds = ((long)g.n * strips * sgs + cus - 1) / cus;
      const long cost = rounds * (rows + (sgs == 1 ? 0 : sgs == 2 ? 3 : 6));
      if (best < 0 || cost < best) { best = cost; segs = sgs; }
    }
  }
  if (segs > g.h) segs = g.h;
  const int seg_rows = (g.h + segs - 1) / segs;
  segs = (g.h + seg_rows - 1) / seg_rows;
  unsigned* rec = (unsigned*)ws;
  if (hipMemsetAsync(rec, 0, 8, s) != hipSuccess) return RTP_ERR_LAUNCH;
  const size_t lds = sizeof(float) * 4 * (72 * DCN_FB_ROW + 8 * 7 * 64);   // tiles + rings: 135 680 B, one block per CU
  static bool attr[RTP_MAX_DEVICES] = {};
  if (rtp_once_per_device(attr)) {   // 135 KB of LDS: above the default limit, the launch fails without the attribute
    bool ok = hipFuncSetAttribute((const void*)dcn_bwd_fused_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)dcn_bwd_fused_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)dcn_bwd_fused_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void*)dcn_bwd_fused_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (!ok) { attr[rtp_device_index()] = false; return RTP_ERR_LAUNCH; }
  }
  const bool split = false;   // (weight gradient by dcn_gradw_fused_kernel, a gather pass of its own: measured slower, round 3)
  float* gwk = split ? nullptr : gradWeight;
  const char* fe = getenv("RTP_DCN_FP32_MFMA");  // the two products on the fp32 matrix instruction (exact products, 3x the cycles)
  const bool x3 = !(fe && atoi(fe));
  const dim3 grid((unsigned)((long)g.n * segs * strips));
#define DCN_FB_LAUNCH(GW_, X3_)                                                                                                \
  hipLaunchKernelGGL((dcn_bwd_fused_kernel<GW_, X3_>), grid, dim3(256), lds, s, input, offset, gradOutput, weight, gradInput,  \
                     gradOffset, gwk, rec, g, scale, strips, segs, seg_rows, overwrite)
  if (gwk) { if (x3) DCN_FB_LAUNCH(true, true); else DCN_FB_LAUNCH(true, false); }
  else { if (x3) DCN_FB_LAUNCH(false, true); else DCN_FB_LAUNCH(false, false); }
  if (split && dcn_gradw_fused_launch(input, offset, nullptr, gradOutput, gradWeight, g, scale, s) != RTP_OK) return RTP_ERR_LAUNCH;
  hipLaunchKernelGGL(dcn_bwd_outlier_rows_kernel, dim3(1024), dim3(64), 0, s, offset, gradOutput, weight, gradInput, rec, g);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}


static int dcn_backward_params(const float* input, const float* offset, const float* mask, const float* gradOutput,
                               float* gradWeight, float* ws, DcnGeom g, float scale, int step, hipStream_t s) {
  const int P = g.ho * g.wo, K = g.kh * g.kw;
  const int cg = g.c / g.group, cog = g.co / g.group;
  RtpProfScope prof(RTP_FAM_DCN, s);
  if (dcn_gradw_fused_ok(g)) {   // no column matrix: gather -> LDS transpose -> MFMA in one launch over all images (ws unused)
    if (dcn_gradw_fused_launch(input, offset, mask, gradOutput, gradWeight, g, scale, s) != RTP_OK) return RTP_ERR_LAUNCH;
    RTP_CHECK_LAUNCH();
    return RTP_OK;
  }
  for (int b0 = 0; b0 < g.n; b0 += step) {
    dcn_im2col(s, input, offset, mask, ws, g, b0, step);
    for (int gi = 0; gi < g.group; ++gi) {
      // gradW[gi] (cog x cg*K) += scale * gradOut chunk (cog x step*P) . columns[gi]^T (step*P x cg*K)
      if (!dcn_gradw_mfma(s, gradOutput + ((long)b0 * g.co + gi * cog) * P, ws + (long)gi * cg * K * step * P,
                          gradWeight + (long)gi * cog * cg * K, cog, cg * K, g.co, P, (long)step * P, scale))
        sgemm(s, mv(gradOutput + ((long)b0 * g.co + gi * cog) * P, P, 1, P, (long)g.co * P),
              mv(ws + (long)gi * cg * K * step * P, 1, (long)step * P),
              mv(gradWeight + (long)gi * cog * cg * K, (long)cg * K, 1), cog, cg * K, step * P, scale, 1.f);
    }
  }
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

#define DCN_GEOM_V1() DcnGeom g{n, c, h, w, co, kH, kW, dH, dW, padH, padW, dilH, dilW, group, deformable_group, 0, 0}

extern "C" int rtp_deform_conv_forward(const float* input, const float* weight, const float* offset, float* output,
                                       void* ws, int n, int c, int h, int w, int co, int kW, int kH, int dW, int dH,
                                       int padW, int padH, int dilW, int dilH, int group, int deformable_group,
                                       int im2col_step, void* stream) {
  if (!input || !weight || !offset || !output || !ws) return RTP_ERR_SHAPE;
  DCN_GEOM_V1();
  const int rc = dcn_check(g, im2col_step);
  if (rc) return rc;
  return dcn_forward(input, weight, nullptr, offset, nullptr, output, (float*)ws, g, im2col_step, (hipStream_t)stream);
}

extern "C" int rtp_deform_conv_backward_input(const float* input, const float* offset, const float* gradOutput,
                                              float* gradInput, float* gradOffset, const float* weight, void* ws, int n,
                                              int c, int h, int w, int co, int kW, int kH, int dW, int dH, int padW,
                                              int padH, int dilW, int dilH, int group, int deformable_group,
                                              int im2col_step, void* stream) {
  if (!input || !offset || !gradOutput || !gradInput || !gradOffset || !weight || !ws) return RTP_ERR_SHAPE;
  DCN_GEOM_V1();
  const int rc = dcn_check(g, im2col_step);
  if (rc) return rc;
  return dcn_backward_input(input, offset, nullptr, gradOutput, gradInput, gradOffset, nullptr, weight, (float*)ws, g,
                            im2col_step, (hipStream_t)stream);
}

extern "C" int rtp_deform_conv_backward_parameters(const float* input, const float* offset, const float* gradOutput,
                                                   float* gradWeight, void* ws, int n, int c, int h, int w, int co,
                                                   int kW, int kH, int dW, int dH, int padW, int padH, int dilW,
                                                   int dilH, int group, int deformable_group, float scale,
                                                   int im2col_step, void* stream) {
  if (!input || !offset || !gradOutput || !gradWeight || !ws) return RTP_ERR_SHAPE;
  DCN_GEOM_V1();
  const int rc = dcn_check(g, im2col_step);
  if (rc) return rc;
  return dcn_backward_params(input, offset, nullptr, gradOutput, gradWeight, (float*)ws, g, scale, im2col_step,
                             (hipStream_t)stream);
}

// Both halves of DeformConvFunction.backward (deform_conv.py:62-98 calls deform_conv_backward_input_cuda and then
// deform_conv_backward_parameters_cuda on the same tensors) in one call, so that the geometry of the DCN head runs the one-pass
// kernel with the weight gradient on board; other geometries run the two halves one after the other.
extern "C" int rtp_deform_conv_backward(const float* input, const float* offset, const float* gradOutput, float* gradInput,
                                        float* gradOffset, const float* weight, float* gradWeight, void* ws, int n, int c,
                                        int h, int w, int co, int kW, int kH, int dW, int dH, int padW, int padH, int dilW,
                                        int dilH, int group, int deformable_group, float scale, int im2col_step, void* stream) {
  if (!input || !offset || !gradOutput || !gradInput || !gradOffset || !weight || !gradWeight || !ws) return RTP_ERR_SHAPE;
  DCN_GEOM_V1();
  int rc = dcn_check(g, im2col_step);
  if (rc) return rc;
  if (dcn_bwd_fused_ok(g, im2col_step))
    return dcn_backward_fused(input, offset, gradOutput, gradInput, gradOffset, weight, gradWeight, (float*)ws, g, scale,
                              (hipStream_t)stream);
  rc = dcn_backward_input(input, offset, nullptr, gradOutput, gradInput, gradOffset, nullptr, weight, (float*)ws, g,
                          im2col_step, (hipStream_t)stream);
  if (rc) return rc;
  return dcn_backward_params(input, offset, nullptr, gradOutput, gradWeight, (float*)ws, g, scale, im2col_step,
                             (hipStream_t)stream);
}

// rtp_deform_conv_backward for callers that own fresh gradient buffers: the outputs are OVERWRITTEN, no zero-initialisation
// needed (the one-pass kernel stores every grad_input / grad_offset element exactly once; other geometries clear the buffers
// here and run the accumulating route).  Saves the 545 MB of memsets and the 168 MB read-back per call at [128,32,64,160].
extern "C" int rtp_deform_conv_backward_overwrite(const float* input, const float* offset, const float* gradOutput,
                                                  float* gradInput, float* gradOffset, const float* weight, float* gradWeight,
                                                  void* ws, int n, int c, int h, int w, int co, int kW, int kH, int dW, int dH,
                                                  int padW, int padH, int dilW, int dilH, int group, int deformable_group,
                                                  float scale, int im2col_step, void* stream) {
  if (!input || !offset || !gradOutput || !gradInput || !gradOffset || !weight || !gradWeight || !ws) return RTP_ERR_SHAPE;
  DCN_GEOM_V1();
  int rc = dcn_check(g, im2col_step);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (dcn_bwd_fused_ok(g, im2col_step))
    return dcn_backward_fused(input, offset, gradOutput, gradInput, gradOffset, weight, gradWeight, (float*)ws, g, scale, s, 1);
  const long P = (long)g.ho * g.wo;
  if (hipMemsetAsync(gradInput, 0, sizeof(float) * g.n * g.c * g.h * g.w, s) != hipSuccess ||
      hipMemsetAsync(gradOffset, 0, sizeof(float) * g.n * g.dg * 2 * g.kh * g.kw * P, s) != hipSuccess ||
      hipMemsetAsync(gradWeight, 0, sizeof(float) * g.co * (g.c / g.group) * g.kh * g.kw, s) != hipSuccess)
    return RTP_ERR_LAUNCH;
  rc = dcn_backward_input(input, offset, nullptr, gradOutput, gradInput, gradOffset, nullptr, weight, (float*)ws, g,
                          im2col_step, s);
  if (rc) return rc;
  return dcn_backward_params(input, offset, nullptr, gradOutput, gradWeight, (float*)ws, g, scale, im2col_step, s);
}

extern "C" int rtp_modulated_deform_conv_forward(const float* input, const float* weight, const float* bias,
                                                 const float* offset, const float* mask, float* output, void* ws,
                                                 int n, int c, int h, int w, int co, int kh, int kw, int sh, int sw,
                                                 int ph, int pw, int dh, int dw, int group, int deformable_group,
                                                 int with_bias, void* stream) {
  if (!input || !weight || !offset || !mask || !output || !ws || (with_bias && !bias)) return RTP_ERR_SHAPE;
  DcnGeom g{n, c, h, w, co, kh, kw, sh, sw, ph, pw, dh, dw, group, deformable_group, 0, 0};
  const int rc = dcn_check(g, 1);  // the reference walks the batch one image at a time (cpp:539)
  if (rc) return rc;
  return dcn_forward(input, weight, with_bias ? bias : nullptr, offset, mask, output, (float*)ws, g, 1,
                     (hipStream_t)stream);
}

extern "C" int rtp_modulated_deform_conv_backward(const float* input, const float* weight, const float* bias,
                                                  const float* offset, const float* mask, float* grad_input,
                                                  float* grad_weight, float* grad_bias, float* grad_offset,
                                                  float* grad_mask, const float* grad_output, void* ws, int n, int c,
                                                  int h, int w, int co, int kh, int kw, int sh, int sw, int ph, int pw,
                                                  int dh, int dw, int group, int deformable_group, int with_bias,
                                                  void* stream) {
  (void)bias;
  if (!input || !weight || !offset || !mask || !grad_input || !grad_weight || !grad_offset || !grad_mask ||
      !grad_output || !ws || (with_bias && !grad_bias))
    return RTP_ERR_SHAPE;
  DcnGeom g{n, c, h, w, co, kh, kw, sh, sw, ph, pw, dh, dw, group, deformable_group, 0, 0};
  int rc = dcn_check(g, 1);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  rc = dcn_backward_input(input, offset, mask, grad_output, grad_input, grad_offset, grad_mask, weight, (float*)ws, g, 1, s);
  if (rc) return rc;
  rc = dcn_backward_params(input, offset, mask, grad_output, grad_weight, (float*)ws, g, 1.f, 1, s);
  if (rc) return rc;
  if (with_bias) {
    hipLaunchKernelGGL(dcn_bias_grad_kernel, dim3(co), dim3(256), 0, s, grad_output, grad_bias, n, co, g.ho * g.wo);
    RTP_CHECK_LAUNCH();
  }
  return RTP_OK;
}
