// CenterHead losses (forward + backward in one pass) and the argmax decode.
//   FastFocalLoss  det3d/models/losses/centernet_loss.py:34-54 (+ _sigmoid clamp, pose_heads/center_head.py:240-242)
//   RegLoss        centernet_loss.py:17-24 (+ code weights, center_head.py:252-258)
//   predict        center_head.py:272-360
// Head logits are fp32 channels-last [n][vox][cpad]; heat-map targets arrive as the dataset makes them
// (fp32 NCDHW); gradients are written as bf16 channels-last with the channel count padded to 32/64 so the
// data-gradient conv contracts a full MFMA K step.
#include "rtp_common.h"
#include "rtp_prof.h"

#define FOCAL_BPS 320  // blocks per sample
extern "C" int rtp_focal_blocks(void) { return FOCAL_BPS; }

struct FocalParams {
  const float* logits; int cpad; const float* target; const long long* ind; const unsigned char* mask;
  const long long* cat; int n, ncls, m; long vox; float gscale; float* scratch; bf16_t* g; int g_cs, g_co, g_c;
  int g_write;  // channels of the gradient row actually stored: g_c, or only the chunks that hold classes (rtp_focal_loss_ex)
};

// Thread per voxel (the fp32 NCDHW targets are then coalesced across lanes); a lane reads its voxel's logits as
// 16-B vectors and writes the padded bf16 gradient row as 16-B vectors.  The positives that fall into a block's voxel
// range (usually none) are listed once per block, in index order, so the per-voxel work has no loop over objects.
__global__ __launch_bounds__(256) void focal_kernel(FocalParams p) {
  __shared__ int s_pv[64], s_pc[64];
  __shared__ float s_pm[64];
  __shared__ int s_np;
  __shared__ float s_npos;
  __shared__ float red[8 * 2];
  const int n = blockIdx.y, tid = threadIdx.x;
  const long vps = (p.vox + gridDim.x - 1) / gridDim.x;
  const long v0 = blockIdx.x * vps, v1 = (v0 + vps < p.vox) ? v0 + vps : p.vox;
  if (tid == 0) {
    int k = 0;
    for (int i = 0; i < p.m; ++i) {
      const long long v = p.ind[(long)n * p.m + i];
      if (v >= v0 && v < v1) {
        s_pv[k] = (int)(v - v0); s_pc[k] = (int)p.cat[(long)n * p.m + i];
        s_pm[k] = p.mask[(long)n * p.m + i] ? 1.f : 0.f; ++k;
      }
    }
    s_np = k;
  }
  {   // number of positives: all threads (a single lane walking n*m bytes paid a load latency per object), fixed-order fold
    float np = 0.f;
    for (int i = tid; i < p.n * p.m; i += 256) np += p.mask[i] ? 1.f : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) np += __shfl_xor(np, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = np;
  }
  __syncthreads();
  if (tid == 0) s_npos = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  const float denom = s_npos > 0.f ? s_npos : 1.f;
  const float gs = -p.gscale / denom;
  const int npos = s_np;
  float negsum = 0.f, possum = 0.f;
  for (long v = v0 + tid; v < v1; v += 256) {
    const float* lg = p.logits + ((long)n * p.vox + v) * p.cpad;
    bf16_t* gout = p.g + ((long)n * p.vox + v) * p.g_cs + p.g_co;
    const float* tg = p.target + (long)n * p.ncls * p.vox + v;
    const int vrel = (int)(v - v0);
    for (int c8 = 0; c8 < p.g_write; c8 += 8) {
      float z[8], gt[8];
      bf16x8 o;
      if (c8 < p.cpad) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(lg + c8), b = *reinterpret_cast<const f32x4*>(lg + c8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { z[j] = a[j]; z[4 + j] = b[j]; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) gt[j] = (c8 + j < p.ncls) ? tg[(long)(c8 + j) * p.vox] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c8 + j;
        float grad = 0.f;
        if (c < p.ncls) {
          const float praw = __frcp_rn(1.f + __expf(-z[j]));
          const bool inside = (praw >= 1e-4f) && (praw <= 1.f - 1e-4f);
          const float pc = fminf(fmaxf(praw, 1e-4f), 1.f - 1e-4f);
          float w = 1.f - gt[j]; w = w * w; w = w * w;
          const float omp = 1.f - pc;
          const float l1p = __logf(omp);
          negsum += l1p * pc * pc * w;
          float dldp = (-pc * pc * __frcp_rn(omp) + 2.f * pc * l1p) * w;
          for (int k = 0; k < npos; ++k)
            if (s_pv[k] == vrel && s_pc[k] == c) {
              const float lp = __logf(pc);
              possum += lp * omp * omp * s_pm[k];
              dldp += (omp * omp * __frcp_rn(pc) - 2.f * omp * lp) * s_pm[k];
            }
          grad = inside ? gs * dldp * praw * (1.f - praw) : 0.f;
        }
        o[j] = f2bf(grad);
      }
      st_bf16x8(gout + c8, o);
    }
  }
  // fixed-order block reduction: lanes of a wave by xor-shuffle, then the 4 waves in order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { negsum += __shfl_xor(negsum, o, 64); possum += __shfl_xor(possum, o, 64); }
  if ((tid & 63) == 0) { red[(tid >> 6) * 2] = negsum; red[(tid >> 6) * 2 + 1] = possum; }
  __syncthreads();
  if (tid == 0) {
    float* sc = p.scratch + ((long)n * gridDim.x + blockIdx.x) * 2;
    sc[0] = (red[0] + red[2]) + (red[4] + red[6]); sc[1] = (red[1] + red[3]) + (red[5] + red[7]);
  }
}

__global__ __launch_bounds__(256) void focal_final(const float* scratch, int nparts, const unsigned char* mask, int nm,
                                                   float* out) {
  __shared__ float red[256 * 2];
  const int tid = threadIdx.x;
  float neg = 0.f, pos = 0.f;
  for (int i = tid; i < nparts; i += 256) { neg += scratch[2 * i]; pos += scratch[2 * i + 1]; }
  red[tid] = neg; red[256 + tid] = pos;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { red[tid] += red[tid + o]; red[256 + tid] += red[256 + tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    float np = 0.f;
    for (int i = 0; i < nm; ++i) np += mask[i] ? 1.f : 0.f;
    out[0] = (np == 0.f) ? -red[0] : -(red[256] + red[0]) / np;
  }
}

extern "C" int rtp_focal_loss_ex(const float* logits, int cpad, const float* target, const long long* ind,
                                 const unsigned char* mask, const long long* cat, int n, int ncls, long vox, int m,
                                 float gscale, float* scratch, float* out_loss, const RtpAct* ghm, int write_pad, void* stream);

extern "C" int rtp_focal_loss(const float* logits, int cpad, const float* target, const long long* ind,
                              const unsigned char* mask, const long long* cat, int n, int ncls, long vox, int m,
                              float gscale, float* scratch, float* out_loss, const RtpAct* ghm, void* stream) {
  return rtp_focal_loss_ex(logits, cpad, target, ind, mask, cat, n, ncls, vox, m, gscale, scratch, out_loss, ghm, 1, stream);
}

// write_pad == 0: the padding channels of the gradient rows (those past the last 8-channel chunk that holds a class) are NOT
// stored -- for a buffer that was zeroed once and is written by nothing else (they stay zero); saves a third of the kernel's
// HBM writes at 15 classes in a 32-channel row.
extern "C" int rtp_focal_loss_ex(const float* logits, int cpad, const float* target, const long long* ind,
                                 const unsigned char* mask, const long long* cat, int n, int ncls, long vox, int m,
                                 float gscale, float* scratch, float* out_loss, const RtpAct* ghm, int write_pad, void* stream) {
  if (!logits || !target || !ghm || m > 64 || ncls > cpad || ncls > ghm->c) return RTP_ERR_SHAPE;
  if ((cpad % 8) || (ghm->c % 8) || (ghm->cs % 8) || (ghm->co % 8)) return RTP_ERR_ALIGN;
  FocalParams p{logits, cpad, target, ind, mask, cat, n, ncls, m, vox, gscale, scratch,
                (bf16_t*)ghm->ptr, ghm->cs, ghm->co, ghm->c, write_pad ? ghm->c : (ncls + 7) / 8 * 8};
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_LOSS, s);
  hipLaunchKernelGGL(focal_kernel, dim3(FOCAL_BPS, n), dim3(256), 0, s, p);
  hipLaunchKernelGGL(focal_final, dim3(1), dim3(256), 0, s, scratch, n * FOCAL_BPS, mask, n * m, out_loss);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// RegLoss: single block; (n*m) rows are few
// ------------------------------------------------------------------------------------------------
struct RegParams {
  const float* reg; int cpad; const float* target; const long long* ind; const unsigned char* mask; const float* cw;
  int n, nreg, m; long vox; float gscale; float* out; bf16_t* g; int g_cs, g_co;
};

// One block.  Rows (sample, object) are spread over the threads and reduced through LDS in fixed order (a single thread walking
// the 120 rows paid one global-load latency per row: 60 us of a 75 us launch).  prev (optional, device, [n*m]): the voxels this
// kernel wrote in the previous step -- only those are cleared instead of zero-filling the whole gradient tensor (it is zero
// everywhere else by construction: nothing else writes it), then this step's voxels are recorded there.
__global__ __launch_bounds__(256) void reg_loss_kernel(RegParams p, long long* prev) {
  __shared__ float s_red[256];
  __shared__ float s_loss[64];
  __shared__ float s_msum;
  const int tid = threadIdx.x, rows = p.n * p.m;
  if (prev) {   // clear last step's voxels (-1: nothing recorded yet)
    for (int i = tid; i < rows * p.nreg; i += 256) {
      const int c = i % p.nreg, r = i / p.nreg, n = r / p.m;
      const long long v = prev[r];
      if (v >= 0) p.g[((long)n * p.vox + v) * p.g_cs + p.g_co + c] = f2bf(0.f);
    }
  }
  {
    float ms = 0.f;
    for (int r = tid; r < rows; r += 256) ms += p.mask[r] ? 1.f : 0.f;
    s_red[tid] = ms;
    __syncthreads();
    if (tid == 0) {
      float a = 0.f;
      for (int k = 0; k < 256; ++k) a += s_red[k];
      s_msum = a;
    }
    __syncthreads();
  }
  const float inv = 1.f / (s_msum + 1e-4f);
  // per-channel loss: 256 / 64 row groups x up to 64 channels, partial sums folded in fixed order
  {
    const int c = tid & 63, grp = tid >> 6;
    float acc = 0.f;
    if (c < p.nreg)
      for (int r = grp; r < rows; r += 4) {
        const int n = r / p.m;
        const float mk = p.mask[r] ? 1.f : 0.f;
        const float pred = p.reg[((long)n * p.vox + p.ind[r]) * p.cpad + c];
        acc += fabsf(pred * mk - p.target[(long)r * p.nreg + c] * mk) * inv;
      }
    s_red[tid] = acc;
    __syncthreads();
    if (tid < p.nreg) {
      const float a = (s_red[tid] + s_red[64 + tid]) + (s_red[128 + tid] + s_red[192 + tid]);
      s_loss[tid] = a;
      p.out[tid] = a;
    }
    __syncthreads();
  }
  if (tid == 0) {
    float loc = 0.f;
    for (int c = 0; c < p.nreg; ++c) loc += s_loss[c] * p.cw[c];
    p.out[p.nreg] = loc;
  }
  __syncthreads();   // (the clears above are ordered before the writes below: same thread set, barrier in between)
  // gradient scatter: the first row of each (sample, voxel) group sums its duplicates
  for (int i = tid; i < rows * p.nreg; i += 256) {
    const int c = i % p.nreg, r = i / p.nreg, n = r / p.m;
    bool first = true;
    for (int r2 = n * p.m; r2 < r; ++r2) first = first && (p.ind[r2] != p.ind[r]);
    if (!first) continue;
    float gsum = 0.f;
    for (int r2 = r; r2 < (n + 1) * p.m; ++r2) {
      if (p.ind[r2] != p.ind[r]) continue;
      const float mk = p.mask[r2] ? 1.f : 0.f;
      const float pred = p.reg[((long)n * p.vox + p.ind[r2]) * p.cpad + c];
      const float d = pred * mk - p.target[(long)r2 * p.nreg + c] * mk;
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      gsum += sg * mk;
    }
    p.g[((long)n * p.vox + p.ind[r]) * p.g_cs + p.g_co + c] = f2bf(p.gscale * p.cw[c] * inv * gsum);
  }
  if (prev) {
    __syncthreads();
    for (int r = tid; r < rows; r += 256) prev[r] = p.ind[r];
  }
}

static int reg_loss_launch(const float* reg, int cpad, const float* target, const long long* ind, const unsigned char* mask,
                           const float* code_w, int n, int nreg, long vox, int m, float gscale, float* out, const RtpAct* greg,
                           long long* prev, void* stream) {
  if (!reg || !target || !greg || nreg > 64 || nreg > cpad || nreg > greg->c) return RTP_ERR_SHAPE;
  if (greg->co != 0 || greg->cs != greg->c) return RTP_ERR_SHAPE;  // zero-filled as one contiguous buffer
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_LOSS, s);
  if (!prev && hipMemsetAsync(greg->ptr, 0, (size_t)n * vox * greg->cs * sizeof(bf16_t), s) != hipSuccess) return RTP_ERR_LAUNCH;
  RegParams p{reg, cpad, target, ind, mask, code_w, n, nreg, m, vox, gscale, out, (bf16_t*)greg->ptr, greg->cs, greg->co};
  hipLaunchKernelGGL(reg_loss_kernel, dim3(1), dim3(256), 0, s, p, prev);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

extern "C" int rtp_reg_loss(const float* reg, int cpad, const float* target, const long long* ind,
                            const unsigned char* mask, const float* code_w, int n, int nreg, long vox, int m,
                            float gscale, float* out, const RtpAct* greg, void* stream) {
  return reg_loss_launch(reg, cpad, target, ind, mask, code_w, n, nreg, vox, m, gscale, out, greg, nullptr, stream);
}

// The same with a state buffer instead of the zero fill: prev_ind (device, int64 [n*m], initialised to -1 by the caller; greg
// zero-initialised once) holds the voxels written by the previous call, which are the only non-zero ones -- a call clears them,
// writes this step's and records them.  greg must not be written by anything else.
extern "C" int rtp_reg_loss_sparse(const float* reg, int cpad, const float* target, const long long* ind,
                                   const unsigned char* mask, const float* code_w, int n, int nreg, long vox, int m,
                                   float gscale, float* out, const RtpAct* greg, long long* prev_ind, void* stream) {
  if (!prev_ind) return RTP_ERR_SHAPE;
  return reg_loss_launch(reg, cpad, target, ind, mask, code_w, n, nreg, vox, m, gscale, out, greg, prev_ind, stream);
}

// ------------------------------------------------------------------------------------------------
// decode: per (sample, class) first-index argmax of sigmoid(logit), then offset decode
// ------------------------------------------------------------------------------------------------
#define DEC_BPS 64
struct DecParams {
  const float* logits; int hm_cpad; const float* reg; int reg_cpad; int n, ncls, nreg, d, h, w;
  float sx, sy, sz, ox, oy, oz; float* part; float* out;
};

__global__ __launch_bounds__(256) void decode_scan(DecParams p) {
  __shared__ float s_val[256];
  __shared__ int s_idx[256];
  const int n = blockIdx.y, tid = threadIdx.x;
  const long vox = (long)p.d * p.h * p.w;
  const long vps = (vox + gridDim.x - 1) / gridDim.x;
  const long v0 = blockIdx.x * vps, v1 = (v0 + vps < vox) ? v0 + vps : vox;
  for (int c = 0; c < p.ncls; ++c) {
    float best = -1.f; int bi = 0x7fffffff;
    for (long v = v0 + tid; v < v1; v += 256) {
      const float sg = 1.f / (1.f + expf(-p.logits[((long)n * vox + v) * p.hm_cpad + c]));
      if (sg > best) { best = sg; bi = (int)v; }
    }
    s_val[tid] = best; s_idx[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        const float ov = s_val[tid + o]; const int oi = s_idx[tid + o];
        if (ov > s_val[tid] || (ov == s_val[tid] && oi < s_idx[tid])) { s_val[tid] = ov; s_idx[tid] = oi; }
      }
      __syncthreads();
    }
    if (tid == 0) {
      float* o = p.part + (((long)n * p.ncls + c) * gridDim.x + blockIdx.x) * 2;
      o[0] = s_val[0]; o[1] = __int_as_float(s_idx[0]);
    }
    __syncthreads();
  }
}

__global__ void decode_final(DecParams p, int nparts) {
  const int n = blockIdx.x / p.ncls, c = blockIdx.x % p.ncls;
  if (threadIdx.x != 0) return;
  const long vox = (long)p.d * p.h * p.w;
  float best = -1.f; int bi = 0x7fffffff;
  for (int i = 0; i < nparts; ++i) {
    const float* q = p.part + (((long)n * p.ncls + c) * nparts + i) * 2;
    const float v = q[0]; const int idx = __float_as_int(q[1]);
    if (v > best || (v == best && idx < bi)) { best = v; bi = idx; }
  }
  int z, y, x;
  vox_decode(bi, p.h, p.w, z, y, x);
  const int nk = p.nreg / 3;
  float* o = p.out + ((long)n * p.ncls + c) * (2 + p.nreg);
  o[0] = (float)bi; o[1] = best;
  const float* r = p.reg + ((long)n * vox + bi) * p.reg_cpad;
  for (int k = 0; k < nk; ++k) {
    o[2 + 3 * k + 0] = ((float)x + r[3 * k + 0]) * p.sx + p.ox;
    o[2 + 3 * k + 1] = ((float)y + r[3 * k + 1]) * p.sy + p.oy;
    o[2 + 3 * k + 2] = ((float)z + r[3 * k + 2]) * p.sz + p.oz;
  }
}

extern "C" int rtp_decode_scratch_floats(int n, int ncls) { return n * ncls * DEC_BPS * 2; }

extern "C" int rtp_decode(const float* logits, int hm_cpad, const float* reg, int reg_cpad, int n, int ncls, int nreg,
                          int d, int h, int w, const float* scale_xyz /*host*/, const float* origin_xyz /*host*/,
                          float* scratch, float* out, void* stream) {
  if (!logits || !reg || !out || !scratch || nreg % 3) return RTP_ERR_SHAPE;
  DecParams p{logits, hm_cpad, reg, reg_cpad, n, ncls, nreg, d, h, w,
              scale_xyz[0], scale_xyz[1], scale_xyz[2], origin_xyz[0], origin_xyz[1], origin_xyz[2], scratch, out};
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_LOSS, s);
  hipLaunchKernelGGL(decode_scan, dim3(DEC_BPS, n), dim3(256), 0, s, p);
  hipLaunchKernelGGL(decode_final, dim3(n * ncls), dim3(64), 0, s, p, DEC_BPS);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
