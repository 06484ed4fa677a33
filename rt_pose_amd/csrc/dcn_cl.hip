// Deformable convolution FORWARD on the plan's own layout (round 5): bf16 channels-last feature in, fp32 channels-last offsets
// in, bf16 channels-last output -- the deformable half of FeatureAdaption (center_head.py:24-62; DCNv1 3x3, stride 1, padding 1,
// 32 channels in 4 deformable groups of 8, deform_conv_cuda_kernel.cu:85-143 for the sampling rule) as the model-level plan runs
// it, per (frame, z) slice.
//
// The fp32 NCHW operator of include/rtp.h section D (dcn.hip) is bound by the texture-address path: in planar layout every
// (channel, corner row) of a sample is a load of its own -- 16 per sample.  In channels-last the 8 channels of a deformable group
// at one pixel are ONE 16-byte piece: 4 loads per sample, and no layout conversion either side (the operator needed the feature
// unpacked to fp32 planes, the offsets transposed, the result packed back: 1.2 GB of traffic per call at [128, 32, 64, 160]).
//   * a wave works on tiles of 16 consecutive output positions; lane (n = position, q) samples (group, tap) pairs 4 s + q,
//     s = 0 .. 8 -- its eight blended channels ARE the B operand of one k step of v_mfma_f32_16x16x32_bf16 (k = (group, tap,
//     channel): 288 = 9 steps of 32), so the samples go from the gather to the matrix core without crossing lanes or LDS;
//   * the offsets of sample 4 s + q are channels 8 s + 2 q, + 1 of the position's offset row: one 8-byte load per k step;
//   * A = the weights [16 cout][32 k] for the two output-channel tiles, packed once per workgroup into LDS (18 KB);
//   * all 36 corner loads of a tile are in flight together (144 VGPRs);
//   * the samples are rounded to bf16 before the product and the weights are bf16 -- the arithmetic of every other conv of the plan
//     (bf16 operands, fp32 accumulation); the backward pass stays on the fp32 operator.
// Out-of-image corners read zero through the buffer resource's range check (voffset 0x80000000), the per-corner bounds of the
// reference (kernel.cu:97-115).
#include "rtp_common.h"
#include "rtp_prof.h"

struct DcnClParams {
  const bf16_t* x; const float* off; const float* w; bf16_t* y;
  int x_cs, off_cs, y_cs;       // channel strides (elements); channel offsets folded into the pointers
  int NI, H, W;                 // images (frames x z slices), rows, columns
  int relu;
  long tiles;                   // NI * H * W / 16
};

#define DCL_OOB ((int)0x80000000)

// one sample of this lane: the four corner loads in flight and the bilinear weights they will be blended with
typedef unsigned dcl_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void dcn_cl_fwd_kernel(DcnClParams p) {
  __shared__ __attribute__((aligned(16))) bf16_t waL[9 * 2 * 64 * 8];   // A operands [k step][cout tile][lane] (18 KB)
  const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
  const long wave_id = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  // A operands: rows = output channels cot * 16 + n, columns = the 8 channels of sample 4 s + q: W[co][g * 8 + c][tap].
  // Built once per workgroup (the four waves hold identical operands: wave w packs k steps w, w + 4, w + 8)
  for (int s = threadIdx.x >> 6; s < 9; s += 4) {
    const int idx = 4 * s + q, g = idx / 9, t = idx - 9 * g;
#pragma unroll
    for (int cot = 0; cot < 2; ++cot) {
      const float* wr = p.w + ((long)(cot * 16 + n) * 32 + g * 8) * 9 + t;
      bf16x8 a;
#pragma unroll
      for (int c = 0; c < 8; ++c) a[c] = f2bf(wr[c * 9]);
      st_bf16x8(waL + ((s * 2 + cot) * 64 + lane) * 8, a);
    }
  }
  __syncthreads();
  // this lane's nine samples: idx = 4 s + q -> group idx / 9, tap idx % 9 = (i, j)
  int s_gc[9], s_di[9], s_dj[9];
#pragma unroll
  for (int s = 0; s < 9; ++s) {
    const int idx = 4 * s + q, g = idx / 9, t = idx - 9 * g;
    s_gc[s] = g * 16;                    // byte offset of the group's 8 channels within a pixel
    s_di[s] = t / 3 - 1;
    s_dj[s] = t % 3 - 1;
  }
  const int HW = p.H * p.W;
  const int rs = p.W * p.x_cs * 2, cs2 = p.x_cs * 2;
  for (long tile = wave_id; tile < p.tiles; tile += nwaves) {
    const long pos = tile * 16 + n;                 // global position: image * HW + ho * W + wo
    const int b = (int)(tile * 16 / HW), pin = (int)(pos - (long)b * HW);   // (H * W % 16 == 0: a tile stays inside one image)
    const int ho = pin / p.W, wo = pin - ho * p.W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (long)b * HW * p.x_cs), 0,
                                                                        (HW - 1) * p.x_cs * 2 + 64, 0x00020000);
    const float* orow = p.off + pos * p.off_cs + 2 * q;
    float2 o[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) o[s] = *reinterpret_cast<const float2*>(orow + 8 * s);
    // (requesting the NEXT tile's offsets a tile ahead changed nothing: 291 us either way -- the kernel is bound by its vector
    // instructions and the 45 wave loads per tile, not by the offset -> corner dependency)
    // ALL 36 corner loads of the tile are issued before the first is used (issued sample by sample behind a wait, the kernel
    // was latency-bound: 397 us at [128, 32, 64, 160]); the bilinear weights wait beside them
    dcl_u32x4 v[9][4];
    float blh[9], blw[9];   // the fractional parts: the four bilinear weights are rebuilt at use (18 registers instead of 36)
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      const float hi = (float)(ho + s_di[s]) + o[s].x, wi = (float)(wo + s_dj[s]) + o[s].y;
      const bool in = hi > -1.f && wi > -1.f && hi < (float)p.H && wi < (float)p.W;
      const float hf = floorf(hi), wf = floorf(wi);
      const int h_low = (int)hf, w_low = (int)wf;
      const float lh = hi - hf, lw = wi - wf;
      const bool r0 = in && h_low >= 0, r1 = in && h_low + 1 <= p.H - 1, c0 = w_low >= 0, c1 = w_low + 1 <= p.W - 1;
      const int base = ((h_low * p.W + w_low) * p.x_cs) * 2 + s_gc[s];
      v[s][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, (r0 && c0) ? base : DCL_OOB, 0, 0);
      v[s][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, (r0 && c1) ? base + cs2 : DCL_OOB, 0, 0);
      v[s][2] = __builtin_amdgcn_raw_buffer_load_b128(rx, (r1 && c0) ? base + rs : DCL_OOB, 0, 0);
      v[s][3] = __builtin_amdgcn_raw_buffer_load_b128(rx, (r1 && c1) ? base + rs + cs2 : DCL_OOB, 0, 0);
      blh[s] = lh; blw[s] = lw;
    }
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks every sample's loads next to their use)
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      bf16x8 sv;
      const float hh = 1.f - blh[s], hw = 1.f - blw[s];
      const float bw[4] = {hh * hw, hh * blw[s], blh[s] * hw, blh[s] * blw[s]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {   // two channels per dword: low half = even channel
        float lo = 0.f, hi2 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          lo += bw[c] * __uint_as_float(v[s][c][k] << 16);
          hi2 += bw[c] * __uint_as_float(v[s][c][k] & 0xffff0000u);
        }
        sv[2 * k] = f2bf(lo);
        sv[2 * k + 1] = f2bf(hi2);
      }
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_bf16x8(waL + ((s * 2 + 0) * 64 + lane) * 8), sv, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_bf16x8(waL + ((s * 2 + 1) * 64 + lane) * 8), sv, acc[1], 0, 0, 0);
    }
    // lane (n, q) holds output channels cot * 16 + 4 q .. + 3 of position n
    bf16_t* yp = p.y + pos * p.y_cs + 4 * q;
#pragma unroll
    for (int cot = 0; cot < 2; ++cot) {
      bf16x4 ob;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float vv = acc[cot][j];
        ob[j] = f2bf(p.relu ? (vv > 0.f ? vv : 0.f) : vv);
      }
      *reinterpret_cast<bf16x4*>(yp + cot * 16) = ob;
    }
  }
}

/* include/rtp.h: rtp_dcn_cl_forward */
extern "C" int rtp_dcn_cl_forward(const RtpAct* x, const RtpAct* off, const float* w, const RtpAct* y, int n_img, int h, int w_,
                                  int relu, void* stream) {
  if (!x || !off || !w || !y || n_img < 1 || h < 1 || w_ < 1) return RTP_ERR_SHAPE;
  if (x->c < 32 || y->c < 32 || off->c < 72) return RTP_ERR_SHAPE;
  if ((x->cs % 8) || (x->co % 8) || (y->cs % 4) || (y->co % 4) || (off->cs % 2) || (off->co % 2)) return RTP_ERR_ALIGN;
  const long npos = (long)n_img * h * w_;
  if (((long)h * w_) % 16 || (long)h * w_ * x->cs * 2 >= (1L << 31)) return RTP_ERR_UNSUPPORTED;
  DcnClParams p;
  p.x = (const bf16_t*)x->ptr + x->co; p.off = (const float*)off->ptr + off->co; p.w = w; p.y = (bf16_t*)y->ptr + y->co;
  p.x_cs = x->cs; p.off_cs = off->cs; p.y_cs = y->cs;
  p.NI = n_img; p.H = h; p.W = w_; p.relu = relu; p.tiles = npos / 16;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_DCN, s);
  long blocks = (p.tiles + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dcn_cl_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
