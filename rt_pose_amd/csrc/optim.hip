// Fused optimiser step for the flat fp32 parameter buffer: global-norm clip + decoupled weight decay + Adam.
//   det3d/solver/fastai_optim.py:154-172 (OptimWrapper.step, true_wd), torchie/trainer/hooks/optimizer.py:14-24
//   (clip_grad_norm_ 35), torch.optim.Adam update.  hyper (device fp32[10]):
//   {lr, beta1, beta2, eps, wd, max_norm, bias_corr1, bias_corr2, grad_scale, unused}
#include "rtp_common.h"
#include "rtp_prof.h"

#define SQN_BLOCKS 256
extern "C" int rtp_sqnorm_blocks(void) { return SQN_BLOCKS; }

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* g, long n, const float* hyper, float* partial) {
  __shared__ float red[256];
  const float gs = hyper[8];
  float acc = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = g[i] * gs;
    acc += v * v;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

extern "C" int rtp_sqnorm(const float* g, long n, const float* hyper, float* partial, void* stream) {
  if (!g || !partial || !hyper) return RTP_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_OPTIM, s);
  hipLaunchKernelGGL(sqnorm_kernel, dim3(SQN_BLOCKS), dim3(256), 0, s, g, n, hyper, partial);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// mode 0: clip + decay + Adam ; mode 1: decay only (parameters that received no gradient this step)
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long n,
                                                   const float* hyper, const float* partial, int mode,
                                                   float* norm_out) {
  __shared__ float red[256];
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], max_norm = hyper[5];
  const float bc1 = hyper[6], bc2 = hyper[7], gs = hyper[8];
  float coef = 1.f;
  if (mode == 0) {
    red[threadIdx.x] = (threadIdx.x < SQN_BLOCKS) ? partial[threadIdx.x] : 0.f;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    const float total = sqrtf(red[0]);
    coef = fminf(max_norm / (total + 1e-6f), 1.f);
    if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = total;
  }
  const float decay = 1.f - wd * lr;
  const float step = lr / bc1, rs2 = rsqrtf(bc2);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float pv = p[i] * decay;
    if (mode == 0) {
      const float gv = g[i] * gs * coef;
      const float mv = b1 * m[i] + (1.f - b1) * gv;
      const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
      m[i] = mv;
      v[i] = vv;
      pv -= step * mv / (sqrtf(vv) * rs2 + eps);
    }
    p[i] = pv;
  }
}

extern "C" int rtp_adam_step(float* p, const float* g, float* m, float* v, long n, const float* hyper,
                             const float* sqnorm_partial, int mode, float* norm_out, void* stream) {
  if (!p || !hyper || n < 0 || (mode == 0 && (!g || !m || !v || !sqnorm_partial))) return RTP_ERR_SHAPE;
  if (n == 0) return RTP_OK;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_OPTIM, s);
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, s, p, g, m, v, n, hyper, sqnorm_partial, mode, norm_out);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

__global__ void zero_f32_kernel(float* dst, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = 0.f;
}

extern "C" int rtp_zero_f32(float* dst, long n, void* stream) {
  if (!dst || n < 1) return RTP_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_OPTIM, s);
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, n);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
