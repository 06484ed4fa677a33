// Native 64 -> 64-channel 3x3x3 stride-1 convolution (forward and, with flipped taps, data gradient) for the feat64 configurations
// (hrnet3D_config.py:149-177: branches of 64 / 64 / 128 / 128 channels; hr_util/common.py:73-148 at Cin = Cout = 64).
//
// Until round 5 these layers ran as four 32 -> 32 slices of conv_tiled.hip through an fp32 workspace (rtp_conv_igemm_ws): four
// launches per layer, each of which re-reads and re-writes 168 MB of fp32 partial sums at the native shape -- 445-500 us per
// full-resolution layer for 290 GFLOP.  A 64 x 64 x 27 weight image is 221 KB, more than a CU's LDS, so the image does not go to
// LDS at all here: it lives in the REGISTERS of the workgroup's eight waves.
//   * wave (ct, kh) owns output-channel tile ct (16 channels) and input-channel half kh (32 channels) for the workgroup's lifetime:
//     27 taps x one MFMA A operand (16 cout x 32 cin bf16 = 4 VGPRs) = 108 VGPRs, loaded once;
//   * a brick is 2(z) x 4(y) x 16(x) output voxels; its haloed input 4 x 6 x 18 voxels x 128 B (55 KB) is staged by LDS-DMA, double
//     buffered (110 KB): brick b+1 is requested before the MFMAs of brick b start;
//   * every wave sweeps the WHOLE brick: 8 accumulator tiles (v_mfma_f32_16x16x32_bf16, B = 16 voxels of an x-row x 32 cin); for a
//     fixed dx each haloed row is read from LDS once and feeds the up to six (dz, dy) taps it contributes to -- 72 ds_read_b128 for
//     216 MFMAs per wave and brick (conv_tiled: one read per two MFMAs);
//   * the two input-channel halves meet in LDS: after its MFMAs one wave of each pair writes its eight tiles to an exchange area,
//     the other adds them and runs the epilogue (class bias, residual, ReLU, bf16 store, statistics) -- the roles alternate from
//     brick to brick, so on every SIMD one wave's epilogue runs beside the other wave's next MFMA loop, and ONE workgroup barrier
//     per brick orders everything (brick buffers, exchange area);
//   * the LDS image rotates the 16-B chunk index by 2 * (x >> 1) (mod 8) -- with 128 B per voxel two voxels share a 256-B bank row,
//     and the rotation makes every ds_read_b128 lane group hit 16 distinct 16-B slots for each of the three x shifts.
// Results: the same arithmetic as the slice chain up to fp32 summation order (one accumulator chain per input-channel half).
#include <stdlib.h>

#include "rtp_common.h"
#include "rtp_multi.h"
#include "rtp_prof.h"

#define C64_TZ 2
#define C64_TY 4
#define C64_TX 16
#define C64_HZ (C64_TZ + 2)
#define C64_HY (C64_TY + 2)
#define C64_HX (C64_TX + 2)
#define C64_HVOX (C64_HZ * C64_HY * C64_HX)      // 432 haloed voxels
#define C64_ROW_B (C64_HX * 128)                   // 2304 B: nine 256-B bank rows
#define C64_BRICK_B (C64_HVOX * 128)               // 55 296 B
#define C64_ITEMS (C64_HVOX * 8)                   // sixteen-byte pieces of a brick
#define C64_ROUNDS ((C64_ITEMS + 511) / 512)       // 7 (the last one: waves 0-5 only)
#define C64_XCH_B (4 * 8 * 1024)                   // exchange area: 4 channel tiles x 8 voxel tiles x 64 lanes x 16 B
#define C64_RED_B (8 * 16 * 2 * 4)                 // statistics hand-over: 8 waves x 16 channels x 2
#define C64_LDS_B (2 * C64_BRICK_B + C64_XCH_B + C64_RED_B)

__device__ __attribute__((aligned(16))) bf16_t g_zero_line_c64[8];

struct C64Params {
  const bf16_t* x; const bf16_t* w; const float* btab; const bf16_t* aux; bf16_t* y; float* stat_out;
  int N, D, H, W;
  int x_cs, x_co, y_cs, y_co, a_cs, a_co;
  int relu, flip, w_per_sample;
  int aux_mode;                 // 0 none, 1 residual (added), 2 second operand of the statistics (sum y * aux instead of sum y^2)
  int wgs_per_sample, tiles_x, tiles_y, tiles_per_sample;
};

__device__ __forceinline__ void c64_dma16(const bf16_t* src, unsigned lds_wave_base) {
  // (inline assembly: the compiler must not model the transfer -- see lds_dma16 in conv_tiled.hip)
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(src) : "memory", "m0");
}

__global__ __launch_bounds__(512, 2) void conv64_kernel(C64Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 3, kh = wave >> 2;
  const int v = lane & 15, q = lane >> 4;
  // XCD-aware placement (as conv_tiled_kernel): every XCD one contiguous run of logical workgroups
  const int bid = (gridDim.x % 8 == 0) ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int n = bid / p.wgs_per_sample, wg = bid - n * p.wgs_per_sample;
  const int t_begin = (int)((unsigned)wg * (unsigned)p.tiles_per_sample / (unsigned)p.wgs_per_sample);
  const int t_end = (int)((unsigned)(wg + 1) * (unsigned)p.tiles_per_sample / (unsigned)p.wgs_per_sample);
  const int nb = t_end - t_begin;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  unsigned char* xch = lds + 2 * C64_BRICK_B;
  float* red = reinterpret_cast<float*>(lds + 2 * C64_BRICK_B + C64_XCH_B);

  // ---- this wave's share of the weight image, into registers (rows = this launch's output channels, 64 input channels per row)
  bf16x8 wreg[27];
  {
    const bf16_t* wsrc = p.w + (p.w_per_sample ? (long)n * 27 * 64 * 64 : 0) + (ct * 16 + v) * 64 + kh * 32 + q * 8;
#pragma unroll
    for (int t = 0; t < 27; ++t) wreg[t] = ld_bf16x8(wsrc + (p.flip ? 26 - t : t) * (64 * 64));
  }
  // ---- staging descriptors (brick independent): this thread's pieces of a brick
  int s_rel[C64_ROUNDS], s_pk[C64_ROUNDS];
#pragma unroll
  for (int k = 0; k < C64_ROUNDS; ++k) {
    const int i = k * 512 + tid;
    const int cp = i & 7, hv = i >> 3;
    const int hx = hv % C64_HX, hy = (hv / C64_HX) % C64_HY, hz = hv / (C64_HX * C64_HY);
    const int ck = (cp - 2 * (hx >> 1)) & 7;   // the logical chunk whose rotated position is cp
    s_pk[k] = (hz == 0) | ((hz == C64_HZ - 1) << 1) | ((hy == 0) << 2) | ((hy == C64_HY - 1) << 3) | ((hx == 0) << 4) |
              ((hx == C64_HX - 1) << 5);
    s_rel[k] = ((hz * p.H + hy) * p.W + hx) * p.x_cs + ck * 8;
  }
  const long vox_n = (long)n * p.D * p.H * p.W;
  const bf16_t* xn = p.x + vox_n * p.x_cs + p.x_co;
  auto stage = [&](int t, int buf) {   // workgroup-uniform arguments
    const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, tz = t / (p.tiles_x * p.tiles_y);
    const int z0 = tz * C64_TZ, y0 = ty * C64_TY, x0 = tx * C64_TX;
    const int tflg = (z0 == 0) | ((z0 + C64_TZ == p.D) << 1) | ((y0 == 0) << 2) | ((y0 + C64_TY == p.H) << 3) | ((x0 == 0) << 4) |
                     ((x0 + C64_TX == p.W) << 5);
    const int org = (((z0 - 1) * p.H + (y0 - 1)) * p.W + (x0 - 1)) * p.x_cs;
#pragma unroll
    for (int k = 0; k < C64_ROUNDS; ++k) {
      if (k * 512 + (tid & ~63) < C64_ITEMS) {   // wave-uniform (3456 = 54 waves' worth)
        const bf16_t* src = (s_pk[k] & tflg) ? g_zero_line_c64 : xn + org + s_rel[k];
        c64_dma16(src, lds0 + buf * C64_BRICK_B + (k * 512 + (tid & ~63)) * 16);
      }
    }
  };
  // per-lane byte offsets of the B operand (voxel v of an x-row, this wave's input-channel half) for the three x shifts
  int boff[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int hx = v + dx;
    boff[dx] = hx * 128 + (((kh * 4 + q) + 2 * (hx >> 1)) & 7) * 16;
  }
  const int c0 = ct * 16 + q * 4;   // the four output channels this lane's accumulator elements stand for
  float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};

  stage(t_begin, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int it = 0; it < nb; ++it) {
    const int t = t_begin + it, buf = it & 1;
    if (it + 1 < nb) stage(t + 1, buf ^ 1);   // (that buffer was last read under brick it-1: every wave has passed its barrier)
    f32x4 acc[C64_TZ][C64_TY];
#pragma unroll
    for (int zo = 0; zo < C64_TZ; ++zo)
#pragma unroll
      for (int yo = 0; yo < C64_TY; ++yo) acc[zo][yo] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (issue order left to the compiler: with two MFMA waves per SIMD its read-then-use order measured 281 us at the native shape,
    // an explicit one-group-ahead software pipeline with sched_group_barrier 319)
    const unsigned char* bufp = lds + buf * C64_BRICK_B;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const unsigned char* bp = bufp + boff[dx];
#pragma unroll
      for (int hz = 0; hz < C64_HZ; ++hz) {
#pragma unroll
        for (int hy = 0; hy < C64_HY; ++hy) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(bp + (hz * C64_HY + hy) * C64_ROW_B);
#pragma unroll
          for (int zo = 0; zo < C64_TZ; ++zo) {
#pragma unroll
            for (int yo = 0; yo < C64_TY; ++yo) {
              const int dz = hz - zo, dy = hy - yo;   // compile-time after unrolling
              if (dz >= 0 && dz < 3 && dy >= 0 && dy < 3)
                acc[zo][yo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[(dz * 3 + dy) * 3 + dx], b, acc[zo][yo], 0, 0, 0);
            }
          }
        }
      }
    }
    const bool epi = kh == (it & 1);   // wave-uniform: this wave finishes the brick, its partner hands over its half
    if (!epi) {
#pragma unroll
      for (int zo = 0; zo < C64_TZ; ++zo)
#pragma unroll
        for (int yo = 0; yo < C64_TY; ++yo)
          *reinterpret_cast<f32x4*>(xch + ((ct * 8 + zo * C64_TY + yo) * 64 + lane) * 16) = acc[zo][yo];
    }
    __builtin_amdgcn_s_waitcnt(0);   // this wave's LDS-DMA pieces of brick it+1 have landed, its exchange stores are done
    __syncthreads();
    if (epi) {
      const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, tz = t / (p.tiles_x * p.tiles_y);
      const int z0 = tz * C64_TZ, y0 = ty * C64_TY, x = tx * C64_TX + v;
      bf16x4 av[C64_TZ][C64_TY];
      f32x4 bv[C64_TZ][C64_TY];
#pragma unroll
      for (int zo = 0; zo < C64_TZ; ++zo)
#pragma unroll
        for (int yo = 0; yo < C64_TY; ++yo) {
          const long vo = vox_n + ((long)(z0 + zo) * p.H + (y0 + yo)) * p.W + x;
          if (p.aux_mode) av[zo][yo] = *reinterpret_cast<const bf16x4*>(p.aux + vo * p.a_cs + p.a_co + c0);
          if (p.btab) {
            const int cls = vox_class(z0 + zo, y0 + yo, x, p.D, p.H, p.W);
            bv[zo][yo] = *reinterpret_cast<const f32x4*>(p.btab + ((long)(p.w_per_sample ? n : 0) * 64 + cls) * 64 + c0);
          }
        }
#pragma unroll
      for (int zo = 0; zo < C64_TZ; ++zo)
#pragma unroll
        for (int yo = 0; yo < C64_TY; ++yo) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(xch + ((ct * 8 + zo * C64_TY + yo) * 64 + lane) * 16);
          const long vo = vox_n + ((long)(z0 + zo) * p.H + (y0 + yo)) * p.W + x;
          float val[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            val[j] = acc[zo][yo][j] + o[j];
            if (p.btab) val[j] += bv[zo][yo][j];
            if (p.aux_mode == 1) val[j] += bf2f(av[zo][yo][j]);
            if (p.relu) val[j] = val[j] > 0.f ? val[j] : 0.f;
          }
          bf16x4 ob;
#pragma unroll
          for (int j = 0; j < 4; ++j) ob[j] = f2bf(val[j]);
          *reinterpret_cast<bf16x4*>(p.y + vo * p.y_cs + p.y_co + c0) = ob;
          if (p.stat_out) {   // of the STORED values, as a read-back pass would see them
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float r = bf2f(ob[j]);
              st_s[j] += r;
              st_q[j] += r * (p.aux_mode == 2 ? bf2f(av[zo][yo][j]) : r);
            }
          }
        }
    }
  }
  if (p.stat_out) {   // one partial per workgroup: lanes -> wave (fixed order), the pair of waves of a channel tile -> workgroup
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { st_s[j] += __shfl_xor(st_s[j], o, 64); st_q[j] += __shfl_xor(st_q[j], o, 64); }
    }
    __syncthreads();   // (every wave is past its last exchange read)
    if (v == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red[(wave * 16 + q * 4 + j) * 2] = st_s[j];
        red[(wave * 16 + q * 4 + j) * 2 + 1] = st_q[j];
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int c = tid >> 1, k = tid & 1, t4 = c >> 4, cc = c & 15;
      const float a = red[(t4 * 16 + cc) * 2 + k] + red[((t4 + 4) * 16 + cc) * 2 + k];
      p.stat_out[(((long)n * p.wgs_per_sample + wg) * 64 + c) * 2 + k] = a;
    }
  }
}

static bool c64_geometry_ok(const RtpAct* x, const RtpConvGeom* g, int transposed) {
  static const bool disabled = getenv("RTP_CONV64") && atoi(getenv("RTP_CONV64")) == 0;
  if (disabled) return false;
  if (g->ks != 3 || g->stride != 1 || g->pad != 1) return false;
  const int Ci = transposed ? (g->co + 31) / 32 * 32 : g->ci;
  const int Co = transposed ? g->ci : g->co;
  if (Ci != 64 || Co != 64) return false;
  if (g->di % C64_TZ || g->hi % C64_TY || g->wi % C64_TX) return false;
  if (x->c < 64 || (x->cs % 8) || (x->co % 8)) return false;
  return true;
}

/* Workgroups per sample of the launch rtp_conv64_try makes for this conv = statistics partials per sample it writes (0: not this
 * kernel's geometry).  One workgroup per CU when the batch divides 256 (half of them for the level-1 tensors), RtpConvGeom::wgs
 * (include/rtp.h) narrower or wider. */
int rtp_conv64_wgs(const RtpAct* x, const RtpConvGeom* g, int transposed) {
  if (!x || !g || !c64_geometry_ok(x, g, transposed)) return 0;
  const int tiles = (g->di / C64_TZ) * (g->hi / C64_TY) * (g->wi / C64_TX);
  // the level-1 tensors (fewer than 4096 bricks in all) on half the chip: phase config, same box, 256 / 128 / 64 workgroups:
  // 27.88-27.90 / 27.84-27.85 / 28.04-28.11 ms per step (alone the launch takes 43 / ~60 / 130 us: the side lanes want the CUs)
  static const int small_wgs = getenv("RTP_CONV64_WGS_SMALL") ? atoi(getenv("RTP_CONV64_WGS_SMALL")) : 128;
  int wgs = (g->wgs > 0 ? (g->wgs > 256 ? 256 : g->wgs) : ((long)tiles * g->n < 4096 ? small_wgs : 256)) / g->n;
  if (wgs < 1) wgs = 1;
  if (wgs > tiles) wgs = tiles;
  return wgs;
}

/* The conv as ONE launch of conv64_kernel on `wgs` = rtp_conv64_wgs(x, g, transposed) workgroups per sample (= the statistics
 * partials per sample the caller's stat_out holds: [n][wgs][64][2]).  RTP_OK, +1 if the geometry / options are not this kernel's, or a negative error. */
int rtp_conv64_try(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res, const RtpAct* y,
                   const RtpConvGeom* g, int relu, int transposed, int y_fp32, const RtpAct* stat_x, float* stat_out, int wgs,
                   hipStream_t s) {
  if (!x || !y || !g || !wf) return RTP_ERR_SHAPE;
  if (!c64_geometry_ok(x, g, transposed) || y_fp32 || (stat_x && res) || wgs < 1) return 1;
  if (y->c < 64) return RTP_ERR_SHAPE;
  if ((y->cs % 4) || (y->co % 4)) return RTP_ERR_ALIGN;
  if (rtp_multi_capture()) return RTP_ERR_UNSUPPORTED;
  const RtpAct* aux = stat_x ? stat_x : res;
  if (aux && (aux->c < 64 || (aux->cs % 4) || (aux->co % 4))) return RTP_ERR_ALIGN;
  if (stat_x && !stat_out) return RTP_ERR_SHAPE;
  C64Params p;
  p.x = (const bf16_t*)x->ptr; p.w = (const bf16_t*)wf; p.btab = btab; p.aux = aux ? (const bf16_t*)aux->ptr : nullptr;
  p.y = (bf16_t*)y->ptr; p.stat_out = stat_out;
  p.N = g->n; p.D = g->di; p.H = g->hi; p.W = g->wi;
  p.x_cs = x->cs; p.x_co = x->co; p.y_cs = y->cs; p.y_co = y->co; p.a_cs = aux ? aux->cs : 0; p.a_co = aux ? aux->co : 0;
  p.relu = relu; p.flip = transposed; p.w_per_sample = w_per_sample;
  p.aux_mode = stat_x ? 2 : (res ? 1 : 0);
  p.tiles_x = p.W / C64_TX; p.tiles_y = p.H / C64_TY;
  p.tiles_per_sample = (p.D / C64_TZ) * p.tiles_y * p.tiles_x;
  if (wgs > p.tiles_per_sample) return 1;   // (never at the shapes of the path: the slice kernels' bricks are twice as long)
  p.wgs_per_sample = wgs;
  if ((long)p.D * p.H * p.W * x->cs >= (1L << 31) || (long)p.tiles_per_sample * (wgs + 1) >= (1L << 31)) return RTP_ERR_SHAPE;
  static bool attr_done[RTP_MAX_DEVICES] = {};
  if (rtp_once_per_device(attr_done))
    (void)hipFuncSetAttribute((const void*)conv64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS_B);
  RtpProfScope prof(RTP_FAM_CONV_TILED, s);
  hipLaunchKernelGGL(conv64_kernel, dim3(p.N * wgs), dim3(512), C64_LDS_B, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
