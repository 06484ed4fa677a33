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
//   * W need not be a multiple of the brick: the last brick column's lanes beyond the volume stage zeros and store nothing (the
//     level-2 tensors of the native shape have 40 columns; level 3 -- 20 columns, 32 bricks in all -- stays on the generic kernel:
//     rtp_conv64_wgs);
//   * (the GroupNorm fold in this kernel's prologue, as conv_tiled.hip has it, was built and measured: 5.100-5.111 ms per hr3d step
//     against 5.087-5.104 with the fold launch in front -- every workgroup folding its sample's 442 KB of fp32 weights costs what the
//     launch saves; removed)
// Results: the same arithmetic as the slice chain up to fp32 summation order (one accumulator chain per input-channel half).
// Measured (B = 8, [16, 64, 160]): forward 281 us, data gradient 267 us = 1.03 / 1.09 PFLOP/s (the slice chain: 441 / 432 us);
// level 1 ([8, 32, 80]) 43 us on the whole chip (tools/microbench_conv64.py).
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

// Every operand is addressed in 32-channel BLOCKS, so that one kernel serves a plain 64 -> 64 layer (blocks = windows of one tensor /
// one weight image) and a PAIR of 32-channel layers that share the launch (rtp_conv64_blocks: the two head towers' first convs,
// center_head.py:86-93 -- forward: two weight images and two outputs over one input; data gradient: two output gradients and two
// weight images into one input gradient).
struct C64Params {
  const bf16_t* x[2]; int x_cs[2];             // input-channel halves (channel offset folded into the pointer)
  const bf16_t* w[2][2];                       // weight blocks [output half][input half]: 32 rows x 32 columns per tap
  int w_row_stride, w_tap_stride; long w_sample_stride;
  const float* bt[2]; int bt_cs;               // class-bias tables by output half: [sample | 1][64 classes][bt_cs]
  const bf16_t* aux[2]; int a_cs[2];           // residual / second statistics operand by output half
  bf16_t* y[2]; int y_cs[2];                   // output halves
  float* stat_out;                             // [n][wgs][64][2]
  float* acc;                                  // fp32 partial sums (n * voxels * 64 floats) of a chain over input-channel slices
  int acc_in, acc_out;                         // add them in the epilogue / write them INSTEAD of the finished output
  int N, D, H, W;
  int relu, flip, w_per_sample;
  int aux_mode;                 // 0 none, 1 residual (added), 2 second operand of the statistics (sum y * aux instead of sum y^2)
  int wgs_per_sample, tiles_x, tiles_y, tiles_per_sample;
};

__device__ __forceinline__ void c64_dma16(const bf16_t* src, unsigned lds_wave_base) {
  // (inline assembly: the compiler must not model the transfer -- see lds_dma16 in conv_tiled.hip)
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(src) : "memory", "m0");
}

// ACC: the launch is a link of a chain over input-channel slices (fp32 partial sums in and / or out)
template <bool ACC>
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

  // ---- this wave's share of the weights, into registers: rows (ct & 1) * 16 .. + 15 of block [ct >> 1][kh]
  bf16x8 wreg[27];
  {
    const bf16_t* wsrc = (ct >> 1 ? (kh ? p.w[1][1] : p.w[1][0]) : (kh ? p.w[0][1] : p.w[0][0])) + (p.w_per_sample ? (long)n * p.w_sample_stride : 0) + ((ct & 1) * 16 + v) * p.w_row_stride + q * 8;
#pragma unroll
    for (int t = 0; t < 27; ++t) wreg[t] = ld_bf16x8(wsrc + (p.flip ? 26 - t : t) * p.w_tap_stride);
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
              ((ck >> 2) << 6) | (hx << 8);   // bit 6: which input half the piece comes from; bits 8-12: the haloed x index
    s_rel[k] = ((hz * p.H + hy) * p.W + hx) * ((ck >> 2) ? p.x_cs[1] : p.x_cs[0]) + (ck & 3) * 8;
  }
  const long vox_n = (long)n * p.D * p.H * p.W;
  const bf16_t* xn0 = p.x[0] + vox_n * p.x_cs[0];
  const bf16_t* xn1 = p.x[1] + vox_n * p.x_cs[1];
  auto stage = [&](int t, int buf) {   // workgroup-uniform arguments
    const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, tz = t / (p.tiles_x * p.tiles_y);
    const int z0 = tz * C64_TZ, y0 = ty * C64_TY, x0 = tx * C64_TX;
    const int tflg = (z0 == 0) | ((z0 + C64_TZ == p.D) << 1) | ((y0 == 0) << 2) | ((y0 + C64_TY == p.H) << 3) | ((x0 == 0) << 4);
    const int xlim = (p.W - x0 + 1) << 8;   // a haloed x index >= this lies beyond the volume (W need not be a multiple of the brick)
    const int orgv = ((z0 - 1) * p.H + (y0 - 1)) * p.W + (x0 - 1);
    const int org0 = orgv * p.x_cs[0], org1 = orgv * p.x_cs[1];
#pragma unroll
    for (int k = 0; k < C64_ROUNDS; ++k) {
      if (k * 512 + (tid & ~63) < C64_ITEMS) {   // wave-uniform (3456 = 54 waves' worth)
        const bool oob = (s_pk[k] & tflg & 0x1f) || (s_pk[k] & 0x1f00) >= xlim;
        const bf16_t* src = oob ? g_zero_line_c64 : ((s_pk[k] & 64) ? xn1 + org1 : xn0 + org0) + s_rel[k];
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
  const int oh = ct >> 1, cl = c0 & 31;   // ... = channels cl .. cl + 3 of output half oh
  const bf16_t* const auxp = (oh ? p.aux[1] : p.aux[0]) + cl;
  const float* const btb = oh ? p.bt[1] : p.bt[0];
  const float* const btp = btb ? btb + (long)(p.w_per_sample ? n : 0) * 64 * p.bt_cs + cl : nullptr;
  bf16_t* const yp = (oh ? p.y[1] : p.y[0]) + cl;
  const int a_cs = oh ? p.a_cs[1] : p.a_cs[0], y_cs = oh ? p.y_cs[1] : p.y_cs[0];
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
      const bool inw = x < p.W;   // (the last brick column of a ragged W: its lanes beyond the volume load and store nothing)
      bf16x4 av[C64_TZ][C64_TY];
      f32x4 bv[C64_TZ][C64_TY];
      // The partial sums of a chain are private to its links (same geometry, same width, hence the same bricks on the same waves):
      // they are kept in the ACCUMULATORS' layout [sample][brick][channel tile][voxel tile][lane] x 16 B -- every access is one
      // contiguous kilobyte per wave (in voxel-major order a lane's four channels are a 16-byte piece of a 256-byte row).
      f32x4* const accp = ACC ? reinterpret_cast<f32x4*>(p.acc) + (((long)n * p.tiles_per_sample + t) * 4 + ct) * (8 * 64) + lane : nullptr;
      if (ACC && p.acc_out) {   // a link of a chain over input-channel slices: raw fp32 sums, nothing else
#pragma unroll
        for (int zo = 0; zo < C64_TZ; ++zo)
#pragma unroll
          for (int yo = 0; yo < C64_TY; ++yo) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(xch + ((ct * 8 + zo * C64_TY + yo) * 64 + lane) * 16);
            f32x4 r = acc[zo][yo] + o;
            if (p.acc_in) r += accp[(zo * C64_TY + yo) * 64];
            accp[(zo * C64_TY + yo) * 64] = r;
          }
        continue;
      }
#pragma unroll
      for (int zo = 0; zo < C64_TZ; ++zo)
#pragma unroll
        for (int yo = 0; yo < C64_TY; ++yo) {
          const long vo = vox_n + ((long)(z0 + zo) * p.H + (y0 + yo)) * p.W + x;
          if (p.aux_mode) av[zo][yo] = inw ? *reinterpret_cast<const bf16x4*>(auxp + vo * a_cs) : bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
          if (btp && inw) {
            const int cls = vox_class(z0 + zo, y0 + yo, x, p.D, p.H, p.W);
            bv[zo][yo] = *reinterpret_cast<const f32x4*>(btp + cls * p.bt_cs);
          }
          if (ACC && p.acc_in) {
            const f32x4 a4 = accp[(zo * C64_TY + yo) * 64];
            bv[zo][yo] = btp ? bv[zo][yo] + a4 : a4;
          }
        }
      const bool has_b = btp != nullptr || (ACC && p.acc_in);
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
            if (has_b) val[j] += bv[zo][yo][j];
            if (p.aux_mode == 1) val[j] += bf2f(av[zo][yo][j]);
            if (p.relu) val[j] = val[j] > 0.f ? val[j] : 0.f;
          }
          bf16x4 ob;
#pragma unroll
          for (int j = 0; j < 4; ++j) ob[j] = f2bf(val[j]);
          if (inw) *reinterpret_cast<bf16x4*>(yp + vo * y_cs) = ob;
          if (p.stat_out && inw) {   // of the STORED values, as a read-back pass would see them
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
  if (g->di % C64_TZ || g->hi % C64_TY || g->wi < 1) return false;   // (any W: the last brick column may be partly outside)
  if (x->c < 64 || (x->cs % 8) || (x->co % 8)) return false;
  return true;
}

/* Workgroups per sample of the launch rtp_conv64_try makes for this conv = statistics partials per sample it writes (0: not this
 * kernel's geometry).  One workgroup per CU when the batch divides 256 (half of them for the level-1 tensors), RtpConvGeom::wgs
 * (include/rtp.h) narrower or wider. */
int rtp_conv64_wgs(const RtpAct* x, const RtpConvGeom* g, int transposed) {
  if (!x || !g || !c64_geometry_ok(x, g, transposed)) return 0;
  const int tiles = (g->di / C64_TZ) * (g->hi / C64_TY) * ((g->wi + C64_TX - 1) / C64_TX);
  // a launch of this kernel costs ~16 us whatever it computes (512-thread workgroups, 27 weight fragments per lane, a staged brick
  // before the first MFMA): below 64 bricks in all -- the level-3 tensors of the native shape, [2, 8, 20] x 8 samples = 32 bricks,
  // 16.7 us here against 12.1 on the generic kernel -- the conv stays there (level 2, 192 bricks: 23.7 against 28.5)
  if ((long)tiles * g->n < 64) return 0;
  // the level-1 tensors (fewer than 4096 bricks in all) on half the chip: phase config, same box, 256 / 128 / 64 workgroups:
  // 27.88-27.90 / 27.84-27.85 / 28.04-28.11 ms per step (alone the launch takes 43 / ~60 / 130 us: the side lanes want the CUs)
  static const int small_wgs = 128;
  int wgs = (g->wgs > 0 ? (g->wgs > 256 ? 256 : g->wgs) : ((long)tiles * g->n < 4096 ? small_wgs : 256)) / g->n;
  if (wgs < 1) wgs = 1;
  if (wgs > tiles) wgs = tiles;
  return wgs;
}

static int c64_launch(C64Params& p, const RtpConvGeom* g, int wgs, hipStream_t s) {
  p.N = g->n; p.D = g->di; p.H = g->hi; p.W = g->wi;
  p.tiles_x = (p.W + C64_TX - 1) / C64_TX; p.tiles_y = p.H / C64_TY;
  p.tiles_per_sample = (p.D / C64_TZ) * p.tiles_y * p.tiles_x;
  if (wgs < 1 || wgs > p.tiles_per_sample) return RTP_ERR_SHAPE;
  p.wgs_per_sample = wgs;
  const long vox = (long)p.D * p.H * p.W;
  if (vox * p.x_cs[0] >= (1L << 31) || vox * p.x_cs[1] >= (1L << 31) || (long)p.tiles_per_sample * (wgs + 1) >= (1L << 31)) return RTP_ERR_SHAPE;
  static bool attr_done[RTP_MAX_DEVICES] = {};
  if (rtp_once_per_device(attr_done)) {
    (void)hipFuncSetAttribute((const void*)conv64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS_B);
    (void)hipFuncSetAttribute((const void*)conv64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS_B);
  }
  RtpProfScope prof(RTP_FAM_CONV64, s);
  if (p.acc_in || p.acc_out) hipLaunchKernelGGL(conv64_kernel<true>, dim3(p.N * wgs), dim3(512), C64_LDS_B, s, p);
  else hipLaunchKernelGGL(conv64_kernel<false>, dim3(p.N * wgs), dim3(512), C64_LDS_B, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

/* The conv as ONE launch of conv64_kernel on `wgs` = rtp_conv64_wgs(x, g, transposed) workgroups per sample (= the statistics
 * partials per sample the caller's stat_out holds: [n][wgs][64][2]).  RTP_OK, +1 if the geometry / options are not this kernel's,
 * or a negative error.  wf: [sample | 1][27][64][64] (rows = this launch's output channels). */
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
  for (int h = 0; h < 2; ++h) {
    p.x[h] = (const bf16_t*)x->ptr + x->co + 32 * h; p.x_cs[h] = x->cs;
    p.y[h] = (bf16_t*)y->ptr + y->co + 32 * h; p.y_cs[h] = y->cs;
    p.aux[h] = aux ? (const bf16_t*)aux->ptr + aux->co + 32 * h : nullptr; p.a_cs[h] = aux ? aux->cs : 0;
    p.bt[h] = btab ? btab + 32 * h : nullptr;
    for (int k = 0; k < 2; ++k) p.w[h][k] = (const bf16_t*)wf + (32 * h) * 64 + 32 * k;
  }
  p.bt_cs = 64; p.w_row_stride = 64; p.w_tap_stride = 64 * 64; p.w_sample_stride = 27L * 64 * 64;
  p.stat_out = stat_out; p.acc = nullptr; p.acc_in = p.acc_out = 0;
  p.relu = relu; p.flip = transposed; p.w_per_sample = w_per_sample;
  p.aux_mode = stat_x ? 2 : (res ? 1 : 0);
  return c64_launch(p, g, wgs, s);
}

/* include/rtp.h: rtp_conv64_blocks */
extern "C" int rtp_conv64_blocks(const RtpConv64* c, const RtpConvGeom* g, void* stream) {
  if (!c || !g) return RTP_ERR_SHAPE;
  // W % 16 == 0 as documented: the chain's fp32 partial sums live in BRICK layout (tiles * 8 192 floats per sample), which only
  // for full brick columns equals the documented n * voxels * 64 floats -- a ragged W would write past a buffer of that size
  if (g->ks != 3 || g->stride != 1 || g->pad != 1 || g->di % C64_TZ || g->hi % C64_TY || g->wi < C64_TX || g->wi % C64_TX || g->n < 1)
    return RTP_ERR_UNSUPPORTED;
  if (rtp_multi_capture()) return RTP_ERR_UNSUPPORTED;
  C64Params p;
  for (int h = 0; h < 2; ++h) {
    if (!c->x[h] || (c->x_cs[h] % 8) || ((uintptr_t)c->x[h] % 16)) return RTP_ERR_ALIGN;
    p.x[h] = (const bf16_t*)c->x[h]; p.x_cs[h] = c->x_cs[h];
    if (c->acc_out) { p.y[h] = nullptr; p.y_cs[h] = 0; }
    else {
      if (!c->y[h] || (c->y_cs[h] % 4) || ((uintptr_t)c->y[h] % 8)) return RTP_ERR_ALIGN;
      p.y[h] = (bf16_t*)c->y[h]; p.y_cs[h] = c->y_cs[h];
    }
    if (c->res[h] && ((c->r_cs[h] % 4) || ((uintptr_t)c->res[h] % 8))) return RTP_ERR_ALIGN;
    p.aux[h] = (const bf16_t*)c->res[h]; p.a_cs[h] = c->r_cs[h];
    p.bt[h] = c->btab[h];
    for (int k = 0; k < 2; ++k) {
      if (!c->w[h][k] || ((uintptr_t)c->w[h][k] % 16)) return RTP_ERR_ALIGN;
      p.w[h][k] = (const bf16_t*)c->w[h][k];
    }
  }
  if ((c->res[0] == nullptr) != (c->res[1] == nullptr)) return RTP_ERR_SHAPE;
  if ((c->w_row_stride % 8) || (c->w_tap_stride % 8) || (c->w_sample_stride % 8) || c->w_row_stride < 32) return RTP_ERR_ALIGN;
  if ((c->acc_in || c->acc_out) && (!c->acc || ((uintptr_t)c->acc % 16))) return RTP_ERR_SHAPE;
  if ((c->btab[0] || c->btab[1]) && (c->bt_cs < 32 || (c->bt_cs % 4))) return RTP_ERR_ALIGN;
  p.bt_cs = c->bt_cs; p.w_row_stride = c->w_row_stride; p.w_tap_stride = c->w_tap_stride; p.w_sample_stride = c->w_sample_stride;
  p.stat_out = nullptr; p.acc = c->acc; p.acc_in = c->acc_in; p.acc_out = c->acc_out;
  p.relu = c->relu; p.flip = c->transposed; p.w_per_sample = c->w_per_sample;
  p.aux_mode = c->res[0] ? 1 : 0;
  int wgs = (g->wgs > 0 ? (g->wgs > 256 ? 256 : g->wgs) : 256) / g->n;
  const int tiles = (g->di / C64_TZ) * (g->hi / C64_TY) * ((g->wi + C64_TX - 1) / C64_TX);
  if (wgs < 1) wgs = 1;
  if (wgs > tiles) wgs = tiles;
  return c64_launch(p, g, wgs, (hipStream_t)stream);
}
