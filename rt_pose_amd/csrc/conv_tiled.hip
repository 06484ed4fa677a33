// LDS-tiled implicit-GEMM 3x3x3 stride-1 convolution for the full-resolution 32-channel layers of HRRadarPose
// (8 backbone convs + 2 head towers forward, the same set as data gradients: SURVEY.md 3.3 "hot loops").
//
// One persistent 8-wave workgroup per CU = two TEAMS of 4 waves that ping-pong: while one team runs MFMAs on its
// brick, the other stages its next brick global -> LDS, and they swap at every workgroup barrier.  Each SIMD thus
// always holds one computing wave and one loading wave (matrix pipe beside the memory pipe).
//   * the sample's folded weights [27][Cout][32] bf16 sit in LDS for the workgroup's lifetime (55 KB, shared);
//   * a team's brick is 2(z) x 4(y) x 32(x) output voxels; its haloed input 4 x 6 x 34 voxels x 64 B (52 KB) is staged
//     with coalesced 16-B loads (batched so all loads of a batch are in flight together), zero-filled outside the
//     volume -- padding happens AFTER the folded GroupNorm;
//   * wave w of a team owns z-plane (w>>1) and x-half (w&1): 4 y-rows x 16 voxels x Cout = 4 (x NT) accumulator tiles
//     of v_mfma_f32_16x16x32_bf16 (A = weights [16 cout][32 cin], B = voxels [32 cin][16]);
//   * sliding-window register reuse: for a fixed (dz,dx) every haloed input row is read from LDS ONCE and feeds the
//     (up to) three output rows it contributes to (dy = 0,1,2);
//   * both LDS images rotate the 16-B chunk index, chunk' = (chunk + 2*(x>>2)) & 3, which makes every ds_read_b128
//     lane group hit 16 distinct 16-B slots for any tap shift (derivation in DESIGN.md).
// Epilogue = generic kernel's: per-boundary-class bias, residual, ReLU, bf16 (or fp32) store.
#include "rtp_common.h"
#include "rtp_prof.h"

#define TZ 2
#define TY 4
#define TX 32
#define HZ (TZ + 2)
#define HY (TY + 2)
#define HX (TX + 2)
#define HALO_VOX (HZ * HY * HX)          // 816
#define HALO_ITEMS (HALO_VOX * 4)        // 16-B items
#define STAGE_BATCH 13
#define STAGE_ROUNDS 1                   // 13 * 256 = 3328 >= 3264: all of a thread's loads in flight at once

struct TiledParams {
  const bf16_t* x; const bf16_t* w; const float* btab; const bf16_t* res; void* y;
  int N, D, H, W, Co;  // Co = NT*16
  int y_cs, y_co, r_cs, r_co;
  int relu, y_fp32, flip, w_per_sample;
  int tiles_y, tiles_x, tiles_per_sample, teams_per_sample;
};

__device__ __forceinline__ int swz(int chunk, int xi) { return ((chunk + 2 * (xi >> 2)) & 3) << 3; }  // bf16 elements

template <int NT>
__global__ __launch_bounds__(512, 2) void conv_tiled_kernel(TiledParams p) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  bf16_t* wL = lds;                                   // [27][NT*16][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int team = wave >> 2, tw = wave & 3, ttid = tid & 255;
  bf16_t* xL = lds + 27 * NT * 16 * 32 + team * (HALO_VOX * 32);  // this team's [HZ][HY][HX][32]
  const int wgs_per_sample = p.teams_per_sample >> 1;
  const int n = blockIdx.x / wgs_per_sample;
  const int team_id = (blockIdx.x - n * wgs_per_sample) * 2 + team;  // team index within the sample
  const int v = lane & 15, q = lane >> 4;
  const int wz = tw >> 1, wx = tw & 1;

  // ---- weights -> LDS (once): item = (row = tap*Co + co, chunk)
  {
    const bf16_t* wsrc = p.w + (p.w_per_sample ? (long)n * 27 * p.Co * 32 : 0);
    const int items = 27 * p.Co * 4;
    for (int i0 = tid; i0 < items; i0 += 512 * 4) {
      bf16x8 val[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 512;
        if (i < items) val[k] = ld_bf16x8(wsrc + (long)(i >> 2) * 32 + (i & 3) * 8);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 512;
        if (i < items) {
          const int ck = i & 3, row = i >> 2;
          const int tap = row / p.Co, co = row - tap * p.Co;
          const int dtap = p.flip ? 26 - tap : tap;
          st_bf16x8(wL + (dtap * p.Co + co) * 32 + swz(ck, co), val[k]);
        }
      }
    }
  }

  const long vox_n = (long)n * p.D * p.H * p.W;
  const int my_tiles = (p.tiles_per_sample - team_id + p.teams_per_sample - 1) / p.teams_per_sample;
  int max_tiles = (p.tiles_per_sample + p.teams_per_sample - 1) / p.teams_per_sample;  // workgroup-uniform
  const int nphase = 2 * max_tiles + 1;
  int load_k = 0, comp_k = 0;

  for (int phase = 0; phase < nphase; ++phase) {
    const bool loading = ((phase + team) & 1) == 0;  // team-uniform (=> wave-uniform)
    if (loading) {
      if (load_k < my_tiles) {
        const int tile = team_id + load_k * p.teams_per_sample;
        const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, tz = tile / (p.tiles_x * p.tiles_y);
        const int z0 = tz * TZ - 1, y0 = ty * TY - 1, x0 = tx * TX - 1;
#pragma unroll
        for (int r = 0; r < STAGE_ROUNDS; ++r) {
          bf16x8 val[STAGE_BATCH];
          int dst[STAGE_BATCH];
#pragma unroll
          for (int k = 0; k < STAGE_BATCH; ++k) {
            const int i = ttid + (r * STAGE_BATCH + k) * 256;
            const int ck = i & 3, hv = i >> 2;
            const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
            const int gz = z0 + hz, gy = y0 + hy, gx = x0 + hx;
            dst[k] = (i < HALO_ITEMS) ? hv * 32 + swz(ck, hx) : -1;
            val[k] = zero_bf16x8();
            if (i < HALO_ITEMS && (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
              val[k] = ld_bf16x8(p.x + (vox_n + ((long)gz * p.H + gy) * p.W + gx) * 32 + ck * 8);
          }
#pragma unroll
          for (int k = 0; k < STAGE_BATCH; ++k)
            if (dst[k] >= 0) st_bf16x8(xL + dst[k], val[k]);
        }
      }
      ++load_k;
    } else if (comp_k < load_k && comp_k < my_tiles) {
      const int tile = team_id + comp_k * p.teams_per_sample;
      ++comp_k;
      const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, tz = tile / (p.tiles_x * p.tiles_y);
      const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;

      f32x4 acc[TY][NT];
#pragma unroll
      for (int t = 0; t < TY; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
      for (int dzx = 0; dzx < 9; ++dzx) {  // fully unrolled: lets the scheduler prefetch the next (dz,dx) fragments under the MFMAs
        const int dz = dzx / 3, dx = dzx - dz * 3;
        bf16x8 a[3][NT];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int co = nt * 16 + v;
            a[dy][nt] = ld_bf16x8(wL + (((dz * 3 + dy) * 3 + dx) * p.Co + co) * 32 + swz(q, co));
          }
        const int hx = wx * 16 + v + dx;
        const bf16_t* xrow = xL + ((wz + dz) * HY * HX + hx) * 32 + swz(q, hx);
#pragma unroll
        for (int ry = 0; ry < HY; ++ry) {
          const bf16x8 b = ld_bf16x8(xrow + ry * HX * 32);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int t = ry - dy;
            if (t >= 0 && t < TY) {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dy][nt], b, acc[t][nt], 0, 0, 0);
            }
          }
        }
      }

      // ---- epilogue
      const int oz = z0 + wz, ox = x0 + wx * 16 + v;
      const int czx = (oz == 0) | ((oz == p.D - 1) << 1) | ((ox == 0) << 4) | ((ox == p.W - 1) << 5);
#pragma unroll
      for (int t = 0; t < TY; ++t) {
        const int oy = y0 + t;
        const long vo = vox_n + ((long)oz * p.H + oy) * p.W + ox;
        const int cls = czx | ((oy == 0) << 2) | ((oy == p.H - 1) << 3);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int c0 = nt * 16 + q * 4;
          f32x4 val = acc[t][nt];
          if (p.btab) val += *reinterpret_cast<const f32x4*>(p.btab + ((long)(p.w_per_sample ? n : 0) * 64 + cls) * p.Co + c0);
          if (p.res) {
            const bf16x4 r = *reinterpret_cast<const bf16x4*>(p.res + vo * p.r_cs + p.r_co + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] += bf2f(r[j]);
          }
          if (p.relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] = val[j] > 0.f ? val[j] : 0.f;
          }
          if (p.y_fp32) {
            *reinterpret_cast<f32x4*>((float*)p.y + vo * p.y_cs + p.y_co + c0) = val;
          } else {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = f2bf(val[j]);
            *reinterpret_cast<bf16x4*>((bf16_t*)p.y + vo * p.y_cs + p.y_co + c0) = o;
          }
        }
      }
    }
    __syncthreads();
  }
}

// Returns RTP_OK if it handled the conv, +1 if the geometry is not this kernel's (caller falls through to the
// generic gather kernel), or a negative error.
int rtp_conv_tiled_try(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                       const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32, hipStream_t s) {
  static const bool disabled = getenv("RTP_DISABLE_TILED") != nullptr;
  if (disabled) return 1;
  if (g->ks != 3 || g->stride != 1 || g->pad != 1) return 1;
  const int Ci = transposed ? (g->co + 31) / 32 * 32 : g->ci;
  const int Co = transposed ? g->ci : g->co;
  if (Ci != 32 || (Co != 16 && Co != 32)) return 1;
  if (g->di % TZ || g->hi % TY || g->wi % TX) return 1;
  if (x->cs != 32 || x->co != 0) return 1;
  TiledParams p;
  p.x = (const bf16_t*)x->ptr; p.w = (const bf16_t*)wf; p.btab = btab;
  p.res = res ? (const bf16_t*)res->ptr : nullptr; p.y = y->ptr;
  p.N = g->n; p.D = g->di; p.H = g->hi; p.W = g->wi; p.Co = Co;
  p.y_cs = y->cs; p.y_co = y->co; p.r_cs = res ? res->cs : 0; p.r_co = res ? res->co : 0;
  p.relu = relu; p.y_fp32 = y_fp32; p.flip = transposed; p.w_per_sample = w_per_sample;
  p.tiles_y = p.H / TY; p.tiles_x = p.W / TX;
  p.tiles_per_sample = (p.D / TZ) * p.tiles_y * p.tiles_x;
  int wgs = 256 / p.N;  // workgroups per sample: one workgroup per CU when N divides 256
  if (wgs < 1) wgs = 1;
  if (wgs * 2 > p.tiles_per_sample) wgs = (p.tiles_per_sample + 1) / 2;
  p.teams_per_sample = wgs * 2;
  const int nt = Co / 16;
  const size_t shm = sizeof(bf16_t) * (27 * (size_t)Co * 32 + 2 * (size_t)HALO_VOX * 32);
  RtpProfScope prof(RTP_FAM_CONV_TILED, s);
  if (nt == 2) {
    static bool attr2 = false;
    if (!attr2) { (void)hipFuncSetAttribute((const void*)conv_tiled_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr2 = true; }
    hipLaunchKernelGGL(conv_tiled_kernel<2>, dim3(p.N * wgs), dim3(512), shm, s, p);
  } else {
    static bool attr1 = false;
    if (!attr1) { (void)hipFuncSetAttribute((const void*)conv_tiled_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr1 = true; }
    hipLaunchKernelGGL(conv_tiled_kernel<1>, dim3(p.N * wgs), dim3(512), shm, s, p);
  }
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
