// LDS-tiled weight-gradient correlation for the full-resolution 32-channel 3x3x3 stride-1 layers (the same layer set
// as conv_tiled.hip):  G[tap][co][ci] = sum_v gy[v][co] * x[v + tap - 1][ci].
//
// Same workgroup shape as the forward kernel: 8 waves = two teams of 4 that ping-pong between staging a brick
// (2z x 4y x 32x output voxels: gy brick 16 KB + haloed x brick 52 KB per team) and running MFMAs on it.
// The contraction runs over voxels, so both operands are read TRANSPOSED from the voxel-major LDS images with
// ds_read_b64_tr_b16; one k-step = one 32-voxel x-row of the brick, and a tap shift is just a different halo address.
// The 27 taps are dealt round-robin to the team's 4 waves (<= 7 taps x 4 accumulator tiles each); every wave keeps its
// accumulators in registers across ALL bricks of the team, so a workgroup emits ONE fp32 slab [27][32][32]
// (team 1's accumulators are folded into team 0's through LDS at the end).
// LDS images rotate the 16-B chunk index by (x>>2) -- conflict-free for the tr-read lane groups (voxels v and v+8).
#include <stdlib.h>

#include <string.h>

#include "rtp_common.h"
#include "rtp_multi.h"
#include "rtp_prof.h"

#define TZ 2
#define TY 4
#define TX 32
#define HZ (TZ + 2)
#define HY (TY + 2)
#define HX (TX + 2)
#define HALO_VOX (HZ * HY * HX)   // 816
#define BRICK_VOX (TZ * TY * TX)  // 256
#define X_ITEMS (HALO_VOX * 4)
#define G_ITEMS (BRICK_VOX * 4)
#define STAGE_ITERS 17            // 17 * 256 = 4352 >= 3264 + 1024 sixteen-byte items per team brick
// MFMA waves per workgroup (a multiple of 4) and tap slots per MFMA wave; four loader waves beside them.
// Round 4 measurements (in-kernel cycle stamps, tools/wgt_prof.py, B = 8 full resolution, per brick and workgroup):
//   4 MFMA waves (default): MFMA code 4 880 cycles for 224 MFMAs (3 584 of pipe time), loaders wait 2 500 at the barrier.  Ablation:
//     the MFMAs alone take 3 900, the 256 transposing LDS reads alone 4 094 -- a wave may have 15 LDS reads outstanding and under
//     load a read takes ~16 cycles of that window, so the read stream, not the matrix pipe, paces the loop.
//   8 MFMA waves (-DWG_NCW=8, 768 threads): 160 reads per wave -- the MFMA code drops to 3 600 cycles, the matrix pipe's own time --
//     but the workgroup now issues 1 280 reads per brick instead of 1 024 beside 68 KB of DMA writes, the LDS saturates, the
//     loaders' DMA issue stretches 2 150 -> 4 100 cycles and the MFMA waves wait 2 000 cycles per brick for data: 71 us against 60.
//     Loader waves at wave priority 3 change nothing (it is bandwidth, not issue arbitration).
//   one 16 x 16 tile per wave for all 27 taps, every haloed x fragment read once (640 reads per brick and workgroup, 160 per
//     wave): the right shape on paper, but hipcc's register allocator lets the 27 accumulator tiles rotate through the file
//     (three-address MFMA: the result lands in the dying operand's registers) and pays 54 copies + 20 spills per brick; with the
//     accumulator tied by inline assembly it splits every tile's live range instead (310 copies).  The attempt's
//     source is in the repository history (round 4, tools/ubench/) for a round with a hand-scheduled loop.
#ifndef WG_NCW
#define WG_NCW 4
#endif
#define WG_NTS ((27 + WG_NCW - 1) / WG_NCW)
#define WG_THREADS ((WG_NCW + 4) * 64)

struct WgTiledParams {
  const bf16_t* gy; const bf16_t* x; float* gp;
  int N, D, H, W, g_cs, g_co;
  int x_cs, x_co;   // x may be a 32-channel slice of a wider tensor
  int tiles_y, tiles_x, tiles_z, tiles_per_sample, wgs_per_sample;
  int dbg;  // timing experiments only (RTP_TILED_DBG): bit0 = consumers skip the MFMA work, bit1 = producers skip the DMA
  // optional: contract this workgroup's slab with the data-gradient weights wd[tap][ci][co] (bf16) -> qpart[n][wg][ci] =
  // sum_{tap,co} wd * slab, the slab's share of Q = sum_v dxhat*x of GroupNorm backward (norm_fold.hip, gn_bwd_coeffs_cls)
  const bf16_t* wd; float* qpart;
  // optional: sums of gy over the volume and its faces / edges / corners (27 "inclusive" subsets: per axis all | first plane
  // | last plane; slot = (az*3 + ay)*3 + ax), accumulated by the otherwise idle LOADER waves from the staged gy bricks and
  // stored as ONE partial table per workgroup tg[N][wgs_per_sample][27][32] (fixed summation order: reproducible).  They
  // give P = sum dxhat of GroupNorm backward, the bias gradient and the un-fold term without any pass over gy
  // (conv_tiled.hip, fused data gradient).
  float* tg;
  // slab geometry: a workgroup's slab is a 32 x 32 window (rows co, columns ci) of [27][slab_rows][slab_cols] floats --
  // dense [27][32][32] for a 32 -> 32 layer; for a wider conv run as (output slice, input slice) launches every launch fills
  // its window of the SAME [27][Co][Ci] slabs, which rtp_wgrad_fold then reads like the generic kernel's
  int slab_rows, slab_cols;
  // per-workgroup outputs (slab, Q partial, subset sums) go to slot n * part_stride + wg: the workgroups per sample of a plain
  // launch; in a shared launch (wgrad_tiled_multi_kernel) a problem runs on fewer and leaves the upper slots untouched (zero)
  int part_stride;
};
#define WG_NONE 0x3fffffff

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#ifdef RTP_WGT_PROF
// Cycle stamps of workgroup 0 (separate build, tools/wgt_prof.py): [0] consumer: MFMA code of a brick, [1] consumer: wait at the
// brick barrier, [2] bricks; [4] loader: DMA issue, [5] loader: subset sums, [6] loader: wait at the barrier, [7] iterations;
// [8] kernel cycles, [9] kernel time in 10-ns ticks
__device__ long long g_wgt_prof[16];
extern "C" int rtp_wgt_prof_read(long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wgt_prof), sizeof(long long) * 16) == hipSuccess ? 0 : -1;
}
#define WPROF_T(var) const long long var = __builtin_readcyclecounter()
#define WPROF_ADD(slot, a, b) wprof[slot] += (b) - (a)
#else
#define WPROF_T(var)
#define WPROF_ADD(slot, a, b)
#endif

__device__ __attribute__((aligned(16))) bf16_t g_zero_line[8];  // zero-initialised: source of padding voxels

__device__ __forceinline__ int rot(int chunk, int xi) { return ((chunk + (xi >> 2)) & 3) << 3; }  // bf16 elements

// Transposed fragment = two ds_read_b64_tr_b16 (voxels xq..xq+3 and xq+4..xq+7 of one x-row, 16 channels): lane gets
// channel (sub*16 + lane&15).  `lo`/`hi` are per-lane byte addresses that already contain the lane's voxel, chunk
// rotation and tap x-shift; OFF is a compile-time row offset, so it folds into the instruction's immediate field and
// ALL rows/taps share 12 (+4) address registers.
template <int OFF>
__device__ __forceinline__ bf16x8 tr_pair(unsigned lo, unsigned hi) {
#ifdef WGT_EXP_NOREAD   // timing experiment: no LDS reads (results wrong)
  s16x8 z = {(short)lo, (short)hi, 1, 2, 3, 4, 5, 6};
  return __builtin_bit_cast(bf16x8, z);
#endif
  s16x4 l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo + OFF));
  s16x4 h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi + OFF));
  s16x8 r = {l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
  return __builtin_bit_cast(bf16x8, r);
}

struct LaneAddr {          // 32-bit LDS byte addresses, already including this wave's tap shifts
  unsigned x[WG_NTS][2][2];  // [tap slot][ci sub-tile][lo/hi]  into the haloed x brick (row offset (dz,dy) and dx folded in)
  unsigned g[2][2];        // [co sub-tile][lo/hi]            into the gy brick
};

// One stage = one tap of one x-row: 4 transposing reads then 4 MFMAs.  The tap (wave-dependent) lives in the address
// registers, the row (compile-time) in the immediate offset, so all four waves run the SAME straight-line code.
template <int R, int T>
__device__ __forceinline__ void wg_tap(const LaneAddr& la, const bf16x8& a0, const bf16x8& a1, f32x4 (&acc)[WG_NTS][2][2]) {
  constexpr int rz = R / TY, ry = R % TY;
  constexpr int off = (rz * HY + ry) * HX * 64;  // bytes (< 64 KB): folds into the ds_read immediate
  const bf16x8 b0 = tr_pair<off>(la.x[T][0][0], la.x[T][0][1]);
  const bf16x8 b1 = tr_pair<off>(la.x[T][1][0], la.x[T][1][1]);
#ifdef WGT_EXP_NOMFMA   // timing experiment: the reads without the MFMAs (results wrong)
  asm volatile("" : : "v"(b0), "v"(b1), "v"(a0), "v"(a1));
  return;
#endif
  acc[T][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[T][0][0], 0, 0, 0);
  acc[T][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, acc[T][0][1], 0, 0, 0);
  acc[T][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, acc[T][1][0], 0, 0, 0);
  acc[T][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[T][1][1], 0, 0, 0);
}

template <int R, int T>
__device__ __forceinline__ void wg_taps(const LaneAddr& la, const bf16x8& a0, const bf16x8& a1, f32x4 (&acc)[WG_NTS][2][2]) {
  if constexpr (T < WG_NTS) {
    wg_tap<R, T>(la, a0, a1, acc);
    wg_taps<R, T + 1>(la, a0, a1, acc);
  }
}

template <int R>
__device__ __forceinline__ void wg_row(const LaneAddr& la, f32x4 (&acc)[WG_NTS][2][2]) {
  constexpr int goff = R * TX * 64;
  const bf16x8 a0 = tr_pair<goff>(la.g[0][0], la.g[0][1]);
  const bf16x8 a1 = tr_pair<goff>(la.g[1][0], la.g[1][1]);
  wg_taps<R, 0>(la, a0, a1, acc);
}

// Issue-order plan for the whole (fully unrolled, branch-free) brick: a software pipeline with the LDS reads of
// stage s+WG_DIST issued before the MFMAs of stage s.  Without it the scheduler front-loads hundreds of reads and spills.
#ifndef WG_DIST
#define WG_DIST 3  // prefetch distance in stages (8 fragment VGPRs each)
#endif
template <int S, int TOTAL>
__device__ __forceinline__ void wg_sched() {
  if constexpr (S < TOTAL) {
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                          // MFMA x4  (stage S)
    if constexpr (S + WG_DIST < TOTAL)
      __builtin_amdgcn_sched_group_barrier(0x100, ((S + WG_DIST) % WG_NTS == 0) ? 8 : 4, 0);         // DS_READ  (stage S+DIST)
    wg_sched<S + 1, TOTAL>();
  }
}

__device__ __forceinline__ void wg_brick(const LaneAddr& la, f32x4 (&acc)[WG_NTS][2][2]) {
  wg_row<0>(la, acc); wg_row<1>(la, acc); wg_row<2>(la, acc); wg_row<3>(la, acc);
  wg_row<4>(la, acc); wg_row<5>(la, acc); wg_row<6>(la, acc); wg_row<7>(la, acc);
  __builtin_amdgcn_sched_group_barrier(0x100, 8 + 4 * (WG_DIST - 1), 0);  // stages 0 .. DIST-1 (stage 0 includes the row's gy fragments)
  wg_sched<0, 8 * WG_NTS>();
}

// (explicit occupancy: with only __launch_bounds__ the scheduler aimed at 3 waves/SIMD once the loader path grew, and squeezed the
// consumers' software pipeline into 168 VGPRs: +9 us per launch)
__global__ __attribute__((amdgpu_flat_work_group_size(WG_THREADS, WG_THREADS), amdgpu_waves_per_eu(WG_THREADS / 256, WG_THREADS / 256))) void wgrad_tiled_kernel(WgTiledParams p);

__device__ __forceinline__ void wgrad_tiled_body(const WgTiledParams& p, const int n, const int wg) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef RTP_WGT_PROF
  long long wprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long wk0 = __builtin_readcyclecounter(), wr0 = __builtin_amdgcn_s_memrealtime();
#endif
  const bool loader = __builtin_amdgcn_readfirstlane(wave) >= WG_NCW;  // the last four waves stage, the others run the MFMAs
  const int tw = __builtin_amdgcn_readfirstlane(wave);                  // MFMA wave index (taps tw, tw + WG_NCW, ...)
  const int ttid = tid - WG_NCW * 64;                                   // loader thread index 0 .. 255
  const long vox_n = (long)n * p.D * p.H * p.W;
  constexpr int BUF = (HALO_VOX + BRICK_VOX) * 32;  // elements per staged brick (x halo + gy)
  // ---- which bricks: iteration k of the workgroup works on brick tile_k of its contiguous run (static deal, z fastest: L2 reuse of
  // shared z-planes); ctl[k & 3] = tile_k, published by the first loader thread before the barrier that ends iteration k - 1.
  int* ctl = reinterpret_cast<int*>(lds + 2 * BUF);
  const int t_begin = (int)((long)wg * p.tiles_per_sample / p.wgs_per_sample);
  const int t_end = (int)((long)(wg + 1) * p.tiles_per_sample / p.wgs_per_sample);
  const int first = t_begin < t_end ? t_begin : WG_NONE;

  if (loader) {
#ifndef WG_LOADER_PRIO
#define WG_LOADER_PRIO 0   // (experiments: wave priority of the loader waves)
#endif
    if (WG_LOADER_PRIO) __builtin_amdgcn_s_setprio(WG_LOADER_PRIO);
    // ---- producer: brick k -> buffer k&1, one brick ahead of the consumers.
    // Per-item descriptors are brick-independent and computed ONCE: element offset relative to the brick origin, and
    // six "touches the low/high face of the halo" bits; a brick then costs ~6 VALU + one LDS-DMA issue per item.
    // LDS-DMA (global_load_lds, 16 B per lane): a wave-instruction's LDS destination is linear (wave base + lane*16),
    // so the chunk rotation is applied on the SOURCE side -- the lane whose slot holds rotated chunk (i&3) fetches
    // logical chunk (i&3) - (x>>2).  Out-of-volume halo voxels fetch a zero line.
    int rel[STAGE_ITERS], flg[STAGE_ITERS];
#pragma unroll
    for (int it = 0; it < STAGE_ITERS; ++it) {
      const int i = ttid + it * 256;
      rel[it] = 0; flg[it] = 0;
      if (i < X_ITEMS) {
        const int cp = i & 3, hv = i >> 2;
        const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
        const int ck = (cp - (hx >> 2)) & 3;
        rel[it] = (((hz - 1) * p.H + (hy - 1)) * p.W + (hx - 1)) * p.x_cs + ck * 8;
        flg[it] = (hz == 0) | ((hz == HZ - 1) << 1) | ((hy == 0) << 2) | ((hy == HY - 1) << 3) | ((hx == 0) << 4) | (hx << 8);
      } else if (i < X_ITEMS + G_ITEMS) {
        const int j = i - X_ITEMS;
        const int cp = j & 3, bv = j >> 2;
        const int bx = bv % TX, by = (bv / TX) % TY, bz = bv / (TX * TY);
        const int ck = (cp - (bx >> 2)) & 3;
        rel[it] = ((bz * p.H + by) * p.W + bx) * p.g_cs + ck * 8;
        flg[it] = (bx + 1) << 8;  // same x-limit test as the halo items (halo index = brick index + 1)
      }
    }
    const bf16_t* xn = p.x + vox_n * p.x_cs + p.x_co;
    const bf16_t* gn = p.gy + vox_n * p.g_cs + p.g_co;
    // class-sum bookkeeping: thread = (x segment of 4 voxels, 8-channel chunk, brick row)
    const int seg = ttid & 7, c8 = (ttid >> 3) & 3, rr = ttid >> 5;
    float call[8], cedge[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) call[j] = cedge[j] = 0.f;
    // One [27][32] subset table PER LOADER WAVE in LDS (only allocated when p.tg: no cross-wave add order to vary from run to
    // run) + a counter: the last loader wave to finish adds the four tables in fixed order and stores the workgroup's partial.
    float* Tall = reinterpret_cast<float*>(lds + 2 * BUF) + 8;   // (behind the four brick words)
    float* T = Tall + (wave & 3) * 27 * 32;
    int* tcnt = reinterpret_cast<int*>(Tall + 4 * 27 * 32);
    if (p.tg) {
      for (int i = lane; i < 27 * 32; i += 64) T[i] = 0.f;
      if (ttid == 0) *tcnt = 0;
    }
    int prev_tile = WG_NONE;
    for (int k = 0;; ++k) {
      const int tile = k == 0 ? first : __builtin_amdgcn_readfirstlane(ctl[k & 3]);
      const bool have = tile < WG_NONE;
      WPROF_T(l0);
      if (have) {
        bf16_t* xL = lds + (k & 1) * BUF;
        const int tz = tile % p.tiles_z, tx = (tile / p.tiles_z) % p.tiles_x, ty = tile / (p.tiles_z * p.tiles_x);  // z fastest
        const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;
        const int org = (z0 * p.H + y0) * p.W + x0;  // brick origin voxel (scalar)
        const int tflg = (z0 == 0) | ((z0 + TZ == p.D) << 1) | ((y0 == 0) << 2) | ((y0 + TY == p.H) << 3) | ((x0 == 0) << 4);
        const int xlim = (p.W - x0 + 1) << 8;  // haloed x index >= this lies beyond the volume (W need not be a multiple of TX)
#pragma unroll
        for (int it = 0; it < STAGE_ITERS; ++it) {
          if (!(p.dbg & 2) && it * 256 + (ttid & ~63) < X_ITEMS + G_ITEMS) {  // wave-uniform: region sizes are multiples of 64 items
            const bool is_x = it * 256 + (ttid & ~63) < X_ITEMS;  // wave-uniform as well (3264 = 51 * 64)
            const bool oob = (flg[it] & tflg & 0xff) || flg[it] >= xlim;
            const bf16_t* src = oob ? g_zero_line : (is_x ? xn + (long)org * p.x_cs + rel[it] : gn + (long)org * p.g_cs + rel[it]);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xL + (it * 256 + (ttid & ~63)) * 8), 16, 0, 0);
          }
        }
      }
      WPROF_T(l1);
      WPROF_ADD(4, l0, l1);
      if (p.tg && k >= 1) {
        // sums of gy brick k-1 (landed before the previous barrier; the consumers are reading the same buffer now)
        const int tz = prev_tile % p.tiles_z, tx = (prev_tile / p.tiles_z) % p.tiles_x, ty = prev_tile / (p.tiles_z * p.tiles_x);
        const int z = tz * TZ + (rr >> 2), y = ty * TY + (rr & 3), x0 = tx * TX;
        const bf16_t* gb = lds + ((k - 1) & 1) * BUF + HALO_VOX * 32 + (rr * TX + seg * 4) * 32 + (((c8 + seg) & 3) << 3);
        bf16x8 v4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v4[j] = ld_bf16x8(gb + j * 32);
        float sv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[j] = (bf2f(v4[0][j]) + bf2f(v4[1][j])) + (bf2f(v4[2][j]) + bf2f(v4[3][j]));
        const int bxl = p.W - 1 - x0;   // brick position of the volume's last x (inside this brick iff 0 <= bxl < TX)
        const int ax = (seg == 0 && x0 == 0) ? 1 : ((bxl >= 0 && bxl < TX && seg == (bxl >> 2)) ? 2 : 0);
        float ev[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ev[j] = (ax == 1) ? bf2f(v4[0][j]) : (ax == 2) ? bf2f(v4[3][j]) : 0.f;   // W % 4 == 0: bxl & 3 == 3
#pragma unroll
        for (int j = 0; j < 8; ++j) { call[j] += sv[j]; cedge[j] += ev[j]; }
        const int az = (z == 0) ? 1 : (z == p.D - 1) ? 2 : 0, ay = (y == 0) ? 1 : (y == p.H - 1) ? 2 : 0;
        if (az | ay) {   // a row of a z / y face (15 % of the rows): also the face / edge subsets, straight into the LDS table
#pragma unroll
          for (int j = 0; j < 8; ++j) {   // fold the 8 x segments (lanes seg = lane & 7): every lane ends with the row sum
            sv[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv[j]), 0xB1, 0xf, 0xf, true));
            sv[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv[j]), 0x4E, 0xf, 0xf, true));
            sv[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv[j]), 0x141, 0xf, 0xf, true));
          }
#pragma unroll
          for (int cmb = 0; cmb < 3; ++cmb) {
            const int a = (cmb == 1) ? 0 : az, b = (cmb == 0) ? 0 : ay;   // (az,0), (0,ay), (az,ay)
            const bool on = (cmb == 0) ? (az != 0) : (cmb == 1) ? (ay != 0) : (az != 0 && ay != 0);
            if (on) {
              float* ts = T + ((a * 3 + b) * 3) * 32 + c8 * 8;
              if (seg == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(ts + j, sv[j]);
              }
              if (ax) {
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(ts + ax * 32 + j, ev[j]);
              }
            }
          }
        }
      }
      if (ttid == 0 && have) ctl[(k + 1) & 3] = tile + 1 < t_end ? tile + 1 : WG_NONE;   // publish brick k + 1
      prev_tile = tile;
      WPROF_T(l2);
      WPROF_ADD(5, l1, l2);
      __syncthreads();  // (drains the DMA: hipcc emits vmcnt(0) before the barrier)
      WPROF_T(l3);
      WPROF_ADD(6, l2, l3);
#ifdef RTP_WGT_PROF
      wprof[7] += 1;
#endif
      if (!have) break;
    }
#ifdef RTP_WGT_PROF
    if (blockIdx.x == 0 && tid == WG_NCW * 64) { for (int i = 4; i < 8; ++i) g_wgt_prof[i] = wprof[i]; }
#endif
    if (p.tg) {   // whole-volume subsets (all, all, all | first x | last x) kept in registers until now
      float* Tg = T;   // (this wave's LDS table; flushed below)
#pragma unroll
      for (int j = 0; j < 8; ++j) {   // call holds this thread's own 4-voxel partials: fold the 8 segments once, here
        call[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, call[j]), 0xB1, 0xf, 0xf, true));
        call[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, call[j]), 0x4E, 0xf, 0xf, true));
        call[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, call[j]), 0x141, 0xf, 0xf, true));
      }
      if (seg == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(Tg + c8 * 8 + j, call[j]);
      }
      const int axr = (seg == 0) ? 1 : 2;   // a thread only ever held one edge role (first x: seg 0; last x: seg 3 or 7)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (cedge[j] != 0.f) atomicAdd(Tg + axr * 32 + c8 * 8 + j, cedge[j]);
      // last loader wave out flushes (LDS atomics of a wave are complete once their lgkmcnt drains: the returning add waits)
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
      int last = 0;
      if (lane == 0) last = (atomicAdd(tcnt, 1) == 3);
      last = __builtin_amdgcn_readfirstlane(last);
      if (last) {
        float* G = p.tg + ((long)n * p.part_stride + wg) * 27 * 32;
        for (int k = lane; k < 27 * 32; k += 64)
          G[k] = (Tall[k] + Tall[27 * 32 + k]) + (Tall[2 * 27 * 32 + k] + Tall[3 * 27 * 32 + k]);
      }
    }
    return;
  }

  // ---- consumers: per-lane fragment addresses for both buffers, accumulators live for the whole kernel
  f32x4 acc[WG_NTS][2][2];
#pragma unroll
  for (int t = 0; t < WG_NTS; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  LaneAddr la;
  {
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) bf16_t*)lds;
    const int q = lane >> 4, i = lane & 15, a = i >> 2, pp = i & 3;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int chunk = 2 * sub + (pp >> 1), within = (pp & 1) * 4;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int xg = 8 * q + a + 4 * h;
        la.g[sub][h] = lds_base + 2u * (HALO_VOX * 32 + xg * 32 + rot(chunk, xg) + within);
#pragma unroll
        for (int t = 0; t < WG_NTS; ++t) {
          int tap = tw + WG_NCW * t;
          if (tap > 26) tap = 26;  // a slot beyond the 27 taps recomputes tap 26 into accumulators nobody stores
          const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
          const int xa = xg + dx;
          la.x[t][sub][h] = lds_base + 2u * ((dz * HY + dy) * HX * 32 + xa * 32 + rot(chunk, xa) + within);
        }
      }
    }
  }
  __syncthreads();  // brick 0 staged
  for (int k = 0;; ++k) {
    const int tile = k == 0 ? first : __builtin_amdgcn_readfirstlane(ctl[k & 3]);
    if (tile >= WG_NONE) break;
    WPROF_T(c0);
    if (!(p.dbg & 1)) wg_brick(la, acc);
#ifdef RTP_WGT_PROF
    __builtin_amdgcn_s_waitcnt(0);
#endif
    WPROF_T(c1);
    WPROF_ADD(0, c0, c1);
    // flip every fragment address to the other staging buffer (in place: no second address set stays live)
    const unsigned delta = (k & 1) ? (unsigned)(-(int)(2u * BUF)) : 2u * BUF;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        la.g[sub][h] += delta;
#pragma unroll
        for (int t = 0; t < WG_NTS; ++t) la.x[t][sub][h] += delta;
      }
    __syncthreads();  // brick k consumed, brick k+1 staged
    WPROF_T(c2);
    WPROF_ADD(1, c1, c2);
#ifdef RTP_WGT_PROF
    wprof[2] += 1;
#endif
  }
#ifdef RTP_WGT_PROF
  if (blockIdx.x == 0 && tid == 0) {
    for (int i = 0; i < 4; ++i) g_wgt_prof[i] = wprof[i];
    g_wgt_prof[8] = __builtin_readcyclecounter() - wk0;
    g_wgt_prof[9] = __builtin_amdgcn_s_memrealtime() - wr0;
  }
#endif

  const int q = lane >> 4, i = lane & 15;
  // ---- one fp32 slab [27][32][32] per workgroup; D[row = co][col = ci]: lane holds rows 4q..4q+3, column lane&15
  float* out = p.gp + ((long)n * p.part_stride + wg) * 27 * p.slab_rows * p.slab_cols;
#pragma unroll
  for (int t = 0; t < WG_NTS; ++t) {
    const int tap = tw + WG_NCW * t;
    if (tap < 27)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            out[(tap * p.slab_rows + a * 16 + q * 4 + r) * p.slab_cols + b * 16 + i] = acc[t][a][b][r];
  }
  if (p.qpart) {
    // lane holds slab[tap][co = a*16 + 4q + r][ci = b*16 + i]; wd[tap][ci][co..co+3] is one 8-byte read
    float qs[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < WG_NTS; ++t) {
      const int tap = tw + WG_NCW * t;
      if (tap < 27)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const bf16x4 w4 = *reinterpret_cast<const bf16x4*>(p.wd + ((long)tap * 32 + b * 16 + i) * 32 + a * 16 + q * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) qs[b] += bf2f(w4[r]) * acc[t][a][b][r];
          }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      qs[b] += __shfl_xor(qs[b], 16, 64);
      qs[b] += __shfl_xor(qs[b], 32, 64);
    }
    // fold the four consumer waves in fixed order through LDS (the staging buffers are idle now; the loader waves have
    // exited, which the hardware barrier accounts for)
    float* red = reinterpret_cast<float*>(lds);
    __syncthreads();
    if (q == 0) { red[tw * 32 + i] = qs[0]; red[tw * 32 + 16 + i] = qs[1]; }
    __syncthreads();
    if (tw == 0 && lane < 32)
    {
      float a = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < WG_NCW; ++w8) a += red[w8 * 32 + lane];
      p.qpart[((long)n * p.part_stride + wg) * 32 + lane] = a;
    }
  }
}

// One problem per launch; XCD-aware placement (see conv_tiled.hip): one contiguous run of bricks per XCD.
__global__ __attribute__((amdgpu_flat_work_group_size(WG_THREADS, WG_THREADS), amdgpu_waves_per_eu(WG_THREADS / 256, WG_THREADS / 256)))
void wgrad_tiled_kernel(WgTiledParams p) {
  const int bid = (gridDim.x % 8 == 0) ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int n = bid / p.wgs_per_sample;
  wgrad_tiled_body(p, n, bid - n * p.wgs_per_sample);
}

// Several weight gradients in one launch (rtp_multi.h, conv_tiled.hip's conv_tiled_multi_kernel): every sample's 256 / n workgroups
// (n = 8: an XCD's 32) split between the problems in proportion to their bricks.
struct WgMulti { const WgTiledParams* jobs; int njobs; int split[RTP_MULTI_MAX + 1]; };
__global__ __attribute__((amdgpu_flat_work_group_size(WG_THREADS, WG_THREADS), amdgpu_waves_per_eu(WG_THREADS / 256, WG_THREADS / 256)))
void wgrad_tiled_multi_kernel(WgMulti m) {
  const int per = m.split[m.njobs];   // workgroups per sample
  const int bid = (gridDim.x % 8 == 0) ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int n = bid / per, l = bid - n * per;
  int j = 0;
#pragma unroll
  for (int k = 1; k < RTP_MULTI_MAX; ++k) j += (k < m.njobs && l >= m.split[k]) ? 1 : 0;
  j = __builtin_amdgcn_readfirstlane(j);
  const WgTiledParams p = m.jobs[j];
  wgrad_tiled_body(p, n, l - m.split[j]);
}

// 32 -> <=32 channels: one launch.  Cin = 32 K, Cout = 32 J (K * J > 1, the feat64 backbone's layers): K x J launches, each
// filling its 32 x 32 window of the [27][Co][Ci] slabs.
static bool wg_tiled_applicable(const RtpConvGeom* g) {
  static const bool disabled = getenv("RTP_DISABLE_TILED") != nullptr;
  static const bool no_slices = false;
  if (disabled) return false;
  const int co32 = (g->co + 31) / 32 * 32;
  if (!(g->ks == 3 && g->stride == 1 && g->pad == 1 && g->di % TZ == 0 && g->hi % TY == 0 && g->wi % 16 == 0)) return false;
  if (g->ci == 32 && co32 == 32) return true;
  return !no_slices && g->ci % 32 == 0 && g->ci <= 256 && co32 <= 256 && g->co % 32 == 0;
}

static int wg_tiled_wgs(const RtpConvGeom* g) {
  const int tiles = (g->di / TZ) * (g->hi / TY) * ((g->wi + TX - 1) / TX);
  static const int total_wgs = 256;   // (a launch's own width: RtpConvGeom::wgs)
  static const int small_wgs = 128;   // (conv_tiled.hip: narrow launches for the lower levels)
  int wgs = ((small_wgs > 0 && (long)tiles * g->n < 2048) ? small_wgs : total_wgs) / g->n;
  if (g->wgs > 0) wgs = (g->wgs > 256 ? 256 : g->wgs) / g->n;   // RtpConvGeom::wgs: launch width = number of slabs (conv_tiled.hip)
  if (wgs < 1) wgs = 1;
  if (wgs > tiles) wgs = tiles;
  return wgs;
}

// Number of slabs rtp_wgrad will write per sample for this geometry if the caller lets it choose (0 = generic kernel,
// any nsplit accepted).
int rtp_wgrad_s2_nsplit(const RtpConvGeom* g);   // wgrad_s2_tiled.hip: the stride-2 convs
extern "C" int rtp_wgrad_nsplit(const RtpConvGeom* g) {
  if (!g) return 0;
  if (wg_tiled_applicable(g)) return wg_tiled_wgs(g);
  return rtp_wgrad_s2_nsplit(g);
}

int rtp_wgrad_tiled_try(const RtpAct* gy, const RtpAct* x, const RtpConvGeom* g, int nsplit, float* gp, hipStream_t s,
                        const void* wd, float* qpart, float* tg, int slab_rows, int slab_cols) {
  if (!wg_tiled_applicable(g)) return 1;
  if (x->cs % 32 || x->co % 8 || nsplit != wg_tiled_wgs(g)) return 1;   // x may be a 32-channel slice of a wider tensor
  const int K = g->ci / 32, J = (g->co + 31) / 32;
  if (K * J > 1) {   // channel slices: independent launches into windows of the wide slabs
    if (rtp_multi_capture()) return RTP_ERR_UNSUPPORTED;
    if (wd || qpart || tg) return RTP_ERR_UNSUPPORTED;
    if (gy->cs % 32 || gy->co % 8) return 1;
    RtpConvGeom gs = *g;
    gs.ci = 32; gs.co = 32; gs.w_ci_total = 0; gs.w_ci_off = 0;
    for (int j = 0; j < J; ++j)
      for (int k = 0; k < K; ++k) {
        RtpAct gj = *gy; gj.co = gy->co + 32 * j; gj.c = 32;
        RtpAct xk = *x; xk.co = x->co + 32 * k; xk.c = 32;
        const int rc = rtp_wgrad_tiled_try(&gj, &xk, &gs, nsplit, gp + (long)(32 * j) * g->ci + 32 * k, s, nullptr, nullptr, nullptr,
                                           32 * J, g->ci);
        if (rc != RTP_OK) return rc > 0 ? RTP_ERR_UNSUPPORTED : rc;
      }
    return RTP_OK;
  }
  WgTiledParams p;
  p.gy = (const bf16_t*)gy->ptr; p.x = (const bf16_t*)x->ptr; p.gp = gp;
  p.slab_rows = slab_rows > 0 ? slab_rows : 32; p.slab_cols = slab_cols > 0 ? slab_cols : 32;
  p.x_cs = x->cs; p.x_co = x->co;
  p.N = g->n; p.D = g->di; p.H = g->hi; p.W = g->wi; p.g_cs = gy->cs; p.g_co = gy->co;
  p.tiles_y = p.H / TY; p.tiles_x = (p.W + TX - 1) / TX; p.tiles_z = p.D / TZ;
  p.tiles_per_sample = (p.D / TZ) * p.tiles_y * p.tiles_x;
  p.wgs_per_sample = nsplit; p.part_stride = nsplit;
  #ifdef RTP_TILED_DBGFLAGS
  static const int dbg = getenv("RTP_TILED_DBG") ? atoi(getenv("RTP_TILED_DBG")) : 0;   // (phase-skipping experiment builds only)
#else
  static const int dbg = 0;
#endif
  p.dbg = dbg;
  p.wd = (const bf16_t*)wd; p.qpart = wd ? qpart : nullptr; p.tg = tg;
  const size_t shm_base = sizeof(bf16_t) * 2 * (size_t)(HALO_VOX + BRICK_VOX) * 32 + 32;
  const size_t shm = shm_base + (tg ? 4 * 27 * 32 * sizeof(float) + 16 : 0);
  if (std::vector<RtpMultiJob>* cap = rtp_multi_capture()) {   // recorded for a shared launch (rtp_multi.h), not issued
    RtpMultiJob job;
    job.kind = RTP_MULTI_WGRAD_TILED;
    job.variant = tg ? 1 : 0;   // (the subset-sum tables change the LDS size)
    job.tiles_per_sample = p.tiles_per_sample; job.n = p.N; job.slots_per_sample = nsplit; job.shm = shm;
    job.fam = RTP_FAM_WGRAD_TILED;
    job.params.assign((const char*)&p, (const char*)&p + sizeof(p));
    cap->push_back(job);
    return RTP_OK;
  }
  RtpProfScope prof(RTP_FAM_WGRAD_TILED, s);
  static bool attr[RTP_MAX_DEVICES] = {};
  if (rtp_once_per_device(attr)) {
    (void)hipFuncSetAttribute((const void*)wgrad_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(shm_base + 4 * 27 * 32 * sizeof(float) + 16));
  }
  hipLaunchKernelGGL(wgrad_tiled_kernel, dim3(p.N * p.wgs_per_sample), dim3(WG_THREADS), shm, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ---- shared launches (rtp_multi.h)
namespace {
struct WgMultiLauncher { WgMulti m; size_t shm; int n; };
}

int rtp_wgrad_tiled_multi_finish(std::vector<RtpMultiJob>& jobs, const int* share, void* dev_params, void** launcher) {
  if (hipFuncSetAttribute((const void*)wgrad_tiled_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)jobs[0].shm) != hipSuccess)
    return RTP_ERR_LAUNCH;
  std::vector<WgTiledParams> host(jobs.size());
  WgMultiLauncher* L = new WgMultiLauncher();
  L->shm = jobs[0].shm;
  L->m.jobs = (const WgTiledParams*)dev_params; L->m.njobs = (int)jobs.size(); L->n = jobs[0].n;
  int at = 0;
  for (size_t j = 0; j < jobs.size(); ++j) {
    if (jobs[j].params.size() != sizeof(WgTiledParams)) { delete L; return RTP_ERR_SHAPE; }
    memcpy(&host[j], jobs[j].params.data(), sizeof(WgTiledParams));
    host[j].wgs_per_sample = share[j];   // (part_stride keeps the slots the slab / partial buffers were sized for)
    L->m.split[j] = at;
    at += share[j];
  }
  for (size_t j = jobs.size(); j <= RTP_MULTI_MAX; ++j) L->m.split[j] = at;
  if (hipMemcpy(dev_params, host.data(), host.size() * sizeof(WgTiledParams), hipMemcpyHostToDevice) != hipSuccess) { delete L; return RTP_ERR_LAUNCH; }
  *launcher = L;
  return RTP_OK;
}

int rtp_wgrad_tiled_multi_launch(void* launcher, hipStream_t s) {
  WgMultiLauncher* L = (WgMultiLauncher*)launcher;
  hipLaunchKernelGGL(wgrad_tiled_multi_kernel, dim3(L->n * L->m.split[L->m.njobs]), dim3(WG_THREADS), L->shm, s, L->m);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

void rtp_wgrad_tiled_multi_drop(void* launcher) { delete (WgMultiLauncher*)launcher; }
