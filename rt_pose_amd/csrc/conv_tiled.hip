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
// Epilogue = generic kernel's: per-boundary-class bias, residual, ReLU, bf16 (or fp32) store -- plus, on request, the
// per-channel statistics the next GroupNorm needs of the tensor just produced (forward: sum y, sum y^2; data gradient:
// P = sum dxhat, Q = sum dxhat * x), accumulated in registers over the workgroup's bricks and written as one partial per
// workgroup, which removes a separate read pass over the tensor.
#include <stdlib.h>
#include <string.h>

#include "rtp_common.h"
#include "rtp_multi.h"
#include "rtp_prof.h"

#define TZ 2
#ifndef RTP_TILED_TY
#define RTP_TILED_TY 4
#endif
#define TY RTP_TILED_TY
#define TX 32
#define HY (TY + 2)
#define HX (TX + 2)
#define PLANE_VOX (HY * HX)              // haloed voxels of one z-plane
#define PAIR_VOX (2 * PLANE_VOX)         // the staging unit is a PAIR of z-planes
#define PAIR_ITEMS (PAIR_VOX * 4)        // sixteen-byte items
// pairs a team keeps in LDS: the two pairs under the MFMAs; the pair the next brick is missing is requested in the team's load
// phase, into the slot of the pair the finished brick no longer needs.  (A three-slot ring at half the brick height, requested a
// phase earlier, was measured slower in round 2 and is gone.)
#define RING 2
#define HALO_VOX (RING * PAIR_VOX)       // 816 voxels x 64 B = 52 KB per team either way
#ifndef RTP_TILED_DIST
#define RTP_TILED_DIST 4
#endif
#ifndef RTP_TILED_WBUF
#define RTP_TILED_WBUF 2
#endif
#define STAGE_BATCH ((PAIR_ITEMS + 255) / 256)   // a thread's loads of one pair, all in flight at once

__device__ __attribute__((aligned(16))) bf16_t g_zero_line_c[8];
#ifdef RTP_TILED_PROF
// Cycle-counter instrumentation (separate build, tools/tiled_prof.sh): per wave of workgroup 0, s_memtime deltas summed over
// the phases: [0] compute-phase preamble, [1] MFMA loop, [2] barrier after compute, [3] load-phase issue, [4] epilogue,
// [5] barrier after load, [6] compute phases, [7] load phases
__device__ long long g_tiled_wg[512][2];   // per workgroup: entry and exit (s_memrealtime, 10-ns ticks)
extern "C" int rtp_tiled_prof_wgs(long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tiled_wg), sizeof(long long) * 1024) == hipSuccess ? 0 : -1;
}
__device__ long long g_tiled_prof[9][8];   // row 8 (wave 0): kernel entry -> weights staged -> phase loop -> loop end, realtime stamps
extern "C" int rtp_tiled_prof_read(long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tiled_prof), sizeof(long long) * 72) == hipSuccess ? 0 : -1;
}
#define PROF_T(var) const long long var = __builtin_readcyclecounter()
#define PROF_ADD(slot, a, b) prof_acc[slot] += (b) - (a)
#else
#define PROF_T(var)
#define PROF_ADD(slot, a, b)
#endif  // zero-initialised: source of padding voxels

// RTP_TILED_DBG (environment; phase skipping for timing experiments, results wrong) only exists in builds made with
// -DRTP_TILED_DBGFLAGS: in the product build the flag tests are compile-time false and cost no control flow.
#ifdef RTP_TILED_DBGFLAGS
#define RTP_DBG(bit) (p.dbg & (bit))
#else
#define RTP_DBG(bit) false
#endif

struct TiledParams {
  const bf16_t* x; const bf16_t* w; const float* btab; const bf16_t* res; void* y;  // res: residual (AUX 1) or the statistics' second operand (AUX 2)
  float* stat_out;  // [N][workgroups per sample][Co][2] or null
  int N, D, H, W, Co;  // Co = NT*16
  int y_cs, y_co, r_cs, r_co;
  int relu, y_fp32, flip, w_per_sample;
  int tiles_y, tiles_x, tiles_z, tiles_per_sample, teams_per_sample;
  int x_cs, x_co;          // x may be a 32-channel slice of a wider tensor (channel stride / first channel)
  const float* acc32; int a_cs;  // optional fp32 partial result [N][vox][a_cs] added before bias / ReLU (input-channel split)
  int dbg;  // timing experiments only (RTP_TILED_DBG): bit0 = skip the MFMA loop, bit1 = skip staging, bit2 = skip epilogue
  // The two teams synchronise only WITHIN themselves (LDS-counter barriers over their four waves) and drift freely against each
  // other (round 4; the lock-step schedule of rounds 2-3 -- roles swapped at a workgroup barrier per phase -- measured 25 % more
  // cycles and 1.3 % on the step, profiles/r04_tiled_prof.txt, and is gone, as is the opt-in dynamic brick claiming that lost its
  // A/B there).  prio: wave priority of a team's MFMA phase (experiments).
  int prio;
  // Per-workgroup partial outputs (statistics, totals) go to slot n * part_stride + wg: part_stride = workgroups per sample of a
  // plain launch; a launch that shares the grid with other problems (conv_tiled_multi_kernel) runs on fewer workgroups per sample
  // than the buffers were sized for and leaves the upper slots untouched (they are zero: nothing else writes them).
  int part_stride;
  // FUSE: a data gradient that writes the FINISHED gradient of its input tensor x (= p.res, the AUX operand):
  //   y = [x > 0] * (A0 * acc + Bt * x + Ct + sum_e Ae * ex_e)
  // coef[0] = this conv's GroupNorm-backward coefficients [N][32][3] (A, B, C) or null (1, 0, 0); ex_e = gradient terms of
  // x's OTHER consumers, already complete: plain addends (coef[1+e] null) or other GroupNorm consumers' dxhat with their
  // coefficients (A scales the term, B and C join Bt / Ct).  Replaces the separate fan-in pass over the tensor.
  const float* coef[4];
  const bf16_t* ex[3]; int ex_cs[3], ex_co[3];
  int nextra, mask;
  float* tot_out;   // FUSE, optional: per-channel sums of the stored output, one partial per workgroup [N][wgs][32]
  // FUSE, optional: this conv's GroupNorm-backward coefficients are computed HERE (every workgroup, in its prologue) instead
  // of by a kernel of their own between the weight gradient and this launch: Q = sum of the slab contractions qpart
  // [N][q_nsplit][32] (rtp_wgrad_q), P [N][32] from the class sums (rtp_gn_bwd_p); workgroup 0 of a sample also writes them
  // (coef_out [N*32*3] + dgamma/dbeta partials [N*32*2]) for the other consumers of the coefficients.
  const float* qpart; int q_nsplit; const float* gn_p; const float* gn_mr; const float* gn_gamma; int gn_groups; float gn_m;
  float* coef_out;
  // ... and P itself from the 27 inclusive subset sums of gy the weight-gradient kernel accumulated (tg [N][q_nsplit][27][32], see
  // wgrad_tiled.hip) when gn_p is null; workgroup 0 of a sample then also writes the per-boundary-class sums csum_out
  // [N][64][32] the deferred weight-gradient fold needs.
  const float* tg; float* csum_out;
  // FOLD (forward): the GroupNorm of the conv's input is folded in the prologue (rtp_conv_gn_fused): fp32 weights in tap-major
  // order [27][Co][32], the input's statistics partials [N][f_nsplit][32][2]; w / btab are then unused
  const float* fw; const float* fbias; const float* fgamma; const float* fbeta; const float* fstats;
  int f_nsplit, f_groups, f_co_real; float f_eps; float* f_mr;
  // SLICE: this launch is one (output-channel slice, input-channel slice) pair of a wider conv (rtp_conv_igemm_ws): the weights
  // are a 32 x 32 window of a wider image [sample][27][Co_total][Ci_total] (element strides), the class-bias table and the
  // statistics partials are 32-channel windows of tables that are bt_cs / st_cs channels wide.  Dense defaults otherwise.
  long w_sample_stride; int w_tap_stride, w_row_stride, bt_cs, st_cs;
};

// Barrier over the four waves of one team: a monotonic LDS counter (no reset, so no re-use hazard); `target` = 4 x the number of
// barriers the team has passed including this one.  s_waitcnt 0 first: the wave's LDS-DMA writes (vmcnt) and its LDS reads of the
// brick (lgkmcnt) must be complete before the team's other waves read / overwrite it.
__device__ __forceinline__ void team_sync(unsigned* cnt, unsigned target, int lane) {
  __builtin_amdgcn_s_waitcnt(0);
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
}

// One LDS-DMA piece (16 B per lane, wave-linear LDS destination from M0), issued from inline assembly so that the compiler does
// not model it: hipcc treats the builtin as a FLAT access to both memories, and while one is outstanding it (a) drains vmcnt
// before the next ds_read of the same address space and (b) turns every counted lgkmcnt wait into lgkmcnt(0) -- which is
// exactly what a brick requested under the previous brick's MFMA loop must not cost.  The price: the kernel waits for these
// itself (s_waitcnt 0 before the barrier that hands the brick to the MFMA phase).
__device__ __forceinline__ void lds_dma16(const bf16_t* src, unsigned lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(src) : "memory", "m0");
}

__device__ __forceinline__ int swz(int chunk, int xi) { return ((chunk + 2 * (xi >> 2)) & 3) << 3; }  // bf16 elements

// Issue-order plan (sched_group_barrier needs literal arguments, hence the recursion): per (dz,dx) tap group a wave issues
// R = 3*NT + HY ds_read_b128 (weights for 3 dy taps + HY haloed rows) and M = 12*NT MFMAs.  Group g+1's reads are
// interleaved one by one under group g's MFMAs; the last group's MFMAs run bare.
template <int NT>
__device__ __forceinline__ void cv_sched_prologue() { __builtin_amdgcn_sched_group_barrier(0x100, 3 * NT + HY, 0); }

template <int NT, int IDX>  // IDX enumerates (group 0..7) x (read slot 0..R-1)
__device__ __forceinline__ void cv_sched() {
  constexpr int R = 3 * NT + HY, M = 12 * NT;
  if constexpr (IDX < 8 * R) {
    constexpr int r = IDX % R;
    __builtin_amdgcn_sched_group_barrier(0x008, (r < M % R) ? M / R + 1 : M / R, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    cv_sched<NT, IDX + 1>();
  } else {
    __builtin_amdgcn_sched_group_barrier(0x008, M, 0);
  }
}

// AUX: 0 none, 1 residual added in the epilogue, 2 second operand of the statistics (not added).  STAT: emit statistics.
// FUSEX: 0 = plain epilogue; 1 + NEX = fused data-gradient epilogue with NEX (0..3) extra gradient terms (compile-time: the
// terms' prefetch registers exist only in the variant that needs them -- the kernel sits at the 256-VGPR limit).
// GEN: the launch also takes an fp32 partial sum in (acc32) and / or writes fp32 (y_fp32): the channel-slice chains and the head's
// logits.  Compile-time, so that the common bf16-in / bf16-out epilogue carries neither their branches nor the register copies
// the merge points cost.
template <int NT, bool HAS_BTAB, int AUX, bool STAT, int FUSEX = 0, bool GEN = false>
__device__ __forceinline__ void conv_tiled_body(const TiledParams& p, const int n, const int wg_in_sample) {
  constexpr bool FUSE = FUSEX > 0;
  static_assert(!(FUSE && GEN), "the fused data-gradient epilogue is bf16 in / bf16 out");
  constexpr int NEX = FUSE ? FUSEX - 1 : 0;
  static_assert(!FUSE || (NT == 2 && !HAS_BTAB && AUX == 2 && !STAT), "fused data-gradient epilogue: 32 channels, x in the AUX slot");
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  PROF_T(pk0);
#ifdef RTP_TILED_PROF
  const long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  bf16_t* wL = lds;                                   // [27][NT*16][32]
  // wave index through readfirstlane: everything derived from it (team, brick coordinates, ring slots, row bases) is then
  // wave-uniform to the compiler and lives on the scalar unit, beside the vector / matrix issue instead of in it
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, tw = wave & 3, ttid = tid & 255;
  bf16_t* xL = lds + 27 * NT * 16 * 32 + team * (HALO_VOX * 32);  // this team's [HZ][HY][HX][32]
  // class-bias table in LDS: 27 reachable classes (per axis: interior / first / last; every axis is >= 2 long here)
  float* bL = reinterpret_cast<float*>(lds + 27 * NT * 16 * 32 + 2 * (HALO_VOX * 32));  // [27][Co]
  const int wgs_per_sample = p.teams_per_sample >> 1;
  // (sample and workgroup-within-sample come from the kernel wrapper below: XCD-aware placement)
  const int bid = n * p.part_stride + wg_in_sample;             // this workgroup's slot in the per-workgroup partial outputs
  const int team_id = wg_in_sample * 2 + team;  // team index within the sample
  const int v = lane & 15, q = lane >> 4;
  const int wz = tw >> 1, wx = tw & 1;
  const bool claimer = tw == 0 && lane == 0;                                    // the one lane per team that publishes the team's run of bricks

  if (p.fw) {   // kernel-uniform: GroupNorm fold in the prologue (both teams' brick regions are scratch until the phase loop)
    float* scr = reinterpret_cast<float*>(lds + 27 * NT * 16 * 32);
    float* sc_l = scr;                                       // [32] gamma * rstd
    float* sh_l = scr + 32;                                  // [32] beta - mean * gamma * rstd
    float2* part = reinterpret_cast<float2*>(scr + 64);      // [512] statistics partials
    float* Tt = scr + 1088;                                  // [27][32]  T[tap][co] = sum_ci w * shift
    float* mrl = scr + 1952;                                 // [groups][2]
    // Every global load of the prologue is requested up front (one memory round trip for all of it): the statistics partials,
    // gamma / beta, and this thread's share of the fp32 weights in tap-major order [27][Co][32] (rtp_tail_desc_pack_wt) --
    // items (tap, output channel, 8-input-channel chunk) = two coalesced 16-byte loads each, in the order of the LDS image.
    constexpr int ITEMS = 27 * NT * 16 * 4, NIT = (ITEMS + 511) / 512;
    float wv[NIT][8];
#pragma unroll
    for (int m = 0; m < NIT; ++m) {
      const int i = tid + m * 512;
      const f32x4* wr = reinterpret_cast<const f32x4*>(p.fw + (long)(i < ITEMS ? i : 0) * 8);
      const f32x4 lo = wr[0], hi = wr[1];
#pragma unroll
      for (int j = 0; j < 4; ++j) { wv[m][j] = lo[j]; wv[m][4 + j] = hi[j]; }
    }
    float gam_r = 0.f, bet_r = 0.f;
    if (tid < 32) { gam_r = p.fgamma[tid]; bet_r = p.fbeta[tid]; }
    {
      const int c = tid & 31, pt = tid >> 5;
      float a0 = 0.f, a1 = 0.f;
      const float2* q = reinterpret_cast<const float2*>(p.fstats) + ((long)n * p.f_nsplit) * 32 + c;
      for (int s_ = pt; s_ < p.f_nsplit; s_ += 16) { const float2 v = q[(long)s_ * 32]; a0 += v.x; a1 += v.y; }
      part[tid] = make_float2(a0, a1);
    }
    __syncthreads();
    const int cg = 32 / p.f_groups;
    if (tid < p.f_groups) {   // fixed order, in double; reciprocal square root in fp32 (as rtp_fold_fwd)
      double s0 = 0.0, s1 = 0.0;
      for (int k = 0; k < 16; ++k)
        for (int j = 0; j < cg; ++j) { const float2 v = part[k * 32 + tid * cg + j]; s0 += v.x; s1 += v.y; }
      const double inv = 1.0 / ((double)cg * p.D * p.H * p.W);
      const double mean = s0 * inv;
      double var = s1 * inv - mean * mean;
      if (var < 0.0) var = 0.0;
      const float rstd = 1.0f / sqrtf((float)var + p.f_eps);
      mrl[tid * 2] = (float)mean;
      mrl[tid * 2 + 1] = rstd;
      if (p.f_mr && wg_in_sample == 0) {
        p.f_mr[((long)n * p.f_groups + tid) * 2] = (float)mean;
        p.f_mr[((long)n * p.f_groups + tid) * 2 + 1] = rstd;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const float sc = mrl[(tid / cg) * 2 + 1] * gam_r;
      sc_l[tid] = sc;
      sh_l[tid] = bet_r - mrl[(tid / cg) * 2] * sc;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NIT; ++m) {
      const int i = tid + m * 512;
      const int ck = i & 3, co = (i >> 2) % (NT * 16), tap = (i >> 2) / (NT * 16);
      bf16x8 o;
      float ts = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[j] = f2bf(wv[m][j] * sc_l[ck * 8 + j]);
        ts += wv[m][j] * sh_l[ck * 8 + j];
      }
      ts += __shfl_xor(ts, 1, 64);   // the four chunks of a row sit in adjacent lanes
      ts += __shfl_xor(ts, 2, 64);
      if (i < ITEMS) {
        const int dtap = p.flip ? 26 - tap : tap;
        const int arow = (NT == 2) ? ((co >> 2) & 1) * 16 + (co >> 3) * 4 + (co & 3) : co;
        st_bf16x8(wL + (dtap * p.Co + arow) * 32 + swz(ck, arow), o);
        if (ck == 0) Tt[tap * 32 + co] = ts;
      }
    }
    __syncthreads();
    for (int i = tid; i < 27 * p.Co; i += 512) {   // class bias: bias + the taps that stay inside the volume for that class
      const int co = i % p.Co, k = i / p.Co;
      const int cz = k / 9, cy = (k / 3) % 3, cx = k % 3;  // 0 interior, 1 first, 2 last
      float acc = (p.fbias && co < p.f_co_real) ? p.fbias[co] : 0.f;
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
        const bool ok = !(cz == 1 && kz == 0) && !(cz == 2 && kz == 2) && !(cy == 1 && ky == 0) && !(cy == 2 && ky == 2) &&
                        !(cx == 1 && kx == 0) && !(cx == 2 && kx == 2);
        acc += ok ? Tt[tap * 32 + co] : 0.f;
      }
      bL[i] = acc;
    }
    __syncthreads();   // the scratch is the bricks' from here on
  }
  // ---- weights -> LDS (once), by LDS-DMA: all of a thread's (up to 7) sixteen-byte pieces are in flight at once and land while
  // the set-up below runs (registers + two dependent rounds of loads and LDS stores cost 3.2 us per launch in front of it).  A wave
  // instruction's LDS destination is linear, so -- as for the bricks -- the layout is applied on the SOURCE side: the lane whose slot
  // is (tap position dtap, MFMA row arow, rotated chunk) fetches output channel co(arow) of tap dtap (flipped for a data gradient),
  // logical chunk (rotated chunk - 2 * (arow >> 2)) & 3.
  // MFMA row (nt, 4q+r) <- output channel 8q + 4nt + r: lane (voxel, q) then owns channels 8q..8q+7, one 16-B chunk, and a wave's
  // store instruction covers 16 voxels x 64 B = 1 KB of contiguous output.
  if (!p.fw) {
    const bf16_t* wsrc = p.w + (p.w_per_sample ? (long)n * p.w_sample_stride : 0);
    constexpr int CO_T = NT * 16, ITEMS = 27 * CO_T * 4, NPIECE = (ITEMS + 511) / 512;
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) {
      const int d = k * 512 + tid;
      if (d < ITEMS) {
        const int cpos = d & 3, row = d >> 2, dtap = row / CO_T, arow = row - dtap * CO_T;
        const int ck = (cpos - 2 * (arow >> 2)) & 3;
        const int co = (NT == 2) ? ((arow >> 2) & 3) * 8 + (arow >> 4) * 4 + (arow & 3) : arow;
        const int tap = p.flip ? 26 - dtap : dtap;
        const bf16_t* src = wsrc + (long)tap * p.w_tap_stride + (long)co * p.w_row_stride + ck * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(wL + (k * 512 + (tid & ~63)) * 8), 16, 0, 0);
      }
    }
  }

  if (HAS_BTAB && !p.fw) {
    for (int i = tid; i < 27 * p.Co; i += 512) {
      const int co = i % p.Co, k = i / p.Co;
      const int cz = k / 9, cy = (k / 3) % 3, cx = k % 3;  // 0 interior, 1 first, 2 last
      const int cls = (cz == 1) | ((cz == 2) << 1) | ((cy == 1) << 2) | ((cy == 2) << 3) | ((cx == 1) << 4) | ((cx == 2) << 5);
      bL[i] = p.btab[((long)(p.w_per_sample ? n : 0) * 64 + cls) * p.bt_cs + co];
    }
  }
  if constexpr (FUSE) {   // coefficient table in the (unused) class-bias region: [A0 | Bt | Ct | A1 | A2 | A3][32]
    float gq = 0.f, gp_ = 0.f, gmu = 0.f, gr = 0.f, ggam = 0.f;
    if (p.qpart && !p.gn_p) {   // P from the subset sums: scratch in team 0's (still unused) brick region
      float* Ts = reinterpret_cast<float*>(lds + 27 * NT * 16 * 32);   // [27][32] inclusive sums
      float* CSs = Ts + 27 * 32;                                        // [27 taps][32 co]
      float* Pp = CSs + 27 * 32;                                        // [16][32] partial P
      for (int i = tid; i < 27 * 32; i += 512) {   // the sample's workgroup partials, in fixed order, eight loads in flight
        const float* src = p.tg + (long)n * p.q_nsplit * 27 * 32 + i;
        float a = 0.f;
        int s_ = 0;
        for (; s_ + 8 <= p.q_nsplit; s_ += 8) {
          float g8[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) g8[k] = src[(long)(s_ + k) * 27 * 32];
#pragma unroll
          for (int k = 0; k < 8; ++k) a += g8[k];
        }
        for (; s_ < p.q_nsplit; ++s_) a += src[(long)s_ * 27 * 32];
        Ts[i] = a;
      }
      __syncthreads();
      // sum of gy over the voxels whose tap (kz,ky,kx) stays in bounds: per axis  all - [k == 0] first - [k == 2] last
      for (int i = tid; i < 27 * 32; i += 512) {
        const int tap = i >> 5, co = i & 31;
        const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
        float v = 0.f;
#pragma unroll
        for (int ia = 0; ia < 2; ++ia)
#pragma unroll
          for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int ic = 0; ic < 2; ++ic) {
              if ((ia && kz == 1) || (ib && ky == 1) || (ic && kx == 1)) continue;
              const int a = ia ? (kz == 0 ? 1 : 2) : 0, b = ib ? (ky == 0 ? 1 : 2) : 0, c = ic ? (kx == 0 ? 1 : 2) : 0;
              const float t = Ts[((a * 3 + b) * 3 + c) * 32 + co];
              v += ((ia + ib + ic) & 1) ? -t : t;
            }
        CSs[i] = v;
      }
      if (p.csum_out && wg_in_sample == 0) {   // exclusive boundary classes for the deferred fold: 64 x 32
        for (int i = tid; i < 64 * 32; i += 512) {
          const int cls = i >> 5, co = i & 31;
          const int sz = cls & 3, sy = (cls >> 2) & 3, sx = (cls >> 4) & 3;   // per axis: 0 interior, 1 first, 2 last, 3 both (empty)
          float v = 0.f;
          if (sz != 3 && sy != 3 && sx != 3) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
              for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                  // interior = all - first - last ; first = first ; last = last
                  const float ma = sz == 0 ? (a == 0 ? 1.f : -1.f) : (a == sz ? 1.f : 0.f);
                  const float mb = sy == 0 ? (b == 0 ? 1.f : -1.f) : (b == sy ? 1.f : 0.f);
                  const float mc = sx == 0 ? (c == 0 ? 1.f : -1.f) : (c == sx ? 1.f : 0.f);
                  v += ma * mb * mc * Ts[((a * 3 + b) * 3 + c) * 32 + co];
                }
          }
          p.csum_out[(long)n * 64 * 32 + i] = v;
        }
      }
      __syncthreads();
      {
        const int c = tid & 31, k = tid >> 5;   // 16 tap groups
        float pa = 0.f;
        for (int tap = k; tap < 27; tap += 16) {
          const bf16_t* wr = p.w + ((long)tap * 32 + c) * 32;
#pragma unroll
          for (int co = 0; co < 32; co += 8) {
            const bf16x8 w8 = ld_bf16x8(wr + co);
#pragma unroll
            for (int j = 0; j < 8; ++j) pa += bf2f(w8[j]) * CSs[tap * 32 + co + j];
          }
        }
        Pp[k * 32 + c] = pa;
      }
      __syncthreads();
      if (tid < 32) {
#pragma unroll
        for (int k = 0; k < 16; ++k) gp_ += Pp[k * 32 + tid];
      }
      __syncthreads();   // the brick region is staged into from here on
    }
    if (p.qpart) {   // workgroup-uniform
      float* scr = bL + 448;   // [256] partial Q, then [2][32] group-sum operands at +256
      if (tid < 256) {
        const int c = tid & 31, k = tid >> 5;
        float q = 0.f;
        for (int s_ = k; s_ < p.q_nsplit; s_ += 8) q += p.qpart[((long)n * p.q_nsplit + s_) * 32 + c];
        scr[tid] = q;
      }
      __syncthreads();
      if (tid < 32) {
        const int cg = 32 / p.gn_groups, g = tid / cg;
#pragma unroll
        for (int k = 0; k < 8; ++k) gq += scr[k * 32 + tid];
        if (p.gn_p) gp_ = p.gn_p[(long)n * 32 + tid];
        gmu = p.gn_mr[((long)n * p.gn_groups + g) * 2];
        gr = p.gn_mr[((long)n * p.gn_groups + g) * 2 + 1];
        ggam = p.gn_gamma[tid];
        scr[256 + tid] = ggam * gp_;
        scr[288 + tid] = ggam * gr * (gq - gmu * gp_);
      }
      __syncthreads();
    }
    if (tid < 32) {
      float a0 = 1.f, bt = 0.f, ct = 0.f;
      if (p.coef[0]) { const float* k = p.coef[0] + ((long)n * 32 + tid) * 3; a0 = k[0]; bt = k[1]; ct = k[2]; }
      if (p.qpart) {
        const float* scr = bL + 448;
        const int cg = 32 / p.gn_groups, g0 = (tid / cg) * cg;
        float s1 = 0.f, s2 = 0.f;
        for (int j = g0; j < g0 + cg; ++j) { s1 += scr[256 + j]; s2 += scr[288 + j]; }
        a0 = gr * ggam;
        bt = -gr * gr * s2 / p.gn_m;
        ct = -gr * s1 / p.gn_m + gr * gr * gmu * s2 / p.gn_m;
        if (p.coef_out && wg_in_sample == 0) {
          float* o = p.coef_out + ((long)n * 32 + tid) * 3;
          o[0] = a0; o[1] = bt; o[2] = ct;
          float* pt = p.coef_out + (long)p.N * 32 * 3 + ((long)n * 32 + tid) * 2;
          pt[0] = gr * (gq - gmu * gp_);
          pt[1] = gp_;
        }
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        float ae = 1.f;
        if (e < NEX && p.coef[1 + e]) { const float* k = p.coef[1 + e] + ((long)n * 32 + tid) * 3; ae = k[0]; bt += k[1]; ct += k[2]; }
        bL[(3 + e) * 32 + tid] = ae;
      }
      bL[tid] = a0; bL[32 + tid] = bt; bL[64 + tid] = ct;
    }
    if (tid < 256) bL[192 + tid] = 0.f;   // per-wave running totals [8 waves][32]
  }
  PROF_T(pk1);
  // ---- work distribution: a static deal.  A team works through ONE contiguous, balanced run of its sample's bricks (z fastest:
  // consecutive bricks share two of their four haloed z-planes).  ctl[2 * team + {0, 1}] = the run [first, end) (first = -1: none);
  // ctl[12 + team] = the team's barrier counter.
  int* ctl = reinterpret_cast<int*>(bL + 27 * p.Co);
  if (claimer) {
    // (32-bit: team index x bricks of a sample stays far below 2^31 -- rtp_conv_tiled_try checks -- and a 64-bit division is
    // ~150 instructions on this chip)
    const int t_begin = (int)((unsigned)team_id * (unsigned)p.tiles_per_sample / (unsigned)p.teams_per_sample);
    const int t_end = (int)((unsigned)(team_id + 1) * (unsigned)p.tiles_per_sample / (unsigned)p.teams_per_sample);
    ctl[2 * team] = t_end > t_begin ? t_begin : -1;
    ctl[2 * team + 1] = t_end;
  }
  if (tid < 2) ctl[12 + tid] = 0;   // the two team-barrier counters
  __syncthreads();   // weights, tables and the teams' runs are in place for both teams
  int both_done = 0;
  unsigned* tcnt = reinterpret_cast<unsigned*>(ctl + 12 + team);
  unsigned tbar = 0;
  // A team barrier per phase; a team leaves after a load phase that staged nothing (its pending epilogue has run there).
#define PHASE_SYNC() do { \
    if (loading && !staged) both_done = 1; \
    tbar += 4; team_sync(tcnt, tbar, lane); \
  } while (0)
  const long vox_n = (long)n * p.D * p.H * p.W;
  // Staging descriptors (brick-independent, computed once): element offset of each of this thread's 13 sixteen-byte
  // items relative to the brick origin, its swizzled LDS slot, and six "on the low/high face of the halo" bits.
  // LDS-DMA staging (global_load_lds, 16 B per lane, no VGPRs for the data): a wave instruction's LDS destination is
  // linear (wave base + lane * 16), so the chunk rotation is applied on the SOURCE side -- the lane whose slot holds
  // rotated chunk cp of haloed voxel hv fetches logical chunk (cp - 2*(hx>>2)) & 3; out-of-volume voxels fetch a zero line.
  // s_pk packs the six face bits (bits 0-5) and the haloed x index (8-13).
  int s_rel[STAGE_BATCH], s_pk[STAGE_BATCH];
#pragma unroll
  for (int k = 0; k < STAGE_BATCH; ++k) {
    const int i = ttid + k * 256;
    const int cp = i & 3, hv = i >> 2;
    const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);   // hz: 0 / 1 = first / second plane of the pair
    const int ck = (cp - 2 * (hx >> 2)) & 3;
    s_pk[k] = (hz == 0) | ((hz == 1) << 1) | ((hy == 0) << 2) | ((hy == HY - 1) << 3) | ((hx == 0) << 4) | (hx << 8);
    s_rel[k] = ((hz * p.H + (hy - 1)) * p.W + (hx - 1)) * p.x_cs + ck * 8;
  }
  const bf16_t* xn = p.x + vox_n * p.x_cs + p.x_co;
  // the team's current unit [u_cur, u_end) (u_cur = its next brick to stage), whether a brick is staged and waiting for its MFMA
  // phase, and whether the team has been told there is no further unit -- all team-uniform
  int u_cur = 0, u_end = 0;
  bool staged = false, finished = false;
  // The team's LDS region is a FIFO ring of RING plane pairs.  A brick (2 output z-planes) reads planes z0-1 .. z0+2 = the pair
  // (z0-1, z0) and the pair (z0+1, z0+2): along a z-column (bricks are dealt z fastest) the upper pair of one brick is the lower
  // pair of the next, so a brick inside a column stages ONE pair (26 KB) and only a column's first brick stages two.  The pair a
  // brick is missing is requested at the START of the previous brick's MFMA phase, into the ring slot that phase does not read,
  // so its HBM latency hides under that whole phase (the second pair of a column's first brick follows in the load phase, into
  // the slot the finished brick frees).  A brick always computes from the last two pairs staged: slots cur-2, cur-1.
  int ring_cur = 0;
  // coordinates of the team's next brick (z fastest), set at the start of a unit and advanced brick by brick
  int c_tz = 0, c_tx = 0, c_ty = 0;
  auto issue_pair = [&](int tz, int tx, int ty, int upper) {   // team-uniform arguments
    const int zp = tz * TZ - 1 + 2 * upper, y0 = ty * TY, x0 = tx * TX;
    const int org = ((zp * p.H + y0) * p.W + x0) * p.x_cs;
    const int tflg = (zp < 0) | ((zp + 1 >= p.D) << 1) | ((y0 == 0) << 2) | ((y0 + TY == p.H) << 3) | ((x0 == 0) << 4);
    const int xlim = (p.W - x0 + 1) << 8;  // haloed x index >= this lies beyond the volume (W need not be a multiple of TX)
    const int ring_slot = ring_cur;
    ring_cur = ring_cur == RING - 1 ? 0 : ring_cur + 1;
    if (RTP_DBG(2)) return;
#pragma unroll
    for (int k = 0; k < STAGE_BATCH; ++k) {
      if (k * 256 + ttid < PAIR_ITEMS) {  // (TY 4: the last piece is half a wave -- EXEC masks the other lanes' transfers)
        const bool oob = (s_pk[k] & tflg & 0xff) || s_pk[k] >= xlim;
        const bf16_t* src = oob ? g_zero_line_c : xn + org + s_rel[k];
        // issued and awaited within one load phase: the compiler's own bookkeeping (counted vmcnt for the residual) is right
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(xL + ring_slot * (PAIR_VOX * 32) + (k * 256 + (ttid & ~63)) * 8), 16, 0, 0);
      }
    }
  };
  // (a brick stages both pairs when it is the team's first or the first of a z-column)
  // State of the brick whose MFMAs are done but whose epilogue is still pending: the epilogue (bias, residual, ReLU,
  // rounding, stores, statistics) runs at the start of the team's NEXT load phase, right after that phase's global loads
  // have been issued -- it then overlaps both those loads' latency and the OTHER team's MFMA phase, instead of sitting
  // between two MFMA phases with the matrix pipe idle (measured: MFMA loop alone 52 us, whole kernel 80 us).
  constexpr int CH = 4 * NT;  // channels this lane owns: [c0, c0 + CH)
  const int c0 = q * CH;
  // Every global address of the epilogue = a wave-uniform 64-bit ROW base (scalar unit) + this lane's constant 32-bit BYTE offset:
  // the loads and stores take the scalar-base form (global_* v_off, data, s[base]) and no vector instruction is spent on
  // addresses inside the row loop (they were 5 per access: two 32-bit multiplies, a 64-bit multiply-add, an add3, a shift-add).
  const unsigned res_lane_b = 2u * (unsigned)(v * p.r_cs + c0);
  const unsigned y_lane_b = ((GEN && p.y_fp32) ? 4u : 2u) * (unsigned)(v * p.y_cs + c0);
  const unsigned a_lane_b = 4u * (unsigned)(v * p.a_cs + c0);
  unsigned ex_lane_b[NEX > 0 ? NEX : 1];
#pragma unroll
  for (int e = 0; e < (NEX > 0 ? NEX : 1); ++e) ex_lane_b[e] = NEX > 0 ? 2u * (unsigned)(v * p.ex_cs[e] + c0) : 0u;
  // The residual joins the accumulators through ONE MFMA per tile with an identity-like A operand (32 -> 32 channels): the
  // prefetched residual row (voxel v, channels 8q .. 8q+7) IS a B fragment, and A[m][k] = 1 where k is the channel that MFMA row m
  // of tile nt stands for (row 4q'+r <- channel 8q' + 4nt + r, the weight image's row permutation), so D += residual exactly
  // (1.0 x bf16 in fp32) -- 8 MFMAs per brick instead of 32 unpacks + 32 adds in the epilogue.
  const float relu_lo = p.relu ? 0.f : -__builtin_inff();
  bf16x8 idf[NT];
  if constexpr (AUX == 1 && NT == 2) {
    const bool on = (lane >> 4) == ((lane & 15) >> 2);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 8; ++j) idf[nt][j] = (bf16_t)((on && j == 4 * nt + (lane & 3)) ? 1.0f : 0.0f);
  }
  f32x4 acc[TY][NT];
  bf16x8 pre_r8[TY];
  bf16x4 pre_r4[TY];
  // MFMA-phase operands that do not depend on the staged brick's DATA are prepared at the END of the team's load phase, which
  // waits ~3 000 cycles at the barrier for the other team's MFMAs anyway (in-kernel stamps: load phase = 1 200 issue + 1 100
  // epilogue + 3 000 waiting; MFMA phase = 1 100-1 450 preamble + 4 400 loop): accumulators initialised with the class bias,
  // the nine row-fragment base addresses, the first tap group's weight fragments.
  constexpr int WBUF = RTP_TILED_WBUF, COT = NT * 16;
  typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) bf16_t*)lds;
  const unsigned a_base = lds0 + 2u * (v * 32 + swz(q, v));  // swz(q, nt*16+v) == swz(q, v): 8*nt == 0 mod 4
  bf16x8 fa[WBUF][3][NT];   // weights: (dz,dx) tap groups in registers, fetched a group (or two) ahead
  unsigned b_base[3][3];    // [dz][dx]: plane wz + dz of the brick's four = pair (wz + dz) >> 1, plane (wz + dz) & 1 of it
  auto load_a = [&](int g, bf16x8 (&a)[3][NT]) {
    const int dz = g / 3, dx = g - dz * 3;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        a[dy][nt] = *(lds_frag)(a_base + 2u * ((((dz * 3 + dy) * 3 + dx) * COT + nt * 16) * 32));
  };
  bool pend = false;
  int e_oz = 0, e_xb = 0, e_y0 = 0;   // pending brick: plane, first x of this wave's 16-voxel column, first row (all wave-uniform)
  float st_p[4 * NT], st_q[4 * NT];  // running per-lane statistics of this lane's 4*NT channels (STAT)
#pragma unroll
  for (int j = 0; j < 4 * NT; ++j) st_p[j] = st_q[j] = 0.f;

#ifdef RTP_TILED_PROF
  long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  PROF_T(pk2);
#endif
  // Every compute phase of a team is followed by a load phase of the same team (its epilogue runs there).
  for (int phase = 0; !both_done; ++phase) {
    const bool loading = (phase & 1) == 0;  // team-uniform (=> wave-uniform)
    PROF_T(pt0);
    if (loading) {
      bf16x8 exr[NEX > 0 ? NEX : 1][TY];
      if constexpr (NEX > 0) {
        // the other consumers' terms of the pending brick: requested first, so they arrive under the DMA issue below
        if (pend) {
#pragma unroll
          for (int e = 0; e < NEX; ++e) {
#pragma unroll
            for (int t = 0; t < TY; ++t)
              exr[e][t] = ld_bf16x8(reinterpret_cast<const bf16_t*>(
                  reinterpret_cast<const char*>(p.ex[e] + (vox_n + ((long)e_oz * p.H + (e_y0 + t)) * p.W + e_xb) * p.ex_cs[e] + p.ex_co[e]) + ex_lane_b[e]));
          }
        }
      }
      if (!finished) {
        bool fresh = false;   // first brick of a unit: nothing of it is in the ring yet
        if (u_cur == u_end) {
          const int s0 = __builtin_amdgcn_readfirstlane(ctl[2 * team]);
          if (s0 < 0) {
            finished = true;
          } else {
            u_cur = s0;
            u_end = __builtin_amdgcn_readfirstlane(ctl[2 * team + 1]);
            c_tz = s0 % p.tiles_z; c_tx = (s0 / p.tiles_z) % p.tiles_x; c_ty = s0 / (p.tiles_z * p.tiles_x);
            fresh = true;
          }
        }
        if (!finished) {   // stage brick u_cur = (c_tz, c_tx, c_ty); both pairs when it opens a unit or a z-column
          if (fresh || c_tz == 0) issue_pair(c_tz, c_tx, c_ty, 0);
          issue_pair(c_tz, c_tx, c_ty, 1);
          staged = true;
        }
      }
      PROF_T(pt1);
      PROF_ADD(3, pt0, pt1);
      if (pend) {
        pend = false;
        // ---- epilogue of the previous brick: bias + residual + ReLU in fp32, one rounding, one 16-B store per lane and row
        const int oz = e_oz, xb = e_xb, y0 = e_y0;
        float tsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float ka[CH], kb[CH], kc[CH], ke[NEX > 0 ? NEX : 1][CH];
        if constexpr (FUSE) {
#pragma unroll
          for (int k = 0; k < CH; k += 4) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(bL + c0 + k), b4 = *reinterpret_cast<const f32x4*>(bL + 32 + c0 + k),
                        c4 = *reinterpret_cast<const f32x4*>(bL + 64 + c0 + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) { ka[k + j] = a4[j]; kb[k + j] = b4[j]; kc[k + j] = c4[j]; }
#pragma unroll
            for (int e = 0; e < NEX; ++e) {
              const f32x4 e4 = *reinterpret_cast<const f32x4*>(bL + (3 + e) * 32 + c0 + k);
#pragma unroll
              for (int j = 0; j < 4; ++j) ke[e][k + j] = e4[j];
            }
          }
        }
#pragma unroll
        for (int t = 0; t < TY; ++t) {
          const int oy = y0 + t;
          const long rs = vox_n + ((long)oz * p.H + oy) * p.W + xb;   // first voxel of this wave's row segment (scalar)
          float ev[CH];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) ev[nt * 4 + j] = acc[t][nt][j];   // (the class bias is already in: accumulator init)
          if constexpr (GEN) {
            if (p.acc32) {   // kernel-uniform
              const float* ap = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.acc32 + rs * p.a_cs) + a_lane_b);
#pragma unroll
              for (int k = 0; k < CH; k += 4) {
                const f32x4 aa = *reinterpret_cast<const f32x4*>(ap + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) ev[k + j] += aa[j];
              }
            }
          }
          float aux[CH];
          if constexpr (AUX == 2 || (AUX == 1 && NT == 1)) {
            if constexpr (NT == 2) {
#pragma unroll
              for (int j = 0; j < 8; ++j) aux[j] = bf2f(pre_r8[t][j]);
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) aux[j] = bf2f(pre_r4[t][j]);
            }
          }
          if constexpr (FUSE) {
#pragma unroll
            for (int j = 0; j < CH; ++j) ev[j] = ev[j] * ka[j] + (kb[j] * aux[j] + kc[j]);
#pragma unroll
            for (int e = 0; e < NEX; ++e) {
#pragma unroll
              for (int j = 0; j < CH; ++j) ev[j] += ke[e][j] * bf2f(exr[e][t][j]);
            }
            if (p.mask) {
#pragma unroll
              for (int j = 0; j < CH; ++j) ev[j] = aux[j] > 0.f ? ev[j] : 0.f;
            }
          }
          if constexpr (AUX == 1 && NT == 1) {   // (32 -> 32 channels: the residual went in through the matrix pipe)
#pragma unroll
            for (int j = 0; j < CH; ++j) ev[j] += aux[j];
          }
          if constexpr (!FUSE) {   // ReLU without a branch: max with 0 or with -inf (a branch costs a register copy per value at its merge)
#pragma unroll
            for (int j = 0; j < CH; ++j) asm("v_max_f32 %0, %1, %2" : "=v"(ev[j]) : "s"(relu_lo), "v"(ev[j]));
          }
          bool stored = false;
          if constexpr (GEN) {
            if (p.y_fp32) {
              float* yp = reinterpret_cast<float*>(reinterpret_cast<char*>((float*)p.y + rs * p.y_cs + p.y_co) + y_lane_b);
#pragma unroll
              for (int k = 0; k < CH; k += 4) *reinterpret_cast<f32x4*>(yp + k) = f32x4{ev[k], ev[k + 1], ev[k + 2], ev[k + 3]};
              stored = true;
            }
          }
          if (!stored) {
            bf16_t* yp = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>((bf16_t*)p.y + rs * p.y_cs + p.y_co) + y_lane_b);
            if constexpr (NT == 2) {
              bf16x8 o;
#pragma unroll
              for (int j = 0; j < 8; ++j) o[j] = f2bf(ev[j]);
#ifdef RTP_EXP_NO_NT
              st_bf16x8(yp, o);
#else
              st_bf16x8_nt(yp, o);
#endif
              if constexpr (FUSE) {
#pragma unroll
                for (int j = 0; j < 8; ++j) tsum[j] += bf2f(o[j]);   // totals of the stored (rounded) values
              }
              if constexpr (STAT) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  const float r = bf2f(o[j]);  // statistics of the stored (rounded) values, as a read-back pass would see
                  st_p[j] += r;
                  st_q[j] += r * (AUX == 2 ? aux[j] : r);
                }
              }
            } else {
              bf16x4 o;
#pragma unroll
              for (int j = 0; j < 4; ++j) o[j] = f2bf(ev[j]);
              *reinterpret_cast<bf16x4*>(yp) = o;
              if constexpr (STAT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const float r = bf2f(o[j]);
                  st_p[j] += r;
                  st_q[j] += r * (AUX == 2 ? aux[j] : r);
                }
              }
            }
          }
        }
        if constexpr (FUSE) {
          if (p.tot_out) {   // fold the 16 voxel lanes (DPP row shifts: lane 15 of a row ends up with the row's sum), then
                             // this wave's own LDS slots (no atomics: fixed order)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              tsum[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tsum[j]), 0x111, 0xf, 0xf, true));
              tsum[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tsum[j]), 0x112, 0xf, 0xf, true));
              tsum[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tsum[j]), 0x114, 0xf, 0xf, true));
              tsum[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tsum[j]), 0x118, 0xf, 0xf, true));
            }
            if (v == 15) {
              float* tp = bL + 192 + wave * 32 + c0;
#pragma unroll
              for (int k = 0; k < 8; k += 4) {
                f32x4 a4 = *reinterpret_cast<const f32x4*>(tp + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) a4[j] += tsum[k + j];
                *reinterpret_cast<f32x4*>(tp + k) = a4;
              }
            }
          }
        }
      }
      {   // ---- operands of the brick just staged that do not need its data (see the declarations above).  UNCONDITIONAL: were
          // it skipped when nothing was staged, the old values would have to survive the whole loop body (the compiler cannot
          // know that an MFMA phase always follows a load phase that staged) -- 33 registers held across the epilogue; computed
          // from whatever the coordinates are, the few wasted instructions of the last phases read valid LDS addresses.
        const int z0 = c_tz * TZ, y0 = c_ty * TY, x0 = c_tx * TX;
        const int slot_a = (ring_cur + RING - 2) % RING, slot_b = (ring_cur + RING - 1) % RING;   // the last two pairs staged
        {
          // Accumulators start from the CLASS BIAS of their output voxel (LDS table, 27 classes x Co floats): the epilogue adds
          // nothing.  The class is (z class, x class) of the lane's voxel -- a per-lane row of the table -- plus the y class of the
          // row, which only the brick's first / last row can have (a scalar select for t = 0 and t = TY - 1).
          const int oz = z0 + wz, xb = x0 + wx * 16, ox = xb + v;
          if constexpr (HAS_BTAB) {
            typedef const __attribute__((address_space(3))) f32x4* lds_f4;
            const int kzx = ((oz == 0) ? 1 : (oz == p.D - 1) ? 2 : 0) * 9 + ((ox == 0) ? 1 : (ox == p.W - 1) ? 2 : 0);
            const unsigned bb = (unsigned)(size_t)(__attribute__((address_space(3))) float*)bL + 4u * (unsigned)(kzx * p.Co + c0);
#pragma unroll
            for (int t = 0; t < TY; ++t) {
              const int oy = y0 + t;
              const unsigned ycls = (t == 0 && oy == 0) ? 3u : (t == TY - 1 && oy == p.H - 1) ? 6u : 0u;   // (H >= 2: never both)
              const unsigned rb = bb + ycls * (4u * (unsigned)p.Co);
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[t][nt] = *(lds_f4)(rb + 16u * nt);
            }
          } else {
#pragma unroll
            for (int t = 0; t < TY; ++t)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
          }

#pragma unroll
          for (int dz = 0; dz < 3; ++dz) {
            const int j = wz + dz;
            const int pv = ((j >> 1) ? slot_b : slot_a) * PAIR_VOX + (j & 1) * PLANE_VOX;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const int hx = wx * 16 + v + dx;
              b_base[dz][dx] = lds0 + 2u * (unsigned)((xL - lds) + (pv + hx) * 32 + swz(q, hx));
              // opaque VGPR: left visible, hipcc keeps the wave-uniform slot part in an SGPR and spends a v_add per fragment read
              // instead of folding the row offset into the ds_read's immediate
              asm volatile("" : "+v"(b_base[dz][dx]));
            }
          }
          load_a(0, fa[0]);
          if (WBUF == 3) load_a(1, fa[1]);
        }
      }
#ifdef RTP_TILED_PROF
      PROF_T(pt2);
      PROF_ADD(4, pt1, pt2);
      PHASE_SYNC();
      PROF_T(pt3);
      PROF_ADD(5, pt2, pt3);
      prof_acc[7] += 1;
      continue;
#endif
    } else if (staged) {
      staged = false;
      const int z0 = c_tz * TZ, y0 = c_ty * TY, x0 = c_tx * TX;
      if (++c_tz == p.tiles_z) { c_tz = 0; if (++c_tx == p.tiles_x) { c_tx = 0; ++c_ty; } }
      const bool unit_ends = ++u_cur == u_end;   // the run's last brick: "no further brick" is published (LDS) at this phase's end
      const bool live = x0 + wx * 16 < p.W;  // wave-uniform: otherwise this wave's 16-voxel column is padding (never the claimer's: wx = 0).
                                             // No early `continue`: every extra path through the loop body costs register copies where
                                             // the paths meet (32 accumulator + 16 residual registers were being shuffled per phase).

      const int oz = z0 + wz, xb = x0 + wx * 16;   // (accumulators, fragment bases, first weights: prepared in the load phase)
      // The residual is fetched NOW, under the MFMA loop: loaded after it, its HBM latency sat on the critical path of
      // every brick (measured: epilogue 78 us of a 132 us launch).  (Not in the load phase: its barrier drains every outstanding
      // vector-memory operation of the wave, so the rows' HBM latency would sit in front of it.)
      // address = wave-uniform 64-bit row base (scalar unit) + this lane's constant 32-bit byte offset
      const bf16_t* res_row = p.res + (vox_n + ((long)oz * p.H + y0) * p.W + (live ? xb : 0)) * p.r_cs + p.r_co;   // (padding wave: any in-bounds row)
#pragma unroll
      for (int t = 0; t < TY; ++t) {
        pre_r8[t] = zero_bf16x8();
        pre_r4[t] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        if constexpr (AUX != 0) {  // compile-time: a runtime select per load makes hipcc branch around each one and wait vmcnt(0) early
          const bf16_t* rp = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(res_row + (long)t * p.W * p.r_cs) + res_lane_b);
          if constexpr (NT == 2) pre_r8[t] = ld_bf16x8(rp);
          else pre_r4[t] = *reinterpret_cast<const bf16x4*>(rp);
        }
      }
      PROF_T(pc1);
      PROF_ADD(0, pt0, pc1);
      if (p.prio) __builtin_amdgcn_s_setprio(2);
      if (live && !RTP_DBG(1)) {
        // Software pipeline over 54 steps = 9 (dz,dx) tap groups x HY haloed rows.  Step s reads its row fragment three
        // steps early into a 4-deep register ring, and a group's 3*NT weight fragments are read during the previous
        // group (double-buffered); sched_barrier fences pin that order (left alone, hipcc issues each ds_read right
        // before its first use and waits lgkmcnt(0) on it, exposing the LDS latency to the single MFMA wave per SIMD).
        constexpr int DIST = RTP_TILED_DIST, FBN = 8, WB = WBUF, NSTEP = 9 * HY;
        bf16x8 fb[FBN];
        auto row_frag = [&](int s) {
          const int g = s / HY, ry = s - g * HY, dz = g / 3, dx = g - dz * 3;
          return *(lds_frag)(b_base[dz][dx] + 2u * (ry * HX * 32));
        };
#pragma unroll
        for (int s = 0; s < DIST; ++s) fb[s % FBN] = row_frag(s);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
          const int g = s / HY, ry = s - g * HY;
#ifndef RTP_EXP_NOLDS
          if (s + DIST < NSTEP) fb[(s + DIST) % FBN] = row_frag(s + DIST);
          if (ry == 0 && g + WB - 1 < 9) load_a(g + WB - 1, fa[(g + WB - 1) % WB]);
#endif
#ifndef RTP_EXP_NOFENCE
          __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int t = ry - dy;
            if (t >= 0 && t < TY) {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[g % WB][dy][nt], fb[s % FBN], acc[t][nt], 0, 0, 0);
            }
          }
#ifndef RTP_EXP_NOFENCE
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      }
      if (p.prio) __builtin_amdgcn_s_setprio(0);
      if (AUX == 1 && NT == 2 && live) {   // + residual, through the matrix pipe (its rows have had the whole loop to arrive)
#pragma unroll
        for (int t = 0; t < TY; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(idf[nt], pre_r8[t], acc[t][nt], 0, 0, 0);
      }
      if (unit_ends && claimer) ctl[2 * team] = -1;
#ifdef RTP_TILED_PROF
      __builtin_amdgcn_s_waitcnt(0);   // (MFMA results are not covered by a counter: the delta below is issue time)
      PROF_T(pc2);
      PROF_ADD(1, pc1, pc2);
      if (live && !RTP_DBG(4)) { pend = true; e_oz = oz; e_xb = xb; e_y0 = y0; }
      PHASE_SYNC();
      PROF_T(pc3);
      PROF_ADD(2, pc2, pc3);
      prof_acc[6] += 1;
      continue;
#endif
      if (live && !RTP_DBG(4)) { pend = true; e_oz = oz; e_xb = xb; e_y0 = y0; }
    }
    PHASE_SYNC();
  }
#ifdef RTP_TILED_PROF
  if (tid == 0 && blockIdx.x < 512) { g_tiled_wg[blockIdx.x][0] = rt0; g_tiled_wg[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime(); }
  if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) g_tiled_prof[wave][k] = prof_acc[k];
    if (wave == 0) {
      PROF_T(pk3);
      g_tiled_prof[8][0] = pk1 - pk0; g_tiled_prof[8][1] = pk2 - pk1; g_tiled_prof[8][2] = pk3 - pk2;
      g_tiled_prof[8][4] = __builtin_amdgcn_s_memrealtime() - rt0;   // in-kernel time of workgroup 0, 10-ns ticks
      g_tiled_prof[8][3] = __builtin_amdgcn_s_memrealtime();   // 100 MHz: kernel-to-kernel period from successive launches
    }
  }
#endif
#undef PHASE_SYNC
  __syncthreads();   // both teams done before the workgroup-wide reductions below
  if constexpr (FUSE) {
    if (p.tot_out && tid < 32) {   // (the phase loop ends with a barrier)
      float a = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) a += bL[192 + w8 * 32 + tid];
      p.tot_out[(long)bid * 32 + tid] = a;
    }
  }
  if constexpr (STAT) {
    // lanes sharing q own the same channels: fold the 16 voxel lanes, then the 8 waves in fixed order through the (now
    // idle) halo region, one partial per workgroup
    constexpr int CHS = 4 * NT;
#pragma unroll
    for (int j = 0; j < CHS; ++j)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        st_p[j] += __shfl_xor(st_p[j], o, 64);
        st_q[j] += __shfl_xor(st_q[j], o, 64);
      }
    float* red = reinterpret_cast<float*>(lds + 27 * NT * 16 * 32);  // [8 waves][Co][2]
    if (v == 0) {
#pragma unroll
      for (int j = 0; j < CHS; ++j) {
        red[(wave * p.Co + q * CHS + j) * 2] = st_p[j];
        red[(wave * p.Co + q * CHS + j) * 2 + 1] = st_q[j];
      }
    }
    __syncthreads();
    for (int i = tid; i < p.Co * 2; i += 512) {
      float a = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) a += red[w8 * p.Co * 2 + i];
      p.stat_out[((long)bid * p.st_cs) * 2 + i] = a;  // bid = n * workgroups_per_sample + workgroup
    }
  }
}

// One problem per launch.  XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share an L2), so the
// logical workgroup index is permuted to give every XCD one contiguous run of bricks -- haloes shared with the y/x neighbour columns
// then hit that XCD's L2 instead of being re-read from HBM (PMC: x fetched 2.4x with the identity map).
template <int NT, bool HAS_BTAB, int AUX, bool STAT, int FUSEX = 0, bool GEN = false>
__global__ __launch_bounds__(512, 2) void conv_tiled_kernel(TiledParams p) {
  const int wgs_per_sample = p.teams_per_sample >> 1;
  const int bid = (gridDim.x % 8 == 0) ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int n = bid / wgs_per_sample;
  conv_tiled_body<NT, HAS_BTAB, AUX, STAT, FUSEX, GEN>(p, n, bid - n * wgs_per_sample);
}

// SEVERAL problems of one kernel variant share a launch (horizontal fusion, round 4): HRNet's branches run the same block structure
// side by side, so the full-resolution conv of a stage and the level-1 conv of the same position are independent launches of the
// same variant -- alone, the level-1 one occupied all 256 CUs for 1.5 bricks per team (26 us for a ninth of the work), and the two
// serialised.  Here the eight samples keep their XCDs and each XCD's 32 workgroups are split between the problems in proportion to
// their bricks (28 + 4 for full resolution + level 1: 11.4 and 12 bricks per team).  jobs[j] = the problem's parameters (built by
// the host exactly as for a plain launch, teams_per_sample = 2 x its share), split[j] = first workgroup of problem j within an XCD.
struct TiledMulti { const TiledParams* jobs; int njobs; int split[RTP_MULTI_MAX + 1]; };
template <int NT, bool HAS_BTAB, int AUX, bool STAT, int FUSEX = 0, bool GEN = false>
__global__ __launch_bounds__(512, 2) void conv_tiled_multi_kernel(TiledMulti m) {
  // grid = samples x workgroups per sample; the XCD-aware order of the plain kernel (every XCD one contiguous run of logical
  // workgroups), cut into samples: n = 8 -> sample = XCD, n = 16 -> two samples per XCD, n = 4 -> a sample on two XCDs
  const int per = m.split[m.njobs];
  const int bid = (gridDim.x % 8 == 0) ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int n = bid / per, l = bid - n * per;
  int j = 0;
#pragma unroll
  for (int k = 1; k < RTP_MULTI_MAX; ++k) j += (k < m.njobs && l >= m.split[k]) ? 1 : 0;
  j = __builtin_amdgcn_readfirstlane(j);
  const TiledParams p = m.jobs[j];   // (uniform address: scalar loads)
  conv_tiled_body<NT, HAS_BTAB, AUX, STAT, FUSEX, GEN>(p, n, l - m.split[j]);
}

static bool tiled_geometry_ok(const RtpAct* x, const RtpConvGeom* g, int transposed, int* Co_out) {
  static const bool disabled = getenv("RTP_DISABLE_TILED") != nullptr;
  if (disabled) return false;
  if (g->ks != 3 || g->stride != 1 || g->pad != 1) return false;
  const int Ci = transposed ? (g->co + 31) / 32 * 32 : g->ci;
  const int Co = transposed ? g->ci : g->co;
  if (Ci != 32 || (Co != 16 && Co != 32)) return false;
  if (g->di % TZ || g->hi % TY || g->wi % 16 || g->di < 2 || g->hi < 2) return false;
  if (x->cs % 32 || x->co % 8) return false;   // a 32-channel slice of a wider channels-last tensor is fine
  *Co_out = Co;
  return true;
}

static int tiled_wgs_per_sample(const RtpConvGeom* g) {
  const int tiles = (g->di / TZ) * (g->hi / TY) * ((g->wi + TX - 1) / TX);
  static const int total_wgs = 256;   // (a launch's own width: RtpConvGeom::wgs)
  // launches of the lower levels (fewer than 2048 bricks in all: the level-1 tensors at the native shape) are kept NARROW -- 64
  // workgroups with four times the bricks each: the 55 KB of weights (and the GroupNorm fold) a workgroup pays before its first brick
  // are amortised over 6 bricks per team instead of 1.5, and they fit on the CUs the hinted main-lane launches leave free (hr3d B = 8
  // step: -0.7 ... -2.3 % depending on the box for 128, another -0.5 % for 64 beside the width hints; RTP_TILED_WGS_SMALL, 0 = as the large ones)
  static const int small_wgs = 64;
  int wgs = ((small_wgs > 0 && (long)tiles * g->n < 2048) ? small_wgs : total_wgs) / g->n;  // workgroups per sample: one workgroup per CU when N divides 256
  // RtpConvGeom::wgs: the caller's launch width (narrower: CUs left to other streams; wider than the narrow default of a small
  // launch: a side chain the main stream is waiting for) -- also the number of per-workgroup partial slots, so queries and launches
  // of one geometry object agree by construction
  if (g->wgs > 0) wgs = (g->wgs > 256 ? 256 : g->wgs) / g->n;
  if (wgs < 1) wgs = 1;
  if (wgs * 2 > tiles) wgs = (tiles + 1) / 2;
  // small volumes: a workgroup pays ~6 us of fixed cost (55 KB of weights, pipeline ramp) whatever it computes; with fewer than
  // `minb` bricks per team that dominates (experiment knob; 0 = off)
  static const int minb = 0;
  if (minb > 0 && wgs * 2 * minb > tiles) { wgs = tiles / (2 * minb); if (wgs < 1) wgs = 1; }
  return wgs;
}

// Number of per-sample statistics partials the fused epilogue writes for this conv (0: not this kernel's geometry).
int rtp_conv_tiled_stat_slots(const RtpAct* x, const RtpConvGeom* g, int transposed) {
  int Co;
  return tiled_geometry_ok(x, g, transposed, &Co) ? tiled_wgs_per_sample(g) : 0;
}

// Returns RTP_OK if it handled the conv, +1 if the geometry is not this kernel's (caller falls through to the
// generic gather kernel), or a negative error.
struct S2Fuse { const float* coef[4]; const RtpAct* ex[3]; int nextra, mask; float* tot_out; const RtpGnBwd* gn; };   // dgrad_s2_tiled.hip
int rtp_dgrad_s2_try(const RtpAct* gy, const void* wd, const RtpAct* dx, const RtpConvGeom* g, const RtpAct* stat_x, float* stat_out,
                     const S2Fuse* fuse, hipStream_t s);
struct TiledFuse { const float* coef[4]; const RtpAct* ex[3]; int nextra, mask; float* tot_out; const RtpGnBwd* gn; };
struct TiledSlice { long w_sample_stride; int w_tap_stride, w_row_stride, bt_cs, st_cs; };

int rtp_conv_tiled_try(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                       const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32,
                       const RtpAct* stat_x, float* stat_out, const float* acc32, int acc_cs, hipStream_t s,
                       const TiledFuse* fuse, const RtpGnFold* fold, const TiledSlice* slice) {
  int Co;
  if (!tiled_geometry_ok(x, g, transposed, &Co)) return 1;
  if (slice && (fuse || fold)) return RTP_ERR_UNSUPPORTED;
  if (fold && (fuse || transposed || acc32 || y_fp32 || wf || btab || !fold->w || !fold->gamma || !fold->beta || !fold->stats ||
               fold->nsplit < 1 || fold->groups < 1 || fold->groups > 32 || 32 % fold->groups || fold->co_real < 1 ||
               fold->co_real > Co || g->w_ci_total > 32 || g->w_ci_off))
    return RTP_ERR_SHAPE;
  if (fuse && (Co != 32 || !transposed || btab || res || !stat_x || stat_out || acc32 || y_fp32 || relu)) return RTP_ERR_UNSUPPORTED;
  if (acc32 && (acc_cs % 4 || acc_cs < Co)) return RTP_ERR_ALIGN;
  if (stat_out && (y_fp32 || (stat_x && res))) return RTP_ERR_UNSUPPORTED;
  if (!fuse && stat_x && !stat_out) return RTP_ERR_SHAPE;
  if (stat_x) res = stat_x;  // rides in the residual's prefetch slot
  TiledParams p;
  p.stat_out = stat_out;
  p.x = (const bf16_t*)x->ptr; p.w = (const bf16_t*)wf; p.btab = btab;
  p.x_cs = x->cs; p.x_co = x->co; p.acc32 = acc32; p.a_cs = acc_cs;
  p.res = res ? (const bf16_t*)res->ptr : nullptr; p.y = y->ptr;
  p.N = g->n; p.D = g->di; p.H = g->hi; p.W = g->wi; p.Co = Co;
  p.y_cs = y->cs; p.y_co = y->co; p.r_cs = res ? res->cs : 0; p.r_co = res ? res->co : 0;
  p.relu = relu; p.y_fp32 = y_fp32; p.flip = transposed; p.w_per_sample = w_per_sample;
  p.tiles_y = p.H / TY; p.tiles_x = (p.W + TX - 1) / TX; p.tiles_z = p.D / TZ;
  p.tiles_per_sample = (p.D / TZ) * p.tiles_y * p.tiles_x;
  int wgs = tiled_wgs_per_sample(g);
  p.part_stride = wgs;   // (one partial slot per workgroup: rtp_conv_stats_nsplit reports the same count for this geometry)
  p.teams_per_sample = wgs * 2;
  if ((long)p.tiles_per_sample * (p.teams_per_sample + 1) >= (1L << 31)) return RTP_ERR_SHAPE;
  #ifdef RTP_TILED_DBGFLAGS
  static const int dbg = getenv("RTP_TILED_DBG") ? atoi(getenv("RTP_TILED_DBG")) : 0;   // (phase-skipping experiment builds only)
#else
  static const int dbg = 0;
#endif
  p.dbg = dbg;
  static const int prio = 0;
  p.prio = prio;
  p.nextra = 0; p.mask = 0; p.tot_out = nullptr;
  p.qpart = nullptr; p.q_nsplit = 0; p.gn_p = p.gn_mr = p.gn_gamma = nullptr; p.gn_groups = 1; p.gn_m = 1.f; p.coef_out = nullptr;
  p.tg = nullptr; p.csum_out = nullptr;
  p.fw = p.fbias = p.fgamma = p.fbeta = p.fstats = nullptr; p.f_nsplit = p.f_groups = p.f_co_real = 0; p.f_eps = 0.f; p.f_mr = nullptr;
  p.w_sample_stride = 27L * Co * 32; p.w_tap_stride = Co * 32; p.w_row_stride = 32; p.bt_cs = Co; p.st_cs = Co;
  if (slice) {
    p.w_sample_stride = slice->w_sample_stride; p.w_tap_stride = slice->w_tap_stride; p.w_row_stride = slice->w_row_stride;
    p.bt_cs = slice->bt_cs; p.st_cs = slice->st_cs;
    if (p.w_row_stride % 8 || p.w_tap_stride % 8 || p.w_sample_stride % 8) return RTP_ERR_ALIGN;
  }
  if (fold) {
    p.fw = fold->w; p.fbias = fold->bias; p.fgamma = fold->gamma; p.fbeta = fold->beta; p.fstats = fold->stats;
    p.f_nsplit = fold->nsplit; p.f_groups = fold->groups; p.f_co_real = fold->co_real; p.f_eps = fold->eps; p.f_mr = fold->mr;
    p.w_per_sample = 1;
  }
  for (int e = 0; e < 4; ++e) p.coef[e] = nullptr;
  for (int e = 0; e < 3; ++e) { p.ex[e] = nullptr; p.ex_cs[e] = p.ex_co[e] = 0; }
  if (fuse) {
    if (fuse->nextra < 0 || fuse->nextra > 3) return RTP_ERR_SHAPE;
    p.nextra = fuse->nextra; p.mask = fuse->mask; p.tot_out = fuse->tot_out;
    p.coef[0] = fuse->coef[0];
    if (fuse->gn) {
      const RtpGnBwd* q = fuse->gn;
      if (!q->qpart || (!q->p && !q->tg) || !q->mr || !q->gamma || q->q_nsplit < 1 || q->groups < 1 || 32 % q->groups) return RTP_ERR_SHAPE;
      p.tg = q->p ? nullptr : q->tg; p.csum_out = q->p ? nullptr : q->csum_out;
      p.qpart = q->qpart; p.q_nsplit = q->q_nsplit; p.gn_p = q->p; p.gn_mr = q->mr; p.gn_gamma = q->gamma;
      p.gn_groups = q->groups; p.gn_m = (float)(32 / q->groups) * (float)((long)g->di * g->hi * g->wi);
      p.coef_out = q->coeff_out;
    }
    for (int e = 0; e < fuse->nextra; ++e) {
      if (!fuse->ex[e] || fuse->ex[e]->c < 32 || (fuse->ex[e]->cs % 8) || (fuse->ex[e]->co % 8)) return RTP_ERR_ALIGN;
      p.ex[e] = (const bf16_t*)fuse->ex[e]->ptr; p.ex_cs[e] = fuse->ex[e]->cs; p.ex_co[e] = fuse->ex[e]->co;
      p.coef[1 + e] = fuse->coef[1 + e];
    }
  }
  const int nt = Co / 16;
  const size_t shm = sizeof(bf16_t) * (27 * (size_t)Co * 32 + 2 * (size_t)HALO_VOX * 32) + 27 * (size_t)Co * sizeof(float) + 64;
  if (std::vector<RtpMultiJob>* cap = rtp_multi_capture()) {   // recorded for a shared launch (rtp_multi.h), not issued
    const int aux_c = stat_x ? 2 : (res ? 1 : 0);
    // (besides the 32-channel variants: the head towers' last convs -- <= 16 output channels, class-bias table, fp32 output)
    const bool head_last = nt == 1 && !acc32 && y_fp32 && !slice && !fuse && (btab || fold) && aux_c == 0 && !stat_out;
    if (!head_last && (nt != 2 || acc32 || y_fp32 || slice)) return RTP_ERR_UNSUPPORTED;
    RtpMultiJob job;
    job.kind = RTP_MULTI_CONV_TILED;
    job.variant = head_last ? 200 : fuse ? (100 + p.nextra) : (((btab || fold) ? 1 : 0) * 8 + aux_c * 2 + (stat_out ? 1 : 0));
    job.tiles_per_sample = p.tiles_per_sample; job.n = p.N; job.slots_per_sample = p.part_stride; job.shm = shm;
    // (the family the problem is timed under when launched alone, below)
    job.fam = (Co == 32 && (long)p.N * p.D * p.H * p.W >= (1L << 20)) ? (transposed ? RTP_FAM_CONV_TILED_FULL_BWD : RTP_FAM_CONV_TILED_FULL)
                                                                        : RTP_FAM_CONV_TILED;
    job.params.assign((const char*)&p, (const char*)&p + sizeof(p));
    cap->push_back(job);
    return RTP_OK;
  }
  RtpProfScope prof((Co == 32 && (long)p.N * p.D * p.H * p.W >= (1L << 20)) ? (transposed ? RTP_FAM_CONV_TILED_FULL_BWD : RTP_FAM_CONV_TILED_FULL)
                                                                                : RTP_FAM_CONV_TILED, s);
  using Kern = void (*)(TiledParams);
#define RTP_TILED_CELL(NT, BT, AUX, ST) {conv_tiled_kernel<NT, BT, AUX, ST, 0, false>, conv_tiled_kernel<NT, BT, AUX, ST, 0, true>}
#define RTP_TILED_ROW(NT, BT) \
  {{RTP_TILED_CELL(NT, BT, 0, false), RTP_TILED_CELL(NT, BT, 0, true)}, \
   {RTP_TILED_CELL(NT, BT, 1, false), RTP_TILED_CELL(NT, BT, 1, true)}, \
   {{nullptr, nullptr}, RTP_TILED_CELL(NT, BT, 2, true)}}
  static const Kern table[2][2][3][2][2] = {{RTP_TILED_ROW(1, false), RTP_TILED_ROW(1, true)},
                                            {RTP_TILED_ROW(2, false), RTP_TILED_ROW(2, true)}};   // [nt][class bias][aux][statistics][fp32 in/out]
#undef RTP_TILED_ROW
#undef RTP_TILED_CELL
  static bool attr_done[RTP_MAX_DEVICES] = {};
  if (rtp_once_per_device(attr_done)) {
    const int big = (int)(sizeof(bf16_t) * (27 * (size_t)32 * 32 + 2 * (size_t)HALO_VOX * 32) + 27 * 32 * sizeof(float) + 64);
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        for (int c = 0; c < 3; ++c)
          for (int d = 0; d < 2; ++d)
            for (int e = 0; e < 2; ++e)
              if (table[a][b][c][d][e])
                (void)hipFuncSetAttribute((const void*)table[a][b][c][d][e], hipFuncAttributeMaxDynamicSharedMemorySize, big);
  }
  const int aux = stat_x ? 2 : (res ? 1 : 0);
  if (fuse) {
    static const Kern ftab[4] = {conv_tiled_kernel<2, false, 2, false, 1>, conv_tiled_kernel<2, false, 2, false, 2>,
                                 conv_tiled_kernel<2, false, 2, false, 3>, conv_tiled_kernel<2, false, 2, false, 4>};
    static bool fattr[RTP_MAX_DEVICES] = {};
    if (rtp_once_per_device(fattr)) {
      for (int e = 0; e < 4; ++e)
        (void)hipFuncSetAttribute((const void*)ftab[e], hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(bf16_t) * (27 * (size_t)32 * 32 + 2 * (size_t)HALO_VOX * 32) + 27 * 32 * sizeof(float) + 64));
    }
    hipLaunchKernelGGL(ftab[p.nextra], dim3(p.N * wgs), dim3(512), shm, s, p);
    RTP_CHECK_LAUNCH();
    return RTP_OK;
  }
  hipLaunchKernelGGL(table[nt - 1][(btab || fold) ? 1 : 0][aux][stat_out ? 1 : 0][(acc32 || y_fp32) ? 1 : 0], dim3(p.N * wgs), dim3(512), shm, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// GroupNorm -> conv -> (+ residual) -> (ReLU) with the fold in the kernel's prologue (include/rtp.h).
extern "C" int rtp_conv_gn_fused(const RtpAct* x, const RtpGnFold* f, const RtpAct* res, const RtpAct* y, const RtpConvGeom* g,
                                 int relu, float* stat_out, void* stream) {
  if (!x || !f || !y || !g) return RTP_ERR_SHAPE;
  if ((x->co % 8) || (x->cs % 8) || (y->co % 8) || (y->cs % 8) || (res && ((res->co % 8) || (res->cs % 8)))) return RTP_ERR_ALIGN;
  const int rc = rtp_conv_tiled_try(x, nullptr, 1, nullptr, res, y, g, relu, 0, 0, nullptr, stat_out, nullptr, 0, (hipStream_t)stream,
                                    nullptr, f, nullptr);
  return rc > 0 ? RTP_ERR_UNSUPPORTED : rc;
}

// Data gradient of a 3x3x3 stride-1 32 -> <=32-channel conv that writes the FINISHED gradient of the conv's input x:
//   dx = [x > 0 if mask] * (A*convT(gy; wd) + B*x + C + sum_k term_k)
// coeff = this conv's GroupNorm-backward coefficients [n][32][3] (rtp_gn_bwd_coeffs / _cls) or NULL (no GroupNorm);
// terms = the contributions of x's other consumers (RtpTerm: plain addend, or dxhat + coefficients of another GroupNorm
// consumer), at most 3.  tot_out (optional): per-channel sums of the stored dx, one partial per workgroup
// [n][rtp_conv_stats_nsplit(gy, g, 1)][32] (rtp_class_sums_boundary completes them to per-class sums).
// Only the LDS-tiled kernel's geometries (rtp_conv_tiled_ok(gy, g, 1)).
extern "C" int rtp_conv_dgrad_fused(const RtpAct* gy, const void* wd, const RtpAct* x, const float* coeff, const RtpGnBwd* gn,
                                    const RtpTerm* terms /*host*/, int nterms, int mask, const RtpAct* dx,
                                    const RtpConvGeom* g, float* tot_out, void* stream) {
  if (!gy || !wd || !x || !dx || !g || nterms < 0 || nterms > 3 || (nterms && !terms)) return RTP_ERR_SHAPE;
  if (coeff && gn) return RTP_ERR_SHAPE;
  if ((gy->co % 8) || (gy->cs % 8) || (dx->co % 8) || (dx->cs % 8) || (x->co % 8) || (x->cs % 8)) return RTP_ERR_ALIGN;
  if (x->c < 32 || dx->c < 32) return RTP_ERR_SHAPE;
  if (gn && gn->groups != 8 && gn->groups != 1) return RTP_ERR_UNSUPPORTED;
  if (g->stride == 2) {   // the parity-class kernel (dgrad_s2_tiled.hip)
    if (rtp_multi_capture()) return RTP_ERR_UNSUPPORTED;
    S2Fuse f2;
    f2.nextra = nterms; f2.mask = mask; f2.tot_out = tot_out; f2.gn = gn;
    f2.coef[0] = coeff;
    for (int e = 0; e < 3; ++e) { f2.ex[e] = e < nterms ? &terms[e].t : nullptr; f2.coef[1 + e] = e < nterms ? terms[e].coeff : nullptr; }
    const int rc2 = rtp_dgrad_s2_try(gy, wd, dx, g, x, nullptr, &f2, (hipStream_t)stream);
    return rc2 > 0 ? RTP_ERR_UNSUPPORTED : rc2;
  }
  TiledFuse f;
  f.nextra = nterms; f.mask = mask; f.tot_out = tot_out; f.gn = gn;
  f.coef[0] = coeff;
  for (int e = 0; e < 3; ++e) { f.ex[e] = e < nterms ? &terms[e].t : nullptr; f.coef[1 + e] = e < nterms ? terms[e].coeff : nullptr; }
  const int rc = rtp_conv_tiled_try(gy, wd, 0, nullptr, nullptr, dx, g, 0, 1, 0, x, nullptr, nullptr, 0, (hipStream_t)stream, &f, nullptr, nullptr);
  return rc > 0 ? RTP_ERR_UNSUPPORTED : rc;
}

int rtp_dgrad_s2_stat_slots(const RtpAct* gy, const RtpConvGeom* g);
// 1 if rtp_conv_dgrad_fused has a kernel for this data gradient (gy = output-side gradient view, g = forward geometry)
extern "C" int rtp_conv_dgrad_fused_ok(const RtpAct* gy, const RtpConvGeom* g) {
  if (!gy || !g) return 0;
  int Co;
  if (g->stride == 2) return rtp_dgrad_s2_stat_slots(gy, g) > 0 ? 1 : 0;
  return (tiled_geometry_ok(gy, g, 1, &Co) && Co == 32) ? 1 : 0;
}

// ---- shared launches (rtp_multi.h)
namespace {
struct ConvMultiLauncher { void (*kern)(TiledMulti); TiledMulti m; size_t shm; int n; };
using MKern = void (*)(TiledMulti);
MKern conv_multi_kernel_for(int variant) {
  switch (variant) {
    case 0 * 8 + 0 * 2 + 0: return conv_tiled_multi_kernel<2, false, 0, false>;
    case 0 * 8 + 0 * 2 + 1: return conv_tiled_multi_kernel<2, false, 0, true>;
    case 0 * 8 + 1 * 2 + 0: return conv_tiled_multi_kernel<2, false, 1, false>;
    case 0 * 8 + 1 * 2 + 1: return conv_tiled_multi_kernel<2, false, 1, true>;
    case 0 * 8 + 2 * 2 + 1: return conv_tiled_multi_kernel<2, false, 2, true>;
    case 1 * 8 + 0 * 2 + 0: return conv_tiled_multi_kernel<2, true, 0, false>;
    case 1 * 8 + 0 * 2 + 1: return conv_tiled_multi_kernel<2, true, 0, true>;
    case 1 * 8 + 1 * 2 + 0: return conv_tiled_multi_kernel<2, true, 1, false>;
    case 1 * 8 + 1 * 2 + 1: return conv_tiled_multi_kernel<2, true, 1, true>;
    case 1 * 8 + 2 * 2 + 1: return conv_tiled_multi_kernel<2, true, 2, true>;
    case 100: return conv_tiled_multi_kernel<2, false, 2, false, 1>;
    case 101: return conv_tiled_multi_kernel<2, false, 2, false, 2>;
    case 102: return conv_tiled_multi_kernel<2, false, 2, false, 3>;
    case 103: return conv_tiled_multi_kernel<2, false, 2, false, 4>;
    case 200: return conv_tiled_multi_kernel<1, true, 0, false, 0, true>;
    default: return nullptr;
  }
}
}  // namespace

int rtp_conv_tiled_multi_finish(std::vector<RtpMultiJob>& jobs, const int* share, void* dev_params, void** launcher) {
  MKern k = conv_multi_kernel_for(jobs[0].variant);
  if (!k) return RTP_ERR_UNSUPPORTED;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)jobs[0].shm) != hipSuccess) return RTP_ERR_LAUNCH;
  std::vector<TiledParams> host(jobs.size());
  ConvMultiLauncher* L = new ConvMultiLauncher();
  L->kern = k; L->shm = jobs[0].shm;
  L->m.jobs = (const TiledParams*)dev_params; L->m.njobs = (int)jobs.size(); L->n = jobs[0].n;
  int at = 0;
  for (size_t j = 0; j < jobs.size(); ++j) {
    if (jobs[j].params.size() != sizeof(TiledParams)) { delete L; return RTP_ERR_SHAPE; }
    memcpy(&host[j], jobs[j].params.data(), sizeof(TiledParams));
    host[j].teams_per_sample = 2 * share[j];   // (part_stride keeps the slots the problem's partial buffers were sized for)
    L->m.split[j] = at;
    at += share[j];
  }
  for (size_t j = jobs.size(); j <= RTP_MULTI_MAX; ++j) L->m.split[j] = at;
  if (hipMemcpy(dev_params, host.data(), host.size() * sizeof(TiledParams), hipMemcpyHostToDevice) != hipSuccess) { delete L; return RTP_ERR_LAUNCH; }
  *launcher = L;
  return RTP_OK;
}

int rtp_conv_tiled_multi_launch(void* launcher, hipStream_t s) {
  ConvMultiLauncher* L = (ConvMultiLauncher*)launcher;
  hipLaunchKernelGGL(L->kern, dim3(L->n * L->m.split[L->m.njobs]), dim3(512), L->shm, s, L->m);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

void rtp_conv_tiled_multi_drop(void* launcher) { delete (ConvMultiLauncher*)launcher; }
