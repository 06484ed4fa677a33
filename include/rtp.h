/* rtp.h -- C ABI of librtp_hip.so, the MI355X (gfx950) kernel library behind rt_pose_amd.
 *
 * Drop-in boundary.  The reference reaches its arithmetic through two doors:
 *   (1) torch.nn modules (Conv3d / GroupNorm / interpolate / losses) built by the det3d registry
 *       -- det3d/models/backbones/hr_util/common.py:40,57, hr_util/hr3d.py:83-90,147-224,
 *          det3d/models/pose_heads/center_head.py:86-93,240-360, det3d/models/losses/centernet_loss.py:17-54.
 *       Those have no native ABI in the reference; the entry points in section A/B/C below are what the
 *       Python host mirror (rt_pose_amd/modules.py, registered under the same registry names) binds.
 *   (2) the pybind module `deform_conv_cuda` (det3d/ops/dcn/src/deform_conv_cuda.cpp:687-701), five
 *       functions taking at::Tensor.  Section D declares their C-ABI replacements one for one.
 *
 * Conventions: plain pointers + sizes, no torch types; every pointer is DEVICE memory unless
 * marked host; the callee never allocates, never synchronises, launches on `stream` (hipStream_t
 * passed as void*), is re-entrant, and returns 0 or a negative RTP_ERR_* code (no exceptions
 * cross the boundary).  Activations are bf16, channels-last: [N][D][H][W][C] with an optional
 * channel stride/offset so a kernel can read or write a channel slice of a wider buffer.
 */
#ifndef RTP_H_
#define RTP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define RTP_OK 0
#define RTP_ERR_SHAPE (-1)       /* inconsistent or unsupported geometry */
#define RTP_ERR_UNSUPPORTED (-2) /* channel count / kernel size this build has no kernel for */
#define RTP_ERR_LAUNCH (-3)      /* hipGetLastError() after the launch */
#define RTP_ERR_ALIGN (-4)       /* pointer or channel offset not 16-byte aligned */

#define RTP_MAX_TERMS 6

/* A channels-last activation (or gradient) tensor view. */
typedef struct RtpAct {
  void* ptr;   /* bf16 unless a kernel says fp32 */
  int cs;      /* channel stride: elements per voxel in the underlying buffer */
  int co;      /* first channel of the view */
  int c;       /* channels in the view */
} RtpAct;

/* Geometry of one convolution (forward orientation). */
typedef struct RtpConvGeom {
  int n;
  int di, hi, wi; /* input  spatial dims */
  int dov, ho, wo; /* output spatial dims */
  int ci, co;     /* logical channels (ci multiple of 32 after padding; co padded to 16) */
  int ks;         /* 1 or 3 (cubic) */
  int stride;     /* 1 or 2 */
  int pad;        /* 0 (ks=1) or 1 (ks=3) */
  int w_ci_total; /* the fp32 weight tensor is [co][w_ci_total][ks^3]; this conv uses input channels   */
  int w_ci_off;   /* [w_ci_off, w_ci_off+ci_real) of it (0,0 = the whole tensor; final_conv chunks)   */
  int wgs;        /* launch width of the persistent LDS-tiled kernels: 0 = their own choice (one workgroup per CU; 64 / 128 for the
                   * small launches of the lower levels); > 0 = this many workgroups in all (wgs / n per sample, at least one, at most
                   * 256 in all and one per brick).  The launch's tensor output is bit-identical.  The width is also the number of
                   * per-workgroup partial slots (statistics, slabs, Q / subset-sum partials): the query functions
                   * (rtp_conv_stats_nsplit, rtp_wgrad_nsplit) report the count FOR THE GEOMETRY THEY ARE GIVEN, wgs included, and a
                   * launch writes every slot -- size the buffers from a query with the same geometry object the launch gets.  The
                   * partials sum to the same totals up to summation order.  Why narrower: a main-stream launch that leaves a quarter
                   * of the CUs alone costs itself 5 % and lets other streams' dependent chains run beside it (DESIGN.md 8, round 4:
                   * -3.8 % on the hr3d step); why wider: a side chain's launches while the main stream waits for them (round 5).
                   * Honoured by conv_tiled / wgrad_tiled / wgrad_s2_tiled and by the 64-wide kernel (conv64_tiled.hip: rtp_conv_igemm_ws on
                   * 64 -> 64 layers -- its statistics slot count follows it -- and rtp_conv64_blocks); other kernels ignore the field. */
} RtpConvGeom;

/* ---------------------------------------------------------------- A. convolution family --- */

/* Per-channel sums for GroupNorm statistics and for the GN backward reductions.
 *   b == NULL : out[n][s][c][0] = sum_v a, [1] = sum_v a*a
 *   b != NULL : out[n][s][c][0] = sum_v a, [1] = sum_v a*b
 * out: fp32 [n][nsplit][c][2]; deterministic (fixed voxel->split assignment, no atomics).
 * Replaces the statistics half of torch.nn.GroupNorm (hr_util/common.py:57). */
int rtp_chan_stats(const RtpAct* a, const RtpAct* b, int n, long vox, int nsplit, float* out, void* stream);

/* Fold GroupNorm into per-sample conv weights (+ boundary-class bias table) and emit mean/rstd.
 *   w      fp32 [co][ci][ks^3]  (reference nn.Conv3d layout), bias fp32 [co] or NULL
 *   gamma/beta fp32 [ci] or NULL (no norm); stats = rtp_chan_stats output of the conv INPUT or NULL
 *   wf     bf16 [nw][ks^3][co_pad][ci_pad]   nw = n with norm, 1 without
 *   btab   fp32 [nw][64][co_pad]  bias for each boundary class of an output voxel
 *   mr     fp32 [n][groups][2]  (mean, rstd) saved for backward, or NULL
 *   wd     bf16 [ks^3][ci_pad][cok] or NULL: the same (un-folded) weights packed for the data-gradient conv
 *          (what rtp_pack_dgrad_w produces), emitted here so training needs no separate packing launch;
 *          wf == NULL with wd != NULL packs only wd (no statistics needed)
 * Replaces GroupNorm-apply + weight layout of common.py:25-71 / hr3d.py:147-155. */
int rtp_fold_fwd(const float* w, const float* bias, const float* gamma, const float* beta, const float* stats,
                 int nsplit, int groups, float eps, const RtpConvGeom* g, int ci_real, int co_real, void* wf,
                 float* btab, float* mr, void* wd, void* stream);

/* Pack weights for the data-gradient conv: wd bf16 [ks^3][ci_pad][cok] with cok = co rounded up to 32. */
int rtp_pack_dgrad_w(const float* w, const RtpConvGeom* g, int ci_real, int co_real, void* wd, void* stream);

/* Implicit-GEMM convolution on MFMA (v_mfma_f32_16x16x32_bf16).
 *   transposed == 0 : y = conv(x; wf) + btab[class] (+ res) (ReLU)
 *   transposed == 1 : y = conv_transpose(x; wf) i.e. the data gradient of a forward conv with geometry g
 *                     (x is then the output-side gradient, y the input-side gradient; no bias)
 * y_fp32: store fp32 instead of bf16 (head logits).  Replaces nn.Conv3d forward/backward-data
 * (common.py:40, hr3d.py:149-197, center_head.py:86-93). */
int rtp_conv_igemm(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                   const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32, void* stream);

/* rtp_conv_igemm that also emits per-channel statistics of the tensor it writes, saving the rtp_chan_stats read pass:
 * stat_out fp32 [n][S][cout][2], S = rtp_conv_stats_nsplit(...) partials per sample (reduced by the consumers exactly
 * like rtp_chan_stats partials).  stat_x == NULL: (sum y, sum y*y) of the stored bf16 output -- what the next
 * GroupNorm's forward needs.  stat_x != NULL (same shape as y; not together with res): (sum y, sum y*stat_x) -- P and
 * Q of GroupNorm backward when y is the data gradient dxhat and stat_x the normalised tensor.  Only geometries for
 * which rtp_conv_stats_nsplit() > 0 (the LDS-tiled kernel's) are accepted; others return RTP_ERR_UNSUPPORTED. */
int rtp_conv_igemm_stats(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                         const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32,
                         const RtpAct* stat_x, float* stat_out, void* stream);

/* Wide 3x3x3 stride-1 convs (Cin = 32 K, Cout = 32 J, K * J > 1: the 64- and 128-channel layers of the feat64 backbone,
 * det3d/models/backbones/hrnet3D_config.py:149-177, hr_util/common.py:98-148) as K x J launches of the LDS-tiled 32 -> 32
 * kernel over channel slices, the partial sums carried in the caller's fp32 workspace; stride-2 forward convs of those widths
 * (hr_util/hr3d.py:162-197, 297-305) likewise on the LDS-tiled stride-2 kernel (csrc/conv_s2_tiled.hip).
 * rtp_conv_sliced_ok: 1 if rtp_conv_igemm_ws runs (x, g, transposed) that way.
 * rtp_conv_igemm_ws: rtp_conv_igemm / rtp_conv_igemm_stats (stat_x / stat_out may be NULL) with ws = n * (output voxels) * 32
 * floats of scratch; geometries that are not sliced ignore ws.  rtp_conv_stats_nsplit gives the partial count either way. */
int rtp_conv_sliced_ok(const RtpAct* x, const RtpConvGeom* g, int transposed);

/* --- 64-channel-wide stride-1 3x3x3 convolution in 32-channel BLOCKS (csrc/conv64_tiled.hip; round 5).
 * The kernel behind the 64 -> 64 layers of rtp_conv_igemm_ws, with every operand addressed per 32-channel half, so that two
 * 32-channel layers can share one launch -- SepHead's two towers over a 128- / 256-channel feature (center_head.py:86-93):
 *   forward          x[k] = input-channel halves of ONE 64-channel slice of the feature, w[h][k] = tower h's weights for that slice,
 *                    y[h] = tower h's output, btab[h] = its class-bias table; a feature wider than 64 channels is a CHAIN of calls
 *                    over its slices through `acc` (n * voxels * 64 floats, in a layout private to the chain: the links must share geometry and
 *                    g->wgs): acc_out on all but the last (raw sums, nothing else is written), acc_in on all but the first;
 *   data gradient    transposed = 1 (taps flipped): x[k] = the gradient of tower k's output, w[h][k] = tower k's data-gradient
 *                    weights for input channels 32 h .. 32 h + 31 of the slice, y[h] = those 32 channels of the feature's gradient --
 *                    the SUM over both towers, one tensor instead of two for the fan-in pass.
 * Weight block w[h][k]: [sample | 1][27 taps][32 rows = output channels of half h][32 columns = input channels of half k] with the
 * given element strides (w_tap_stride between taps, w_row_stride between rows, w_sample_stride between samples when w_per_sample).
 * btab[h]: [sample | 1][64 classes][bt_cs] or NULL; res[h]: residual added before the ReLU (both or neither).  Pointers carry the
 * channel offsets (x, w: 16-byte aligned; y, res: 8-byte).  Geometry: ks 3, stride 1, pad 1, D % 2 == 0, H % 4 == 0, W % 16 == 0;
 * g->ci / g->co are not read; g->wgs as everywhere (0: one workgroup per CU).  No allocation, no statistics. */
typedef struct RtpConv64 {
  const void* x[2]; int x_cs[2];
  const void* w[2][2]; int w_row_stride, w_tap_stride; long w_sample_stride; int w_per_sample;
  const float* btab[2]; int bt_cs;
  const void* res[2]; int r_cs[2];
  void* y[2]; int y_cs[2];
  float* acc; int acc_in, acc_out;
  int relu, transposed;
} RtpConv64;
int rtp_conv64_blocks(const RtpConv64* c, const RtpConvGeom* g, void* stream);

int rtp_conv_stats_nsplit_ws(const RtpAct* x, const RtpConvGeom* g, int transposed);   /* partial count of rtp_conv_igemm_ws */
int rtp_conv_igemm_ws(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                      const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32,
                      const RtpAct* stat_x, float* stat_out, float* ws, void* stream);
int rtp_conv_stats_nsplit(const RtpAct* x, const RtpConvGeom* g, int transposed);

/* Contraction split of a wide conv (head towers, center_head.py:86-93): each 32-channel slice of the contracted tensor
 * (an RtpAct with cs = total channels, co = 32*k) is one conv on the LDS-tiled kernel; slice k > 0 adds the fp32 partial
 * result of the slices before it (acc32: dense fp32 [n][vox][acc_cs], written by the previous slice with y_fp32 = 1)
 * before bias / ReLU.  Forward (Cin = 64/128/256 split) and transposed (data gradient of a Cout = 48 conv: the contraction
 * over the output-side gradient's channels is split); geometries for which rtp_conv_tiled_ok() returns 0 are
 * RTP_ERR_UNSUPPORTED. */
int rtp_conv_igemm_acc(const RtpAct* x, const void* wf, int w_per_sample, const float* btab, const RtpAct* res,
                       const RtpAct* y, const RtpConvGeom* g, int relu, int transposed, int y_fp32, const float* acc32,
                       int acc_cs, void* stream);
int rtp_conv_tiled_ok(const RtpAct* x, const RtpConvGeom* g, int transposed);

/* Weight-gradient correlation: gp[n][s][tap][co32][ci_pad] (fp32 partial slabs, s = voxel split) =
 * sum over the split's output voxels of gy[v][co] * x[v*stride + tap - pad][ci].  Replaces
 * nn.Conv3d backward-weight. */
int rtp_wgrad(const RtpAct* gy, const RtpAct* x, const RtpConvGeom* g, int nsplit, float* gp, void* stream);
/* Slab count the LDS-tiled weight-gradient kernel wants for this geometry (one slab per workgroup); 0 = the generic
 * kernel will run and any nsplit >= 1 is accepted. */
int rtp_wgrad_nsplit(const RtpConvGeom* g);

/* rtp_wgrad on the LDS-tiled kernel (rtp_wgrad_nsplit(g) > 0 and nsplit equal to it) that ALSO contracts every slab with
 * the data-gradient weights wd (bf16 [ks^3][ci_pad][cok], what rtp_fold_fwd / rtp_pack_dgrad_w write):
 *   qpart[n][s][ci] = sum_{tap,co} wd[tap][ci][co] * gp[n][s][tap][co][ci]
 * Summed over s this is Q = sum_v dxhat[v][ci] * x[v][ci] of GroupNorm backward (dxhat = the data gradient), obtained
 * without a pass over dxhat -- so the data gradient can run AFTER its GroupNorm coefficients are known and write the
 * finished gradient (rtp_conv_dgrad_fused).
 * tg (optional, fp32 [n][nsplit][27][32]): the kernel's loader waves also sum gy over the volume and its faces / edges /
 * corners (27 subsets: per axis all | first plane | last plane, slot (az*3+ay)*3+ax), one partial table per slab, summed in
 * a fixed order -- what P, the bias gradient and the GroupNorm un-fold need, without a pass over gy. */
int rtp_wgrad_q(const RtpAct* gy, const RtpAct* x, const RtpConvGeom* g, int nsplit, float* gp, const void* wd,
                float* qpart, float* tg, void* stream);
/* rtp_wgrad on the LDS-tiled kernel that also emits the subset sums of gy (tg [n][nsplit][27][32]) and nothing else. */
int rtp_wgrad_tg(const RtpAct* gy, const RtpAct* x, const RtpConvGeom* g, int nsplit, float* gp, float* tg, void* stream);
/* The slab contraction of rtp_wgrad_q for slabs ANY weight-gradient kernel wrote (the generic kernel's, e.g. the stride-2
 * layers): qpart[n][s][ci] = sum_{tap,co} wd[tap][ci][co] * gp[n][s][tap][co][ci];  gp fp32 [n][nsplit][ntap][co32][ci]. */
int rtp_qpart_from_slabs(const float* gp, int n, int nsplit, int ntap, int co32, int ci, const void* wd, float* qpart,
                         void* stream);
/* dst[0..n) = 0 (device fp32), on `stream`: one launch for every accumulate-into buffer of a step. */
int rtp_zero_f32(float* dst, long n, void* stream);

/* Per boundary-class channel sums of an output-side gradient: out fp32 [n][64][c];
 * scratch fp32 [n][nsplit][64][c] (row-split partials, reduced in fixed order).  out == NULL: partials only. */
int rtp_class_sums(const RtpAct* gy, int n, int d, int h, int w, int nsplit, float* scratch, float* out, void* stream);
/* rtp_class_sums of a tensor whose per-channel TOTALS are already known (tot_part fp32 [n][tot_nsplit][c], e.g. written by
 * rtp_conv_dgrad_fused): only the boundary voxels are read (1/6 of a 16x64x160 volume); the interior class is the total
 * minus the boundary classes.  out fp32 [n][64][c]; scratch fp32 [n][nsplit][64][c]. */
int rtp_class_sums_boundary(const RtpAct* gy, int n, int d, int h, int w, int nsplit, float* scratch, const float* tot_part,
                            int tot_nsplit, float* out, void* stream);
/* Class sums of gy AND (wd != NULL) P of the GroupNorm backward of the conv whose output gradient gy is (rtp_gn_bwd_p), in
 * ONE launch: the last scan block of each sample (agent-scope counter) reduces the partials and computes P.
 * tot_part != NULL: boundary-only scan as rtp_class_sums_boundary.  counters: device int [n], zeroed once by the caller
 * (each launch leaves them zero).  csum_out fp32 [n][64][c]; p_out fp32 [n][ci]. */
int rtp_class_sums_p(const RtpAct* gy, int n, int d, int h, int w, int nsplit, float* scratch, const float* tot_part,
                     int tot_nsplit, float* csum_out, const void* wd, const RtpConvGeom* g, int ci_real, int co_real,
                     float* p_out, int* counters, void* stream);
/* Fixed-order reduction of class-sum partials [n][nsplit][64][c] -> out [n][64][c]. */
int rtp_class_sums_reduce(const float* scratch, int nsplit, int n, int c, float* out, void* stream);

/* Finish the weight gradient: reduce slabs, undo the GroupNorm fold, write reference-layout fp32 grads.
 *   dw[co][ci][tap] (+)= sum_n scale[n][ci] * sum_s gp + shift[n][ci] * sum_{classes where tap in-bounds} csum[n][cls][co]
 *   dbias[co] (+)= sum of csum over all classes (if dbias != NULL) */
int rtp_wgrad_fold(const float* gp, int nsplit, const float* csum, const float* mr, const float* gamma,
                   const float* beta, int groups, const RtpConvGeom* g, int ci_real, int co_real, float* dw,
                   float* dbias, int accumulate, void* stream);

/* Deferred tail of the backward sweep.  Class-sum reductions, slab folds and GroupNorm parameter sums only feed the
 * optimiser and are microseconds of work each; instead of ~110 launches per step the host records one descriptor per
 * item and runs every item of a dependency stage as ONE grid.  Usage: fill host descriptors
 * (rtp_tail_desc_bytes() bytes each) with the rtp_tail_desc_* calls -- same arguments and semantics as the single-item
 * entry points, plus the item's block count and dynamic LDS bytes -- copy the array to the device together with
 * block_start[count+1] (exclusive prefix sums of the block counts) and launch with the MAX of the LDS sizes.
 * Items of one launch must be independent (class reductions + GroupNorm parameter sums first, the folds second). */
int rtp_tail_desc_bytes(void);
int rtp_tail_desc_class_reduce(const float* scratch, int nsplit, int n, int c, float* out, void* desc /*host*/,
                               int* blocks, int* shm_bytes);
int rtp_tail_desc_wgrad_fold(const float* gp, int nsplit, const float* csum, const float* mr, const float* gamma,
                             const float* beta, int groups, const RtpConvGeom* g, int ci_real, int co_real, float* dw,
                             float* dbias, int accumulate, void* desc /*host*/, int* blocks, int* shm_bytes);
/* ... for a conv with bias and without GroupNorm whose weight gradient came from rtp_wgrad_tg: dbias from the subset-sum
 * partials tg [n][nsplit][27][32] (slot 0 = whole-volume sums of gy) -- no class-sum pass over gy (center_head.py:86-93 towers) */
int rtp_tail_desc_wgrad_fold_tg(const float* gp, int nsplit, const float* tg, const RtpConvGeom* g, int ci_real, int co_real,
                                float* dw, float* dbias, int accumulate, void* desc, int* blocks, int* shm_bytes);
/* coeff: the buffer rtp_gn_bwd_coeffs(..., dgamma = NULL, dbeta = NULL, ...) wrote. */
int rtp_tail_desc_gn_param(const float* coeff, int n, int c, float* dgamma, float* dbeta, int accumulate,
                           void* desc /*host*/, int* blocks, int* shm_bytes);
/* Weight packing that needs no activations (a conv without GroupNorm; wf == NULL: only the data-gradient packing wd):
 * batched so one launch at the start of a step serves every layer. */
/* wt[tap][co_pad][ci] (fp32) = w[co][ci][tap], rows co >= co_real zero: the weight layout rtp_conv_gn_fused reads. */
int rtp_tail_desc_pack_wt(const float* w, int co_real, int co_pad, int ci, int ntap, float* wt, void* desc /*host*/, int* blocks,
                          int* shm_bytes);
int rtp_tail_desc_fold_fwd(const float* w, const float* bias, const float* gamma, const float* beta, const float* stats,
                           int nsplit, int groups, float eps, const RtpConvGeom* g, int ci_real, int co_real, void* wf,
                           float* btab, float* mr, void* wd, void* desc /*host*/, int* blocks, int* shm_bytes);
int rtp_tail_launch(const void* descs /*device*/, const int* block_start /*device*/, int count, int total_blocks,
                    int shm_bytes, void* stream);

/* GroupNorm backward coefficients from pq = rtp_chan_stats(dxhat, x):
 *   coeff[n][c] = (A, B, C) with  dx = A*dxhat + B*x + C ;  dgamma/dbeta (+)= per-channel param grads.
 *   The coeff buffer must hold n*c*5 floats (n*c*3 coefficients followed by n*c*2 of scratch).
 *   dgamma == dbeta == NULL: only the coefficients (and the scratch rtp_tail_desc_gn_param later sums). */
int rtp_gn_bwd_coeffs(const float* pq, int nsplit, const float* mr, const float* gamma, int n, int c, int groups,
                      long vox, float* coeff, float* dgamma, float* dbeta, int accumulate, void* stream);

/* The same coefficients without a pass over dxhat (one block per sample):
 *   Q[n][ci] = sum_s qpart[n][s][ci]                              (rtp_wgrad_q)
 *   P[n][ci] = sum_{tap,co} wd[tap][ci][co] * sum_{boundary classes in which `tap` stays in bounds} csum[n][cls][co]
 * cls_part: fp32 [n][cls_nsplit][64][co32] class-sum partials of the conv's OUTPUT gradient (rtp_class_sums with
 * out == NULL, rtp_grad_combine_cls); csum_out (optional) receives their fixed-order reduction [n][64][co32], i.e. what
 * rtp_class_sums_reduce would write.  coeff: as rtp_gn_bwd_coeffs (n*ci*5 floats).  ci_real == g->ci required. */
int rtp_gn_bwd_coeffs_cls(const float* qpart, int q_nsplit, const float* cls_part, int cls_nsplit, float* csum_out,
                          const void* wd, const float* mr, const float* gamma, const RtpConvGeom* g, int ci_real,
                          int co_real, int groups, float* coeff, void* stream);

/* P alone (the class-sum half of rtp_gn_bwd_coeffs_cls): p_out fp32 [n][ci]; runs beside the weight gradient. */
int rtp_gn_bwd_p(const float* cls_part, int cls_nsplit, float* csum_out, const void* wd, const RtpConvGeom* g, int ci_real,
                 int co_real, float* p_out, void* stream);

/* ---------------------------------------------------------------- B. point-wise family --- */

typedef struct RtpTerm {
  RtpAct t;          /* DIRECT: the addend.  GN: dxhat */
  const float* coeff; /* NULL = DIRECT term, else GN-backward coefficients [n][c][3] */
  int d, h, w;       /* spatial dims of the term (fuse_sum: low-res terms are upsampled) */
} RtpTerm;

/* Lazy GroupNorm-backward coefficients of a GN term of rtp_grad_combine_cls_lazy: computed in that kernel's prologue (the
 * arithmetic of rtp_gn_bwd_coeffs) from the statistics partials the data-gradient launch left behind; the term's `coeff`
 * buffer ([n*c*5] fp32, as rtp_gn_bwd_coeffs) is then an OUTPUT (coefficients + dgamma/dbeta partials, written once per
 * sample).  pq == NULL: the term is not lazy. */
typedef struct RtpGnLazy {
  const float* pq; int nsplit;     /* [n][nsplit][c][2]: P = sum dxhat, Q = sum dxhat * x */
  const float* mr; const float* gamma; int groups;
} RtpGnLazy;

/* GroupNorm-backward inputs from which rtp_conv_dgrad_fused computes its coefficients in its own prologue (no kernel of
 * their own between the weight gradient and the data gradient). */
typedef struct RtpGnBwd {
  const float* qpart; int q_nsplit; /* [n][q_nsplit][32] slab contractions (rtp_wgrad_q): Q = their sum                 */
  const float* p;                   /* [n][32]  P = sum dxhat, from the class sums (rtp_gn_bwd_p); or NULL with ...      */
  const float* tg;                  /* [n][q_nsplit][27][32] subset-sum partials from rtp_wgrad_q: P is computed from them */
  float* csum_out;                  /* (with tg) optional [n][64][32]: per-boundary-class sums of gy for rtp_wgrad_fold   */
  const float* csum;                /* or (stride-2 data gradients): [n][64][32] boundary-class sums of gy (rtp_class_sums)  */
  const float* mr;                  /* [n][groups][2] (mean, rstd) saved by rtp_fold_fwd                                 */
  const float* gamma; int groups;
  float* coeff_out;                 /* optional [n*32*5]: the coefficients + dgamma/dbeta partials, as rtp_gn_bwd_coeffs */
} RtpGnBwd;

/* Data gradient of a 3x3x3 stride-1 conv with 32 input channels that writes the FINISHED gradient of the conv's input x,
 * absorbing the gradient fan-in pass (rtp_grad_combine) into its epilogue:
 *   dx = [x > 0 if mask] * (A * conv_transpose(gy; wd) + B*x + C + sum_k term_k)
 * coeff: this conv's GroupNorm-backward coefficients [n][32][3], or NULL with gn != NULL (computed in the kernel from P and
 * the slab contractions), or both NULL (conv without GroupNorm: A = 1, B = C = 0);
 * terms (host, <= 3): gradient contributions of x's OTHER consumers, already complete -- DIRECT addends or another
 * GroupNorm consumer's dxhat with its coefficients (evaluated like rtp_grad_combine's GN terms).
 * Geometries: rtp_conv_tiled_ok(gy, g, 1) (stride 1), or the stride-2 convs of csrc/dgrad_s2_tiled.hip (32 -> <= 32 channels,
 * input dims = 2 x output dims, Ho % 2 == 0, Wo % 16 == 0; gn then carries csum instead of tg); others return
 * RTP_ERR_UNSUPPORTED (use rtp_conv_igemm + rtp_grad_combine).
 * tot_out (optional): per-channel sums of the stored dx, one fp32 partial per workgroup
 * [n][rtp_conv_stats_nsplit(gy, g, 1)][32], which rtp_class_sums_boundary completes to per-boundary-class sums. */
int rtp_conv_dgrad_fused(const RtpAct* gy, const void* wd, const RtpAct* x, const float* coeff, const RtpGnBwd* gn /*host*/,
                         const RtpTerm* terms /*host*/, int nterms, int mask, const RtpAct* dx, const RtpConvGeom* g,
                         float* tot_out, void* stream);

int rtp_conv_dgrad_fused_ok(const RtpAct* gy, const RtpConvGeom* g);

/* GroupNorm -> Conv3d(3x3x3, stride 1, 32 -> 16|32 channels) -> [+ residual] -> [ReLU] in ONE launch: the fold of the
 * input's GroupNorm into per-sample weights and the boundary-class bias table (rtp_fold_fwd) happens in the prologue of
 * every workgroup of the LDS-tiled conv kernel, from the statistics partials the producer of x left behind -- no fold launch
 * between two convs of a chain (create_conv's 'gcr' order, hr3d.py:58-61,91-93).  Same arithmetic as rtp_fold_fwd followed by
 * rtp_conv_igemm (fp32 weight * fp32 scale -> one bf16 rounding; group statistics summed in double).
 * w: the conv's fp32 weights in tap-major order [27][Co][32] (rtp_tail_desc_pack_wt of the master [co_real][32][27]); stats: [n][nsplit][32][2] (sum, sum of squares) of x;
 * mr (optional out): [n][groups][2] (mean, rstd) for the backward pass; stat_out as rtp_conv_igemm_stats.
 * Geometries: rtp_conv_tiled_ok(x, g, 0); others RTP_ERR_UNSUPPORTED. */
typedef struct RtpGnFold {
  const float* w; const float* bias; const float* gamma; const float* beta; const float* stats;
  int nsplit, groups, co_real; float eps; float* mr;
} RtpGnFold;
int rtp_conv_gn_fused(const RtpAct* x, const RtpGnFold* f /*host*/, const RtpAct* res, const RtpAct* y, const RtpConvGeom* g,
                      int relu, float* stat_out, void* stream);

/* rtp_grad_combine_cls with lazy[k] (host array of nterms, or NULL) describing terms whose coefficients the kernel computes. */
int rtp_grad_combine_cls_lazy(const RtpTerm* terms /*host*/, int nterms, const RtpGnLazy* lazy /*host*/, const RtpAct* x,
                              const RtpAct* relu_src, const RtpAct* out, int n, int d, int h, int w, int nsplit,
                              float* cls_scratch, void* stream);

/* out = mask(relu_src > 0) * sum_k term_k ; GN terms evaluate A*dxhat + B*x + C.  All same resolution. */
int rtp_grad_combine(const RtpTerm* terms /*host*/, int nterms, const RtpAct* x, const RtpAct* relu_src,
                     const RtpAct* out, int n, long vox, void* stream);
/* Same combine, plus per-boundary-class channel partial sums of the result written to cls_scratch fp32
 * [n][nsplit][64][c] (rows of (z,y) split nsplit ways; reduce with rtp_class_sums_reduce).  Saves the separate
 * rtp_class_sums scan of a gradient tensor this call produces.  c <= 64. */
int rtp_grad_combine_cls(const RtpTerm* terms /*host*/, int nterms, const RtpAct* x, const RtpAct* relu_src,
                         const RtpAct* out, int n, int d, int h, int w, int nsplit, float* cls_scratch, void* stream);

/* out = (relu) sum_k up(term_k): trilinear align_corners=True upsample for terms whose dims differ
 * (hr3d.py:205-229, hrnet3d.py:37-39). bias fp32 [c] or NULL. */
int rtp_fuse_sum(const RtpTerm* terms /*host*/, int nterms, const float* bias, const RtpAct* out, int n, int d, int h,
                 int w, int relu, void* stream);
/* The same row that also emits the per-channel statistics of what it stores (sum y, sum y^2; one partial per block,
 * stat_out fp32 [n][nsplit][c][2] with nsplit = rtp_fuse_stats_nsplit(n, c, d, h, w) > 0) -- the input of the next GroupNorm
 * without a read pass of its own (rtp_chan_stats). */
int rtp_fuse_stats_nsplit(int n, int c, int d, int h, int w);
int rtp_fuse_sum_stats(const RtpTerm* terms /*host*/, int nterms, const float* bias, const RtpAct* out, int n, int d, int h,
                       int w, int relu, float* stat_out, int nsplit, void* stream);

/* Adjoint of the trilinear upsample: glow[n][dl][hl][wl][c] = up^T(ghi), as three separable 1-D passes;
 * scratch fp32 [rtp_upsample_bwd_scratch_floats(...)]. */
int rtp_upsample_bwd(const RtpAct* ghi, int d, int h, int w, const RtpAct* glow, int dl, int hl, int wl, int n,
                     float* scratch, void* stream);
long rtp_upsample_bwd_scratch_floats(int n, int c, int d, int h, int w, int dl, int hl, int wl);

/* layer1.conv1 when Cin == 1 (common.py:111-113): y[v][c] = x[v]*w[c] + b[c]; x fp32 [n][vox]. */
int rtp_stem_fwd(const float* x, const float* w, const float* b, const RtpAct* y, int n, long vox, void* stream);
/* ... that also leaves the statistics the first GroupNorm (hr_util/common.py:57, applied to this output by layer1.conv2) needs:
 * stat_out [n][nsplit][c][2] = per-block partial (sum y, sum y^2) of the STORED bf16 values, nsplit = rtp_stem_stats_nsplit(n, c, vox)
 * (0: not offered for this channel count) -- what rtp_chan_stats computes in a read pass of its own. */
int rtp_stem_stats_nsplit(int n, int c, long vox);
int rtp_stem_fwd_stats(const float* x, const float* w, const float* b, const RtpAct* y, int n, long vox, float* stat_out, int nsplit,
                       void* stream);
/* dw[c] (+)= sum x*g, db[c] (+)= sum g ; scratch fp32 [nblk][c][2] with nblk = rtp_stem_bwd_blocks(). */
int rtp_stem_bwd(const float* x, const RtpAct* gy, int n, long vox, float* scratch, float* dw, float* db,
                 int accumulate, void* stream);
int rtp_stem_bwd_blocks(void);

/* NCDHW fp32 -> channels-last bf16 (network input with Cin in {32,64}) and back (feature export). */
int rtp_pack_ncdhw(const float* x, const RtpAct* y, int n, int c, long vox, void* stream);
int rtp_unpack_ncdhw(const RtpAct* x, float* y, int n, int c, long vox, void* stream);
/* fp32 channels-last [n][vox][x_cs] (channels [x_co, x_co + c)) -> fp32 NC(D)HW [n][c][vox]. */
int rtp_unpack_ncdhw_f32(const float* x, int x_cs, int x_co, float* y, int n, int c, long vox, void* stream);
/* y = bf16([relu](x (+ x2))): fp32 NC(D)HW -> channels-last, with an optional second addend and ReLU -- the hand-off from the
 * fp32 deformable-convolution operator (section D) back into the plan (FeatureAdaption: relu(conv_adaption(x, offset)),
 * det3d/models/pose_heads/center_head.py:59-62; backward: grad_input of the two paths summed). */
int rtp_pack_ncdhw_ex(const float* x, const float* x2, const RtpAct* y, int n, int c, long vox, int relu, void* stream);

/* --- Deformable convolution forward on the plan's layout (csrc/dcn_cl.hip; round 5): the deformable half of FeatureAdaption
 * (center_head.py:24-62) per (frame, z) slice without the hand-off to the fp32 NCHW operator of section D.
 *   x   bf16 channels-last [n_img][h][w][>= 32]   (n_img = frames * Z: the 5-D feature with Z folded into the batch)
 *   off fp32 channels-last [n_img][h][w][>= 72]   (RtpAct over floats: 4 deformable groups x 9 taps x (dh, dw), the order of
 *                                                  deform_conv_cuda_kernel.cu:190-243)
 *   w   fp32 [32][32][3][3] (DeformConv.weight, no bias);  y bf16 channels-last [n_img][h][w][>= 32], ReLU when relu != 0.
 * DCNv1, 3x3, stride 1, padding 1, dilation 1, 32 -> 32 channels, 4 deformable groups; h * w a multiple of 16.  Sampling rule and
 * per-corner bounds as the section-D operator; the samples and the weights enter the product as bf16 (fp32 accumulation), like the
 * operands of every other conv of the plan -- rtp_deform_conv_forward keeps fp32 operands.  No allocation. */
int rtp_dcn_cl_forward(const RtpAct* x, const RtpAct* off, const float* w, const RtpAct* y, int n_img, int h, int w_, int relu,
                       void* stream);

/* ---------------------------------------------------------------- C. head: loss / decode / optimiser --- */

/* FastFocalLoss forward+backward (centernet_loss.py:34-54, center_head.py:240-242).
 *   logits fp32 [n][vox][cpad]; target fp32 NCDHW [n][ncls][vox]; ind/cat int64 [n][m]; mask uint8 [n][m]
 *   scratch fp32 [rtp_focal_blocks()]; out_loss[0] = hm_loss (fp32, device); ghm bf16 view: d(gscale*hm_loss)/dlogits */
int rtp_focal_loss(const float* logits, int cpad, const float* target, const long long* ind, const unsigned char* mask,
                   const long long* cat, int n, int ncls, long vox, int m, float gscale, float* scratch,
                   float* out_loss, const RtpAct* ghm, void* stream);
/* write_pad == 0: padding channels past the last class chunk of the gradient rows are not stored (buffer zeroed once, written by
 * nothing else). */
int rtp_focal_loss_ex(const float* logits, int cpad, const float* target, const long long* ind, const unsigned char* mask,
                      const long long* cat, int n, int ncls, long vox, int m, float gscale, float* scratch, float* out_loss,
                      const RtpAct* ghm, int write_pad, void* stream);
int rtp_focal_blocks(void);

/* RegLoss forward+backward (centernet_loss.py:17-24, center_head.py:252-258).
 *   reg fp32 [n][vox][cpad]; target fp32 [n][m][nreg]; code_w fp32 [nreg]
 *   out[0..nreg) = per-channel loss, out[nreg] = loc_loss; greg (bf16, pre-zeroed by this call) gets
 *   gscale * weight * d loc_loss / d reg. */
int rtp_reg_loss(const float* reg, int cpad, const float* target, const long long* ind, const unsigned char* mask,
                 const float* code_w, int n, int nreg, long vox, int m, float gscale, float* out,
                 const RtpAct* greg, void* stream);
/* rtp_reg_loss without the zero fill of greg: prev_ind (device int64 [n*m], initialised to -1; greg zero-initialised once and
 * written by nothing else) remembers the voxels of the previous call, the only non-zero ones: they are cleared, this call's are
 * written and recorded. */
int rtp_reg_loss_sparse(const float* reg, int cpad, const float* target, const long long* ind, const unsigned char* mask,
                        const float* code_w, int n, int nreg, long vox, int m, float gscale, float* out, const RtpAct* greg,
                        long long* prev_ind, void* stream);

/* sigmoid -> per-channel first-index argmax -> offset decode (center_head.py:272-360).
 *   scale_xyz / origin_xyz: HOST fp32[3] = (out_size_factor*voxel_size, pc_range) per x,y,z
 *   scratch fp32 [rtp_decode_scratch_floats(n,ncls)]
 *   out fp32 [n][ncls][2 + nreg] = (argmax voxel index, score, x0,y0,z0, x1,...) */
int rtp_decode(const float* logits, int hm_cpad, const float* reg, int reg_cpad, int n, int ncls, int nreg, int d,
               int h, int w, const float* scale_xyz, const float* origin_xyz, float* scratch, float* out,
               void* stream);
int rtp_decode_scratch_floats(int n, int ncls);

/* ---------------------------------------------------------------- C'. input pipeline --- */

/* Raw radar cubes -> network input (det3d/datasets/cruw_pose/cruw_pose.py:167-194 get_cube / get_cube_phase, plus the
 * channel-axis rule of det3d/datasets/pipelines/pose.py:163-170).
 *   cube_f16  device fp16 [lead][zs][ys][xs]: `lead` = frames x leading channels (1, D, or 2*D for the phase cubes)
 *   roi_zyx   HOST int[6] = inclusive index ranges (z0,z1,y0,y1,x0,x1) (CRUW_POSE_Dataset.consider_roi_cube)
 *   normalise != 0: out = max(0, (x - norm_lo) / (norm_hi - norm_lo)) in fp32 (get_cube); 0: crop only (get_cube_phase)
 *   out       device fp32 [lead][Z][Y][X] -- for a batch this IS the NCDHW network input */
int rtp_cube_prep(const void* cube_f16, long lead, int zs, int ys, int xs, const int* roi_zyx, float norm_lo,
                  float norm_hi, int normalise, float* out, void* stream);

/* gaussian3D((2r+1,)*3, sigma=(2r+1)/6) of det3d/core/utils/center_utils.py:67-72 in float64, rounded to fp32, into HOST
 * memory out_host[(2r+1)^3] (upload it once; it is the `table` of rtp_assign_labels). */
int rtp_gaussian_table(int radius, float* out_host);

/* CenterNet-style label assignment on the device (pipelines/pose.py:186-254 AssignLabelPose when one_hm == 0,
 * :386-451 AssignLabelPose2 when one_hm != 0), one task.
 *   poses   device fp64 [frames][max_in][15][3] (x,y,z metres), nposes device int [frames] (poses present per frame)
 *   M = 15*max_poses slots (one_hm == 0: slot k = key-point k%15 of pose k/15, class = key-point) or max_poses slots
 *       (one_hm != 0: slot k = pose k, centre = key-point 0, class 0)
 *   hm      device fp32 [frames][15|1][fz][fy][fx]: max-splat of the gaussian table (radius r) at each in-range centre.
 *           NOT memset here: `prev` (device int [frames][M][4], zero-initialised by the caller once) records the boxes
 *           written by the previous call, which are cleared first.
 *   anno    device fp32 [frames][M][3|45], ind/cat device int64 [frames][M], mask device uint8 [frames][M]
 * Voxel coordinate (x - radar_range[k]) / voxel_size[k] / out_size_factor[k] (pose.py:222-227; range_min_zyx must hold the
 * fp32-ROUNDED bounds, the reference keeps them in an np.float32 array, :190): numpy_legacy == 0 evaluates every step in
 * fp32 as NumPy >= 2 does (NEP 50; what the captured vectors pin), != 0 keeps NumPy 1.x's float64 intermediate with one
 * rounding to fp32.  The two differ by <= 1 ulp, which decides the integer voxel of a key-point on a voxel boundary.
 * A key-point whose voxel lies outside the map keeps its slot with ind = mask = cat = 0 (the reference's `continue`).
 * The reference raises IndexError when 0 < poses present < max_poses (15-map variant); the host wrapper checks that. */
int rtp_assign_labels(const double* poses, const int* nposes, int frames, int max_in, int max_poses, int one_hm,
                      int radius, const double* range_min_zyx, const double* voxel_size_xyz,
                      const int* out_size_factor_zyx, int fz, int fy, int fx, const float* table, float* hm, float* anno,
                      long long* ind, unsigned char* mask, long long* cat, int* prev, int numpy_legacy, void* stream);

/* ---------------------------------------------------------------- C''. LiDAR stream --- */

/* points[:, :3] = (P_L2R @ [x, y, z, 1])[:3] in float64, rounded once to fp32, in place
 * (det3d/datasets/pipelines/pose.py:34-38).  points device fp32 [n][c]; P_L2R HOST double[12] = first three rows. */
int rtp_lidar_transform(float* points, int n, int c, const double* P_L2R, void* stream);

/* Dynamic voxelisation (det3d/models/readers/dynamic_voxel_encoder.py:8-19 `voxelization` + core/utils/scatter.py):
 *   keep points with min <= p <= max on x, y, z (both ends inclusive); voxel coordinate = trunc((p - min) / voxel_size) in
 *   fp32; voxels = the distinct (z, y, x) coordinates in ascending lexicographic order (torch.unique(dim=0));
 *   voxels[v][:] = mean over the voxel's points (summed in input order) of ALL c point features.
 *   points device fp32 [n][c]; pc_range HOST fp32[6] = (xmin,ymin,zmin,xmax,ymax,zmax); voxel_size HOST fp32[3] = (x,y,z)
 *   voxels device fp32 [n][c] (capacity), coords device int64 [n][3] = (z,y,x), num_voxels device int
 *   workspace: rtp_voxelize_workspace_bytes(n) bytes of device memory (keys, permutation, rocPRIM temporaries).
 * Stable radix sort by voxel key + one sequential sum per (voxel, feature): deterministic, bit-identical to the reference's
 * CPU arithmetic (its GPU path sums with atomics). */
long rtp_voxelize_workspace_bytes(int n);
int rtp_dynamic_voxelize(const float* points, int n, int c, const float* pc_range, const float* voxel_size, float* voxels,
                         long long* coords, int* num_voxels, void* workspace, long ws_bytes, void* stream);

/* Voxel means scattered into a dense grid for fusion with the radar feature (this repo's own step; the reference has no
 * fusion detector): grid fp32 [Z][Y][X][c] (zeroed here), occ uint8 [Z][Y][X] or NULL; voxels outside the grid (a point
 * exactly on the upper range bound) are dropped.  max_voxels bounds the launch; the device count decides. */
int rtp_voxels_to_dense(const float* voxels, const long long* coords, const int* num_voxels, int max_voxels, int c, int Z,
                        int Y, int X, float* grid, unsigned char* occ, void* stream);

/* Global grad-norm partials and the fused clip + decoupled-weight-decay + Adam step over a flat fp32
 * parameter buffer (fastai_optim.py:154-172, hooks/optimizer.py:14-24, torch.optim.Adam).
 *   hyper (device fp32[10]) = {lr, beta1, beta2, eps, wd, max_norm, 1-beta1^t, 1-beta2^t, grad_scale, -}
 *   mode 0: clip+decay+Adam;  mode 1: decay only (parameters that received no gradient);
 *   norm_out (device, optional) receives the pre-clip global L2 norm. */
int rtp_sqnorm(const float* g, long n, const float* hyper, float* partial /*[rtp_sqnorm_blocks()]*/, void* stream);
int rtp_sqnorm_blocks(void);
int rtp_adam_step(float* p, const float* g, float* m, float* v, long n, const float* hyper,
                  const float* sqnorm_partial, int mode, float* norm_out, void* stream);

/* ---------------------------------------------------------------- D. deformable convolution --- */
/* One-for-one replacements of deform_conv_cuda.cpp:152-157, 262-268, 376-381, 490-496, 571-578.
 * Tensors are contiguous NCHW fp32 as in the reference; `columns`/`ones` scratch is replaced by
 * an explicit workspace (size from rtp_dcn_workspace_bytes, n = im2col_step images; opaque scratch: transposed
 * weights for the fused forward, columns, the column gradient in its own layout -- contents are undefined after a
 * call, 16-byte aligned base required).  Argument order keeps the reference's
 * (kW,kH,dW,dH,padW,padH,dilW,dilH) W-before-H convention. */
long rtp_dcn_workspace_bytes(int n, int c, int h, int w, int co, int kh, int kw, int ho, int wo);
int rtp_deform_conv_forward(const float* input, const float* weight, const float* offset, float* output, void* ws,
                            int n, int c, int h, int w, int co, int kW, int kH, int dW, int dH, int padW, int padH,
                            int dilW, int dilH, int group, int deformable_group, int im2col_step, void* stream);
int rtp_deform_conv_backward_input(const float* input, const float* offset, const float* gradOutput, float* gradInput,
                                   float* gradOffset, const float* weight, void* ws, int n, int c, int h, int w,
                                   int co, int kW, int kH, int dW, int dH, int padW, int padH, int dilW, int dilH,
                                   int group, int deformable_group, int im2col_step, void* stream);
int rtp_deform_conv_backward_parameters(const float* input, const float* offset, const float* gradOutput,
                                        float* gradWeight, void* ws, int n, int c, int h, int w, int co, int kW, int kH,
                                        int dW, int dH, int padW, int padH, int dilW, int dilH, int group,
                                        int deformable_group, float scale, int im2col_step, void* stream);
/* Both halves of DeformConvFunction.backward (det3d/ops/dcn/deform_conv.py:62-98: deform_conv_backward_input_cuda, then
 * deform_conv_backward_parameters_cuda on the same tensors) in one call.  For the DCN head's geometry (3x3, stride 1, pad 1,
 * 32 channels in 4 deformable groups, <= 32 output channels) one kernel produces all three gradients without a column matrix;
 * any other geometry runs the two entry points above one after the other.
 * Outputs: gradInput and gradWeight are ACCUMULATED into (the reference's contract: the caller zero-fills them, deform_conv.py:77-80);
 * gradOffset is ASSIGNED -- every element is written exactly once, nothing is read back (the reference zero-fills it and its
 * col2im_coord kernel then assigns it too, deform_conv_cuda_kernel.cu:436-507, so callers see the same values).
 * Precision of the one-kernel route: its two matrix products (W^T * gradOutput and gradOutput * columns^T) run on the bf16 matrix
 * core with both operands split into (hi, lo) bf16 pairs and the lo * lo product dropped: 3.9e-6 norm-wise against the exact fp32
 * products of the reference operator and of the two-call route -- not bit-equal to either.  RTP_DCN_FP32_MFMA=1 (environment, read
 * once) selects the fp32 matrix instruction instead (exact fp32 products, 1.4x the time); RTP_DCN_NO_FUSED_BWD=1 the two-call route. */
int rtp_deform_conv_backward(const float* input, const float* offset, const float* gradOutput, float* gradInput,
                             float* gradOffset, const float* weight, float* gradWeight, void* ws, int n, int c, int h,
                             int w, int co, int kW, int kH, int dW, int dH, int padW, int padH, int dilW, int dilH,
                             int group, int deformable_group, float scale, int im2col_step, void* stream);
/* The same for callers that own fresh gradient buffers: the three outputs are OVERWRITTEN (no zero-initialisation needed). */
int rtp_deform_conv_backward_overwrite(const float* input, const float* offset, const float* gradOutput, float* gradInput,
                                       float* gradOffset, const float* weight, float* gradWeight, void* ws, int n, int c,
                                       int h, int w, int co, int kW, int kH, int dW, int dH, int padW, int padH, int dilW,
                                       int dilH, int group, int deformable_group, float scale, int im2col_step, void* stream);
int rtp_modulated_deform_conv_forward(const float* input, const float* weight, const float* bias, const float* offset,
                                      const float* mask, float* output, void* ws, int n, int c, int h, int w, int co,
                                      int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int group,
                                      int deformable_group, int with_bias, void* stream);
int rtp_modulated_deform_conv_backward(const float* input, const float* weight, const float* bias, const float* offset,
                                       const float* mask, float* grad_input, float* grad_weight, float* grad_bias,
                                       float* grad_offset, float* grad_mask, const float* grad_output, void* ws, int n,
                                       int c, int h, int w, int co, int kh, int kw, int sh, int sw, int ph, int pw,
                                       int dh, int dw, int group, int deformable_group, int with_bias, void* stream);

/* ---------------------------------------------------------------- E. diagnostics --- */
/* Per-kernel-family HIP-event timing on the launch stream (bench.py roofline leg). */
int rtp_prof_enable(int family, int on);
int rtp_prof_collect(int family, float* total_ms, int* launches); /* synchronises the recorded events */
const char* rtp_version(void);
/* Several independent launches of ONE LDS-tiled kernel variant as one launch (csrc/rtp_multi.h; round 4).  HRNet's branches run the
 * same block structure side by side: the full-resolution conv of a stage and the level-1 conv of the same position are launches
 * of the same kernel on the same samples, and alone the small one occupies every CU for a ninth of the work.
 *   rtp_multi_begin();  <call the ordinary entry points of the launches to merge: rtp_conv_gn_fused / rtp_conv_igemm* /
 *   rtp_conv_dgrad_fused / rtp_wgrad* on the tiled kernels -- they validate and RECORD instead of launching>;
 *   rtp_multi_end(dev_params, rtp_multi_param_bytes(), &h);  then rtp_multi_launch(h, stream) per step;  rtp_multi_free(h) at the end.
 *   dev_params: device memory of rtp_multi_param_bytes() bytes that the CALLER owns and keeps alive until rtp_multi_free (the
 *   library allocates nothing; the blocks are uploaded inside rtp_multi_end with a synchronous copy).
 *   Results and buffers are those of the separate launches (a problem merely runs on its share of every sample's 256 / n
 *   workgroups).  rtp_multi_end returns RTP_ERR_UNSUPPORTED when the recorded launches cannot share one (different kernels or
 *   variants, different sample counts n or an n that does not divide 256, a generic-kernel geometry, fewer than 2 or more than 4
 *   launches): the caller keeps the separate launches.  An EMPTY capture left open on the thread is replaced by the next
 *   rtp_multi_begin; one with recorded launches (a nested begin, or a caller that failed without rtp_multi_abort) is discarded and
 *   that begin returns RTP_ERR_UNSUPPORTED -- no mixed set is ever merged.  rtp_multi_launch returns RTP_ERR_UNSUPPORTED on another
 *   device than the handle's.  rtp_multi_free: no launch of the handle may still be in flight.
 *   The default plan uses it for the two head towers (center_head.py:66-109: hm and reg are independent chains of the same convs):
 *   conv .0 / .2, data gradient .2 and both weight gradients run pairwise in one launch (-1.4 % on the hr3d step). */
int rtp_multi_begin(void);
long rtp_multi_param_bytes(void);
int rtp_multi_end(void* dev_params, long dev_bytes, int* handle_out);
int rtp_multi_abort(void);
int rtp_multi_launch(int handle, void* stream);
int rtp_multi_free(int handle);

/* Ordering events between streams of ONE device (the lane plan's cross-lane dependencies, rt_pose_amd/lanes.py).  system_fence = 0:
 * hipEventDisableTiming | hipEventDisableSystemFence -- recording the event does not write back and invalidate the caches for the
 * host (what an ordinary event does after every kernel it follows); a kernel on another stream of the same device that waits for
 * it is ordered behind the producer by the dispatch packets' own agent-scope release / acquire.  Do NOT use such an event to make
 * device memory visible to the host or to another GPU.  system_fence = 1: an ordinary event (A/B).  Handles are hipEvent_t. */
int rtp_event_create(void** ev_out, int system_fence);
int rtp_event_destroy(void* ev);
int rtp_event_record(void* ev, void* stream);
int rtp_stream_wait_event(void* stream, void* ev);

#ifdef __cplusplus
}
#endif
#endif /* RTP_H_ */
