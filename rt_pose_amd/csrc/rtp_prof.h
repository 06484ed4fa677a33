// Optional per-kernel-family timing with HIP events recorded on the launch stream (bench.py roofline leg).
#pragma once
#include <hip/hip_runtime.h>

enum RtpFamily {
  RTP_FAM_CONV = 0,      // rtp_conv_igemm (generic gather kernel)
  RTP_FAM_CONV_TILED = 1,  // LDS-tiled full-resolution conv
  RTP_FAM_WGRAD = 2,
  RTP_FAM_POINTWISE = 3,
  RTP_FAM_NORM = 4,
  RTP_FAM_LOSS = 5,
  RTP_FAM_OPTIM = 6,
  RTP_FAM_DCN = 7,
  RTP_FAM_WGRAD_TILED = 8,
  RTP_FAM_CONV_TILED_FULL = 9,  // the LDS-tiled conv at its dominant geometry: 32 -> 32 channels, >= 2^20 output voxels per launch
  RTP_FAM_CONV_TILED_FULL_BWD = 10,  // ... its data-gradient launches (flipped taps; the fused variants carry the fan-in / GroupNorm-backward epilogue)
  RTP_FAM_CONV64 = 11,  // csrc/conv64_tiled.hip: the 64-wide stride-1 kernel (64 -> 64 layers, paired head towers)
  RTP_FAM_COUNT = 12
};

void rtp_prof_begin(int fam, hipStream_t s);
void rtp_prof_end(int fam, hipStream_t s);
extern int g_rtp_prof_on[RTP_FAM_COUNT];

struct RtpProfScope {
  int fam;
  hipStream_t s;
  bool on;
  RtpProfScope(int f, hipStream_t st) : fam(f), s(st), on(g_rtp_prof_on[f] != 0) {
    // Every entry point constructs one of these right before its launches: clear any stale "last error" left by
    // other HIP users in the process (PyTorch's event queries leave hipErrorNotReady behind) so that
    // RTP_CHECK_LAUNCH reports only this call's launches.
    (void)hipGetLastError();
    if (on) rtp_prof_begin(fam, s);
  }
  ~RtpProfScope() {
    if (on) rtp_prof_end(fam, s);
  }
};
