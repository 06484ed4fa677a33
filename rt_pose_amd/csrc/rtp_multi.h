// Horizontal fusion of independent launches of ONE kernel variant into one launch (round 4).
//
// The plan's launches are host closures over the ordinary entry points (rtp_conv_gn_fused, rtp_conv_dgrad_fused, rtp_wgrad_q, ...).
// Between rtp_multi_begin() and rtp_multi_end() those entry points do everything they always do -- validation, parameter block -- but
// RECORD the launch instead of issuing it; rtp_multi_end() checks that the recorded launches are the same kernel variant on the same
// number of samples n (a divisor of 256: a sample's 256 / n workgroups are one contiguous run of the XCD-aware workgroup order, so
// for n = 8 a sample is an XCD), splits every sample's workgroups between them in proportion to their bricks, uploads the parameter
// blocks into memory the CALLER provides and returns a handle that rtp_multi_launch() issues as ONE kernel.  Nothing about the problems' buffers or results changes: a problem just runs
// on fewer workgroups per sample than alone, beside the others.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

enum RtpMultiKind { RTP_MULTI_NONE = 0, RTP_MULTI_CONV_TILED = 1, RTP_MULTI_WGRAD_TILED = 2 };

struct RtpMultiJob {
  int kind;                 // RtpMultiKind
  int variant;              // kernel variant key within the kind (all jobs of a handle must agree)
  long tiles_per_sample;    // bricks of one sample: the share of an XCD's workgroups is proportional to it
  int n;                    // samples (a divisor of 256; all jobs of a handle agree)
  int slots_per_sample;     // per-workgroup partial slots the problem's buffers hold per sample (its share must not exceed them)
  size_t shm;               // dynamic LDS bytes
  int fam;                  // profiling family
  std::vector<char> params; // the kernel's parameter block (TiledParams / WgTiledParams), patched with the share at rtp_multi_end
};

#define RTP_MULTI_MAX 4            // problems per shared launch
#define RTP_MULTI_PARAM_SLOT 1024  // bytes of device memory per problem's parameter block (TiledParams / WgTiledParams fit)

// non-null while a capture is open on this thread: entry points append their launch and return RTP_OK without launching
std::vector<RtpMultiJob>* rtp_multi_capture();

// per-kind hooks, defined beside the kernels
int rtp_conv_tiled_multi_finish(std::vector<RtpMultiJob>& jobs, const int* share, void* dev_params, void** launcher);
int rtp_conv_tiled_multi_launch(void* launcher, hipStream_t s);
void rtp_conv_tiled_multi_drop(void* launcher);
int rtp_wgrad_tiled_multi_finish(std::vector<RtpMultiJob>& jobs, const int* share, void* dev_params, void** launcher);
int rtp_wgrad_tiled_multi_launch(void* launcher, hipStream_t s);
void rtp_wgrad_tiled_multi_drop(void* launcher);
