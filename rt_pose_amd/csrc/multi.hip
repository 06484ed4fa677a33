// rtp_multi_begin / _end / _launch: several independent launches of one kernel variant as one launch (rtp_multi.h).
#include <stdlib.h>

#include <mutex>

#include "rtp_common.h"
#include <stdio.h>
#include <stdlib.h>
#include "rtp_multi.h"
#include "rtp_prof.h"

namespace {
thread_local std::vector<RtpMultiJob>* t_capture = nullptr;
struct Handle { int kind; void* launcher; int fam; };
std::mutex g_mu;
std::vector<Handle> g_handles;
}  // namespace

std::vector<RtpMultiJob>* rtp_multi_capture() { return t_capture; }

extern "C" int rtp_multi_begin(void) {
  if (t_capture) return RTP_ERR_UNSUPPORTED;   // no nesting
  t_capture = new std::vector<RtpMultiJob>();
  return RTP_OK;
}

// Closes the capture.  handle_out >= 0 on success.  RTP_ERR_UNSUPPORTED: the recorded launches cannot share a launch (different
// kernels / variants, not eight samples, a share larger than a problem's partial slots, fewer than two launches recorded) -- the
// caller keeps issuing them one by one.
extern "C" int rtp_multi_end(int* handle_out) {
  if (!t_capture || !handle_out) return RTP_ERR_SHAPE;
  std::vector<RtpMultiJob> jobs;
  jobs.swap(*t_capture);
  delete t_capture;
  t_capture = nullptr;
  *handle_out = -1;
  const int nj = (int)jobs.size();
  if (nj < 2 || nj > 4) return RTP_ERR_UNSUPPORTED;
  long total = 0;
  if (getenv("RTP_MERGE_DEBUG"))
    for (const RtpMultiJob& j : jobs)
      fprintf(stderr, "[multi] job kind %d variant %d n %d shm %zu tiles %d slots %d\n", j.kind, j.variant, j.n, (size_t)j.shm, j.tiles_per_sample, j.slots_per_sample);
  for (const RtpMultiJob& j : jobs) {
    if (j.kind != jobs[0].kind || j.variant != jobs[0].variant || j.n != 8 || j.shm != jobs[0].shm || j.tiles_per_sample < 1) return RTP_ERR_UNSUPPORTED;
    total += j.tiles_per_sample;
  }
  // shares of an XCD's 32 workgroups: proportional to the bricks, at least one each, largest remainders first
  // workgroups per XCD the shared launch is dealt over (experiments: RTP_MULTI_WGS_PER_XCD < 32 leaves CUs to the other lanes)
  static const int wg_env = getenv("RTP_MULTI_WGS_PER_XCD") ? atoi(getenv("RTP_MULTI_WGS_PER_XCD")) : 32;
  const int WG = wg_env < nj ? nj : (wg_env > 32 ? 32 : wg_env);
  int share[4] = {0, 0, 0, 0}, used = 0;
  double frac[4];
  for (int k = 0; k < nj; ++k) {
    const double ex = (double)WG * jobs[k].tiles_per_sample / (double)total;
    share[k] = (int)ex < 1 ? 1 : (int)ex;
    frac[k] = ex - share[k];
    used += share[k];
  }
  while (used < WG) { int b = 0; for (int k = 1; k < nj; ++k) if (frac[k] > frac[b]) b = k; ++share[b]; frac[b] -= 1.0; ++used; }
  while (used > WG) { int b = -1; for (int k = 0; k < nj; ++k) if (share[k] > 1 && (b < 0 || frac[k] < frac[b])) b = k; if (b < 0) return RTP_ERR_UNSUPPORTED; --share[b]; frac[b] += 1.0; --used; }
  for (int k = 0; k < nj; ++k)
    if (share[k] > jobs[k].slots_per_sample || 2L * share[k] > jobs[k].tiles_per_sample) return RTP_ERR_UNSUPPORTED;
  size_t bytes = 0;
  for (const RtpMultiJob& j : jobs) bytes += (j.params.size() + 255) / 256 * 256;
  void* dev = nullptr;
  if (hipMalloc(&dev, bytes) != hipSuccess) return RTP_ERR_LAUNCH;
  void* launcher = nullptr;
  int rc = RTP_ERR_UNSUPPORTED;
  if (jobs[0].kind == RTP_MULTI_CONV_TILED) rc = rtp_conv_tiled_multi_finish(jobs, share, dev, &launcher);
  else if (jobs[0].kind == RTP_MULTI_WGRAD_TILED) rc = rtp_wgrad_tiled_multi_finish(jobs, share, dev, &launcher);
  if (rc != RTP_OK) { (void)hipFree(dev); return rc; }
  std::lock_guard<std::mutex> lk(g_mu);
  g_handles.push_back(Handle{jobs[0].kind, launcher, jobs[0].fam});
  *handle_out = (int)g_handles.size() - 1;
  return RTP_OK;
}

// Drops an open capture without building anything (error paths of the caller).
extern "C" int rtp_multi_abort(void) {
  if (t_capture) { delete t_capture; t_capture = nullptr; }
  return RTP_OK;
}

extern "C" int rtp_multi_launch(int handle, void* stream) {
  Handle h;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (handle < 0 || handle >= (int)g_handles.size()) return RTP_ERR_SHAPE;
    h = g_handles[handle];
  }
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(h.fam, s);
  if (h.kind == RTP_MULTI_CONV_TILED) return rtp_conv_tiled_multi_launch(h.launcher, s);
  if (h.kind == RTP_MULTI_WGRAD_TILED) return rtp_wgrad_tiled_multi_launch(h.launcher, s);
  return RTP_ERR_SHAPE;
}
