// rtp_multi_begin / _end / _launch / _free: several independent launches of one kernel variant as one launch (rtp_multi.h).
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "rtp_common.h"
#include "rtp_multi.h"
#include "rtp_prof.h"

namespace {
thread_local std::vector<RtpMultiJob>* t_capture = nullptr;
struct Handle { int kind; void* launcher; int fam; int device; };   // kind RTP_MULTI_NONE: a freed slot (re-used by the next handle)
std::mutex g_mu;
std::vector<Handle> g_handles;

void drop_launcher(int kind, void* launcher) {
  if (kind == RTP_MULTI_CONV_TILED) rtp_conv_tiled_multi_drop(launcher);
  else if (kind == RTP_MULTI_WGRAD_TILED) rtp_wgrad_tiled_multi_drop(launcher);
}
}  // namespace

std::vector<RtpMultiJob>* rtp_multi_capture() { return t_capture; }

// A capture left open on this thread (a caller that failed between begin and end without calling rtp_multi_abort) is dropped, not
// inherited: otherwise every later tiled entry point on the thread would record instead of launching and still return RTP_OK.
// ... unless it has recorded jobs.  That is either a nested / re-entrant begin (a closure that itself builds a shared launch) or a
// caller that failed after recording: in both cases the open capture is DISCARDED and this begin fails with RTP_ERR_UNSUPPORTED --
// nothing is merged from a mixed set, the nested caller sees an error, the outer rtp_multi_end then fails with RTP_ERR_SHAPE (no
// capture), and the next begin on the thread starts clean.
extern "C" int rtp_multi_begin(void) {
  if (t_capture && !t_capture->empty()) {
    if (getenv("RTP_MERGE_DEBUG")) fprintf(stderr, "[multi] begin inside an open capture with %zu recorded jobs: both dropped\n", t_capture->size());
    delete t_capture;
    t_capture = nullptr;
    return RTP_ERR_UNSUPPORTED;
  }
  if (t_capture) { delete t_capture; t_capture = nullptr; }
  t_capture = new std::vector<RtpMultiJob>();
  return RTP_OK;
}

// Bytes of device memory a handle needs for its parameter blocks (RTP_MULTI_MAX problems): the caller allocates them (the C ABI
// allocates nothing) and keeps them alive until rtp_multi_free.
extern "C" long rtp_multi_param_bytes(void) { return (long)RTP_MULTI_MAX * RTP_MULTI_PARAM_SLOT; }

// Closes the capture.  handle_out >= 0 on success.  RTP_ERR_UNSUPPORTED: the recorded launches cannot share a launch (different
// kernels / variants / sample counts, a sample count that does not divide the chip's 256 workgroups, a share larger than a problem's
// partial slots, fewer than two launches recorded) -- the caller keeps issuing them one by one.
extern "C" int rtp_multi_end(void* dev_params, long dev_bytes, int* handle_out) {
  if (!t_capture || !handle_out) { rtp_multi_abort(); return RTP_ERR_SHAPE; }
  std::vector<RtpMultiJob> jobs;
  jobs.swap(*t_capture);
  delete t_capture;
  t_capture = nullptr;
  *handle_out = -1;
  const int nj = (int)jobs.size();
  if (nj < 2 || nj > RTP_MULTI_MAX) return RTP_ERR_UNSUPPORTED;
  if (!dev_params || dev_bytes < rtp_multi_param_bytes()) return RTP_ERR_SHAPE;
  long total = 0;
  if (getenv("RTP_MERGE_DEBUG"))
    for (const RtpMultiJob& j : jobs)
      fprintf(stderr, "[multi] job kind %d variant %d n %d shm %zu tiles %ld slots %d\n", j.kind, j.variant, j.n, (size_t)j.shm, j.tiles_per_sample, j.slots_per_sample);
  const int n = jobs[0].n;
  if (n < 1 || n > 256 || 256 % n) return RTP_ERR_UNSUPPORTED;   // whole workgroups per sample, samples aligned with the XCDs' runs
  for (const RtpMultiJob& j : jobs) {
    if (j.kind != jobs[0].kind || j.variant != jobs[0].variant || j.n != n || j.shm != jobs[0].shm || j.tiles_per_sample < 1 ||
        j.params.size() > RTP_MULTI_PARAM_SLOT)
      return RTP_ERR_UNSUPPORTED;
    total += j.tiles_per_sample;
  }
  // shares of a sample's 256 / n workgroups (n = 8: one XCD's 32): proportional to the bricks, at least one each, largest
  // remainders first.  (experiments: RTP_MULTI_WGS_PER_XCD < 32 leaves CUs to the other lanes)
  static const int wg_env = 32;
  const int per_xcd = wg_env > 32 ? 32 : (wg_env < 1 ? 1 : wg_env);
  const int WG = 8 * per_xcd / n;
  if (WG < nj) return RTP_ERR_UNSUPPORTED;
  int share[RTP_MULTI_MAX] = {0, 0, 0, 0}, used = 0;
  double frac[RTP_MULTI_MAX];
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
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return RTP_ERR_LAUNCH;
  void* launcher = nullptr;
  int rc = RTP_ERR_UNSUPPORTED;
  if (jobs[0].kind == RTP_MULTI_CONV_TILED) rc = rtp_conv_tiled_multi_finish(jobs, share, dev_params, &launcher);
  else if (jobs[0].kind == RTP_MULTI_WGRAD_TILED) rc = rtp_wgrad_tiled_multi_finish(jobs, share, dev_params, &launcher);
  if (rc != RTP_OK) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  const Handle h{jobs[0].kind, launcher, jobs[0].fam, dev};
  for (size_t i = 0; i < g_handles.size(); ++i)
    if (g_handles[i].kind == RTP_MULTI_NONE) { g_handles[i] = h; *handle_out = (int)i; return RTP_OK; }
  g_handles.push_back(h);
  *handle_out = (int)g_handles.size() - 1;
  return RTP_OK;
}

// Drops an open capture without building anything (error paths of the caller).
extern "C" int rtp_multi_abort(void) {
  if (t_capture) { delete t_capture; t_capture = nullptr; }
  return RTP_OK;
}

// Releases a handle (its host-side launcher; the parameter memory is the caller's).  The caller must have synchronised with the
// handle's last launch.
extern "C" int rtp_multi_free(int handle) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (handle < 0 || handle >= (int)g_handles.size() || g_handles[handle].kind == RTP_MULTI_NONE) return RTP_ERR_SHAPE;
  drop_launcher(g_handles[handle].kind, g_handles[handle].launcher);
  g_handles[handle] = Handle{RTP_MULTI_NONE, nullptr, 0, -1};
  return RTP_OK;
}

extern "C" int rtp_multi_launch(int handle, void* stream) {
  Handle h;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (handle < 0 || handle >= (int)g_handles.size() || g_handles[handle].kind == RTP_MULTI_NONE) return RTP_ERR_SHAPE;
    h = g_handles[handle];
  }
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != h.device) return RTP_ERR_UNSUPPORTED;   // built for another GPU's memory and attributes
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(h.fam, s);
  if (h.kind == RTP_MULTI_CONV_TILED) return rtp_conv_tiled_multi_launch(h.launcher, s);
  if (h.kind == RTP_MULTI_WGRAD_TILED) return rtp_wgrad_tiled_multi_launch(h.launcher, s);
  return RTP_ERR_SHAPE;
}
