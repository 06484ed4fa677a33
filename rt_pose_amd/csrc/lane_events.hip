// Cross-stream ordering events of the lane plan (rt_pose_amd/lanes.py), created WITHOUT the system-scope fence.
//
// The plan orders its four streams with ~110 events per step.  An event made by hipEventCreateWithFlags(hipEventDisableTiming) --
// what torch.cuda.Event() is -- performs a SYSTEM-scope release when it is recorded: the producing kernel's dirty L2 lines are
// written back for the host and the L2 is invalidated for whatever runs next (hip_runtime_api.h, hipEventDisableSystemFence: "the
// cost of cache writeback and invalidation, and the performance impact of those actions on the execution of following work").
// Every consumer of these events is a kernel on another stream of the SAME device, for which the dispatch packets' own agent-scope
// acquire / release is the required ordering; nothing on the host or on another GPU reads memory on the strength of them (the host
// synchronises with torch.cuda.synchronize / its own events, the gradient all-reduce is stream-ordered behind the backward pass).
#include <hip/hip_runtime.h>

#include "../../include/rtp.h"
#include "rtp_common.h"

extern "C" int rtp_event_create(void** ev_out, int system_fence) {
  if (!ev_out) return RTP_ERR_SHAPE;
  hipEvent_t e = nullptr;
  const unsigned flags = hipEventDisableTiming | (system_fence ? 0u : hipEventDisableSystemFence);
  if (hipEventCreateWithFlags(&e, flags) != hipSuccess) return RTP_ERR_LAUNCH;
  *ev_out = (void*)e;
  return RTP_OK;
}

extern "C" int rtp_event_destroy(void* ev) {
  if (!ev) return RTP_ERR_SHAPE;
  return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? RTP_OK : RTP_ERR_LAUNCH;
}

extern "C" int rtp_event_record(void* ev, void* stream) {
  if (!ev) return RTP_ERR_SHAPE;
  return hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) == hipSuccess ? RTP_OK : RTP_ERR_LAUNCH;
}

extern "C" int rtp_stream_wait_event(void* stream, void* ev) {
  if (!ev) return RTP_ERR_SHAPE;
  return hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0) == hipSuccess ? RTP_OK : RTP_ERR_LAUNCH;
}
