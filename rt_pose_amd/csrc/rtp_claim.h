// Dynamic work claiming for the persistent LDS-tiled kernels (conv_tiled / wgrad_tiled / the stride-2 tiled kernels).
//
// Why: a persistent kernel that DEALS its bricks statically finishes when its slowest workgroup does.  In the lane plan
// (rt_pose_amd/lanes.py) side-lane kernels hold some CUs for 20-50 us at a time, so a few of the main lane's workgroups start
// late and the whole launch stretches by that much (profiles/r03_main_lane_trace.txt: 1.3 ms of a 5.9 ms step).  With the
// bricks CLAIMED from a counter, a workgroup that starts late simply takes fewer of them.
//
// Layout: every launch owns a SLOT of 32-bit counters in a zero-initialised device pool -- one counter per (sample, range)
// (the slot's last word is spare).  A slot is keyed by the launch's output pointer: two launches that write the same buffer are
// never in flight together (write-after-write dependency), so a slot never serves two kernels at once.  The claimer that makes a
// counter's last take of a launch resets it (below), which the next launch on the slot observes (kernel boundary): no memset sits
// between launches.
// All accesses are agent-scope atomics (the L2s of the 8 XCDs are not coherent with each other).
#pragma once
#include <hip/hip_runtime.h>

#define RTP_CLAIM_MAX_CTRS 1024         // counters per slot at most; word [nctr] of a slot is its done counter

// host: the slot (device pointer, nctr + 1 ints, all zero between launches) for this key on the current device; nullptr when
// the pool is exhausted or nctr is out of range (callers fall back to the static deal)
int* rtp_claim_slot(const void* key, int nctr);
// 1: RTP_CLAIM=1 (opt-in; the default is the static deal)
int rtp_claim_enabled();

// device: take `unit` consecutive items from counter `ctr`; returns the first item's index (>= limit: nothing left)
__device__ __forceinline__ int rtp_claim_take(int* ctr, int unit) {
  return __hip_atomic_fetch_add(ctr, unit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Resetting without a second counter: every claimer keeps taking from a counter until its take FAILS (index beyond the last unit)
// and then never touches that counter again, so a launch makes exactly units + claimers takes on it -- the claimer that draws the
// index units + claimers - 1 knows it is the last one and stores 0 (nobody waits for anything at the end of the kernel).
__device__ __forceinline__ void rtp_claim_reset_if_last(int* ctr, int drawn, int units, int claimers) {
  if (drawn == units + claimers - 1) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// per-launch width hint for the output buffer `key` (claim.hip); 0 = none
int rtp_tiled_width_for(const void* key);
