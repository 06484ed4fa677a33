// Slot pool of the dynamic work-claiming counters (rtp_claim.h).
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "rtp_claim.h"
#include "rtp_common.h"

#define RTP_CLAIM_POOL_INTS (1 << 20)   // 4 MB of zero-initialised device memory per GPU (the code object's .bss): 65 536 slots of 16 ints

__device__ int g_claim_pool[RTP_CLAIM_POOL_INTS];

namespace {
struct DevPool {
  int* base = nullptr;
  int next = 0;
  std::unordered_map<const void*, std::pair<int, int>> slot_of;   // key -> (offset, ints)
};
std::mutex g_mu;
DevPool g_pool[16];
}  // namespace

// Opt-in (RTP_CLAIM=1).  Measured on MI355X (hr3d, B = 8, DESIGN.md section 8 "dynamic claiming"): the claimed kernels run 3-6 %
// slower than the static deal on an undisturbed GPU (more double-pair stagings, the unit bookkeeping) and the step gains at most
// 1 % even with the lower levels' launches kept narrow -- the lanes' kernels serialise as WHOLE kernels (every tiled launch, main
// lane or not, asks for all 256 CUs), they do not delay individual workgroups.  Claiming also makes the per-workgroup statistics
// partials depend on who took which brick (forward results differ in the last bit from run to run), so the default stays static.
int rtp_claim_enabled() {
  static const int on = (getenv("RTP_CLAIM") && atoi(getenv("RTP_CLAIM")) != 0) ? 1 : 0;
  return on;
}

int* rtp_claim_slot(const void* key, int nctr) {
  int dev = 0;
  if (nctr < 1 || nctr > RTP_CLAIM_MAX_CTRS) return nullptr;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  const int need = (nctr + 1 + 15) / 16 * 16;   // whole 64-byte lines: two slots never share one
  std::lock_guard<std::mutex> lk(g_mu);
  DevPool& P = g_pool[dev];
  if (!P.base) {
    void* sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_claim_pool)) != hipSuccess || !sym) return nullptr;
    P.base = (int*)sym;
  }
  auto it = P.slot_of.find(key);
  if (it != P.slot_of.end() && it->second.second >= need) return P.base + it->second.first;
  if (P.next + need > RTP_CLAIM_POOL_INTS) return nullptr;
  const int off = P.next;
  P.next += need;
  P.slot_of[key] = std::make_pair(off, need);
  return P.base + off;
}

// Test / tooling hook: ints of the pool handed out on the current device.
extern "C" int rtp_claim_slots_in_use(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  return g_pool[dev].next;
}

// ---- per-launch width hints -------------------------------------------------------------------------------------------------
// rtp_tiled_width_hint(key, total_wgs): the LDS-tiled launch whose OUTPUT buffer (conv / data gradient: y + channel offset; weight
// gradient: the slab buffer) starts at `key` runs on total_wgs workgroups instead of one per CU (0 removes the hint).  The launch
// plan uses it to keep a few CUs free for the other lanes' dependent chains exactly where the main lane would otherwise wait for them
// (rt_pose_amd/engine.py); per-workgroup partial buffers keep their size (the upper slots stay zero).
namespace {
std::mutex g_wmu;
std::unordered_map<const void*, int> g_width;
}
extern "C" int rtp_tiled_width_hint(const void* key, int total_wgs) {
  std::lock_guard<std::mutex> lk(g_wmu);
  if (total_wgs > 0) g_width[key] = total_wgs; else g_width.erase(key);
  return RTP_OK;
}
int rtp_tiled_width_for(const void* key) {
  std::lock_guard<std::mutex> lk(g_wmu);
  auto it = g_width.find(key);
  return it == g_width.end() ? 0 : it->second;
}
