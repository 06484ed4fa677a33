#include "rtp_prof.h"

#include <vector>

#include "../../include/rtp.h"

int g_rtp_prof_on[RTP_FAM_COUNT] = {0};
namespace {
struct Pool {
  std::vector<hipEvent_t> start, stop;
  size_t used = 0;
};
Pool g_pool[RTP_FAM_COUNT];
}  // namespace

void rtp_prof_begin(int fam, hipStream_t s) {
  Pool& p = g_pool[fam];
  if (p.used == p.start.size()) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    p.start.push_back(a);
    p.stop.push_back(b);
  }
  (void)hipEventRecord(p.start[p.used], s);
}

void rtp_prof_end(int fam, hipStream_t s) {
  Pool& p = g_pool[fam];
  (void)hipEventRecord(p.stop[p.used], s);
  p.used++;
}

extern "C" int rtp_prof_enable(int family, int on) {
  if (family < 0 || family >= RTP_FAM_COUNT) return RTP_ERR_SHAPE;
  g_rtp_prof_on[family] = on;
  if (on) g_pool[family].used = 0;
  return RTP_OK;
}

extern "C" int rtp_prof_collect(int family, float* total_ms, int* launches) {
  if (family < 0 || family >= RTP_FAM_COUNT) return RTP_ERR_SHAPE;
  Pool& p = g_pool[family];
  float tot = 0.f;
  for (size_t i = 0; i < p.used; ++i) {
    (void)hipEventSynchronize(p.stop[i]);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, p.start[i], p.stop[i]);
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (int)p.used;
  p.used = 0;
  return RTP_OK;
}

extern "C" const char* rtp_version(void) { return "rt_pose_amd 0.1 (gfx950)"; }
