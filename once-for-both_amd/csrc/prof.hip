// Optional per-launch HIP-event timing, on the SAME stream the kernel is launched on.
// bench.py enables it around the timed region and reads per-tag {launches, ms, work} afterwards.
#include "ofb_common.h"
#include <vector>

namespace {
struct Rec { hipEvent_t a, b; int tag; double work; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e; hipEventCreate(&e); return e;
}
}  // namespace

void ofb_prof_pre(int tag, hipStream_t s, double work) {
  if (!g_on) return;
  Rec r; r.a = get_event(); r.b = get_event(); r.tag = tag; r.work = work;
  hipEventRecord(r.a, s);
  g_recs.push_back(r);
}
void ofb_prof_post(int tag, hipStream_t s) {
  if (!g_on) return;
  (void)tag;
  hipEventRecord(g_recs.back().b, s);
}

extern "C" int ofb_prof_enable(int32_t on) {
  g_on = on != 0;
  return OFB_OK;
}

// Synchronises the recorded events, accumulates per-tag totals into out[tag*3 + {0:launches,1:ms,2:work}]
// for tag < ntags, and clears the records.
extern "C" int ofb_prof_collect(double* out, int32_t ntags) {
  if (!out || ntags <= 0) return OFB_EINVAL;
  for (int i = 0; i < ntags * 3; ++i) out[i] = 0.0;
  for (auto& r : g_recs) {
    hipEventSynchronize(r.b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.a, r.b);
    if (r.tag < ntags) { out[r.tag * 3] += 1.0; out[r.tag * 3 + 1] += ms; out[r.tag * 3 + 2] += r.work; }
    g_pool.push_back(r.a); g_pool.push_back(r.b);
  }
  g_recs.clear();
  return OFB_OK;
}
