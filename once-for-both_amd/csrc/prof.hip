// Optional per-launch HIP-event timing, on the SAME stream the kernel is launched on.
// bench.py enables it around the timed region and reads per-tag {launches, ms, work} afterwards.
#include "ofb_common.h"
#include <vector>

namespace {
struct Rec { hipEvent_t a, b; int tag; double work; };
unsigned g_on = 0;       // bit t: bracket the launches of tag t
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  // device-scope release only: a default event's system-scope fence between kernels costs ~15 us per bracket
  if (hipEventCreateWithFlags(&e, hipEventReleaseToDevice) != hipSuccess) (void)hipEventCreate(&e);
  return e;
}
}  // namespace

void ofb_prof_pre(int tag, hipStream_t s, double work) {
  if (!((g_on >> tag) & 1u)) return;
  Rec r; r.a = get_event(); r.b = get_event(); r.tag = tag; r.work = work;
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
}
void ofb_prof_post(int tag, hipStream_t s) {
  if (!((g_on >> tag) & 1u)) return;
  (void)hipEventRecord(g_recs.back().b, s);
}

// on: bit t set -> launches of tag t are bracketed (0: off; each bracket costs an inter-kernel bubble, so the bench samples
// only the tags it reports)
extern "C" int ofb_prof_enable(int32_t on) {
  g_on = (unsigned)on;
  return OFB_OK;
}

// Synchronises the recorded events, accumulates per-tag totals into out[tag*3 + {0:launches,1:ms,2:work}]
// for tag < ntags, and clears the records.
extern "C" int ofb_prof_collect(double* out, int32_t ntags) {
  if (!out || ntags <= 0) return OFB_EINVAL;
  for (int i = 0; i < ntags * 3; ++i) out[i] = 0.0;
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    if (r.tag < ntags) { out[r.tag * 3] += 1.0; out[r.tag * 3 + 1] += ms; out[r.tag * 3 + 2] += r.work; }
    g_pool.push_back(r.a); g_pool.push_back(r.b);
  }
  g_recs.clear();
  return OFB_OK;
}

// Diagnostic: back-to-back v_mfma_f32_32x32x2_f32 on every SIMD (4 independent accumulators per wave); used by
// scripts/ to read the f32-MFMA rate this device sustains (the practical roof next to the 157.3 TF datasheet peak).
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = (float)threadIdx.x * 1e-3f, b = (float)blockIdx.x * 1e-3f + 1.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

extern "C" int ofb_diag_mfma_peak(float* out, int32_t blocks, int32_t iters, void* stream) {
  if (!out || blocks <= 0 || iters <= 0) return OFB_EINVAL;
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  return (int)hipGetLastError();
}

// Diagnostic: `blocks` workgroups of 256 threads that do nothing but hold their place for `usec` microseconds (s_memrealtime: 100 MHz;
// they sleep between looks at the clock) - a stand-in for the kernels of ANOTHER stream that sit on some CUs while the step runs (an
// RCCL all-reduce holds a workgroup per channel for the whole exchange).  lds_bytes > 0: the workgroup also holds that much LDS; the
// GEMM's two workgroups per CU use all 160 KB, so ANY LDS here keeps one of them off the CU (scripts/cu_thief.py).  Bounded: every
// wave leaves after `usec` (at most 1 s).
__global__ __launch_bounds__(256) void cu_thief_kernel(unsigned long long ticks, int* sink) {
  extern __shared__ int thief_lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && sink == reinterpret_cast<int*>(1)) thief_lds[0] = 1;     // (keeps the allocation referenced)
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

extern "C" int ofb_diag_cu_thief(int32_t blocks, int32_t usec, int32_t lds_bytes, void* stream) {
  if (blocks <= 0 || blocks > 1024 || usec <= 0 || usec > 1000000 || lds_bytes < 0 || lds_bytes > 163840) return OFB_EINVAL;
  hipLaunchKernelGGL(cu_thief_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (unsigned long long)usec * 100ull,
                     (int*)nullptr);
  return (int)hipGetLastError();
}
