// Lightweight in-library kernel timer: when enabled, every MFMA kernel launch is bracketed by two HIP events
// recorded on the launch stream, tagged with a kernel class and its ALGORITHMIC work (FLOPs); bench.py reads
// the per-class totals after the timed region (the `roofline` object of its JSON line).
#include <mutex>
#include <vector>
#include "ph_common.h"
#include "ph_kernels.h"
#include "ph_dense.h"

namespace {
struct Rec { hipEvent_t a, b; int cls; double work, bytes; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;
bool g_on = false;
std::mutex g_mu;
constexpr size_t MAX_EVENTS = 1 << 15;

hipEvent_t take_event() {
  if (g_pool_next < g_pool.size()) return g_pool[g_pool_next++];
  if (g_pool.size() >= MAX_EVENTS) return nullptr;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  g_pool.push_back(e);
  ++g_pool_next;
  return e;
}
}  // namespace

bool ph_prof_on() { return g_on; }

void ph_prof_begin(int cls, double work, hipStream_t st, void** token) { ph_prof_begin2(cls, work, 0.0, st, token); }

void ph_prof_begin2(int cls, double work, double bytes, hipStream_t st, void** token) {
  *token = nullptr;
  if (!g_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t a = take_event(), b = take_event();
  if (!a || !b) return;
  g_recs.push_back(Rec{a, b, cls, work, bytes});
  (void)hipEventRecord(a, st);
  *token = reinterpret_cast<void*>(g_recs.size());   // index + 1
}

void ph_prof_end(void* token, hipStream_t st) {
  if (!token) return;
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_recs[reinterpret_cast<size_t>(token) - 1].b, st);
}

extern "C" {

int ph_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = on != 0;
  return PH_OK;
}

int ph_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.clear();
  g_pool_next = 0;
  return PH_OK;
}

// out[cls*4 + {0,1,2,3}] = {launches, total milliseconds, total algorithmic work, total algorithmic bytes}
int ph_prof_summary4(double* out, int nclasses) {
  if (hipDeviceSynchronize() != hipSuccess) return PH_ELAUNCH;
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < nclasses * 4; ++i) out[i] = 0.0;
  for (const Rec& r : g_recs) {
    if (r.cls < 0 || r.cls >= nclasses) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    out[r.cls * 4 + 0] += 1.0;
    out[r.cls * 4 + 1] += ms;
    out[r.cls * 4 + 2] += r.work;
    out[r.cls * 4 + 3] += r.bytes;
  }
  return PH_OK;
}

// out[cls*3 + {0,1,2}] = {launches, total milliseconds, total algorithmic work}; synchronises the device.
int ph_prof_summary(double* out, int nclasses) {
  if (hipDeviceSynchronize() != hipSuccess) return PH_ELAUNCH;
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < nclasses * 3; ++i) out[i] = 0.0;
  for (const Rec& r : g_recs) {
    if (r.cls < 0 || r.cls >= nclasses) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    out[r.cls * 3 + 0] += 1.0;
    out[r.cls * 3 + 1] += ms;
    out[r.cls * 3 + 2] += r.work;
  }
  return PH_OK;
}

// a one-thread kernel that stores the 100 MHz wall clock: a launch like any other, so it can sit inside a captured graph
// and mark when the stream reached that point of a replayed step (tests/bench_phases_gpu.py)
__global__ void stamp_kernel(unsigned long long* out) { *out = wall_clock64(); }

int ph_prof_stamp(unsigned long long* out, hipStream_t st) {
  if (!out) return PH_EINVAL;
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, st, out);
  PH_LAUNCH_CHECK();
  return PH_OK;
}

}  // extern "C"

int ph_num_cus() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cached[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}
