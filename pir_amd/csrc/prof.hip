// Optional per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the roofline
// object: average launch duration and units processed per kernel family).  Disabled by default: zero overhead.
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>
#include "pir_common.h"

namespace {
struct Rec { int id; hipEvent_t a, b; double units; };
std::mutex g_mu;
std::vector<Rec> g_recs;
bool g_enabled = false;
}  // namespace

extern "C" {

int pir_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (on && !g_enabled) {       // start from zero: discard what an earlier profiled region left on the device
    double drop[3];
    (void)pir_trace_profile_read(drop);
  }
  g_enabled = on != 0;
  return PIR_OK;
}

int pir_prof_enabled(void) { return g_enabled ? 1 : 0; }

// internal: returns a token (>= 0) or -1 when profiling is off
int pir_prof_begin(int id, double units, void* stream) {
  if (!g_enabled) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r{id, nullptr, nullptr, units};
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
  (void)hipEventRecord(r.a, (hipStream_t)stream);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}
void pir_prof_end(int token, void* stream) {
  if (token < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (token < (int)g_recs.size()) (void)hipEventRecord(g_recs[token].b, (hipStream_t)stream);
}

// Synchronises; fills launches[i], total_ms[i], units[i] for kernel ids 0..n_ids-1 and clears the records.
int pir_profile_read(int n_ids, long* launches, double* total_ms, double* units) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < n_ids; ++i) { launches[i] = 0; total_ms[i] = 0.0; units[i] = 0.0; }
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess &&
        r.id >= 0 && r.id < n_ids) {
      launches[r.id] += 1; total_ms[r.id] += ms; units[r.id] += r.units;
    }
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  g_recs.clear();
  // the tracer kernels count the SDF evaluations they execute on the device (their list sizes are not known to the
  // host): units of PIR_K_TRACE_STEP / PIR_K_SAMPLER / PIR_K_BISECT = evaluations (current device)
  double ev[3] = {0, 0, 0};
  if (pir_trace_profile_read(ev) == PIR_OK) {
    if (PIR_K_TRACE_STEP < n_ids) units[PIR_K_TRACE_STEP] = ev[0];
    if (PIR_K_SAMPLER < n_ids) units[PIR_K_SAMPLER] = ev[1];
    if (PIR_K_BISECT < n_ids) units[PIR_K_BISECT] = ev[2];
  }
  return PIR_OK;
}

}  // extern "C"
