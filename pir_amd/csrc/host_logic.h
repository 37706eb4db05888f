// Host-side arithmetic of the C-ABI entry points, free of any HIP dependency so that it also builds with a plain
// C++ compiler: tests/host/test_host_logic.cpp compiles exactly these functions with -fsanitize=address,undefined
// (sanitizers run on the host build only) and drives them over edge-case sizes.
//   - tracer workspace layout (trace.hip: ws_layout) and argument checks
//   - K-slice assignment of the weight-gradient GEMMs (wgrad.hip: wgrad256_run, wgrad_run)
//   - chunk capacity of the SDF backward for a given scratch size (wgrad.hip: pir_sdf_get_all_backward)
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/pir_hip.h"

namespace pir_host {

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// ---- tracer workspace -------------------------------------------------------------------------------------------
constexpr int kTraceCounters = 64;
enum TraceArray { kTrCounters = 0, kTrList0, kTrList1, kTrSList, kTrBList, kTrDLo, kTrDHi, kTrFLo, kTrFHi, kTrBWork,
                  kTrMinIdx, kTrArrays };

// Byte offsets of the workspace arrays for n rays in `nchunks` reference chunks; returns the total size.
inline size_t trace_ws_offsets(long n, int nchunks, int max_bisect, size_t (&off)[kTrArrays]) {
  size_t cur = 0;
  auto take = [&](int which, size_t bytes) { off[which] = cur; cur += align256(bytes); };
  take(kTrCounters, sizeof(int) * kTraceCounters + sizeof(int) * (size_t)(max_bisect + 2) * (size_t)nchunks);
  take(kTrList0, sizeof(int) * (size_t)n);
  take(kTrList1, sizeof(int) * (size_t)n);
  take(kTrSList, sizeof(int) * (size_t)n);
  take(kTrBList, sizeof(int) * (size_t)n);
  take(kTrDLo, sizeof(float) * (size_t)n);
  take(kTrDHi, sizeof(float) * (size_t)n);
  take(kTrFLo, sizeof(float) * (size_t)n);
  take(kTrFHi, sizeof(float) * (size_t)n);
  take(kTrBWork, (size_t)n);
  take(kTrMinIdx, sizeof(int) * (size_t)n);
  return cur;
}

inline int trace_chunks(long n_rays, long chunk) {
  if (chunk <= 0) chunk = n_rays;
  return (int)((n_rays + chunk - 1) / chunk);
}

inline int trace_check_args(const void* packed, long n_rays, const void* ray_o, const void* ray_d, const void* min_dis,
                            const void* max_dis, const void* work_mask, int sphere_tracing_iters, int n_steps,
                            int max_bisect, const void* convergent, const void* points, const void* sdf,
                            const void* distance) {
  if (n_rays < 0 || n_rays > 0x7fffffffL / 4) return PIR_ERR_ARG;
  if (n_rays == 0) return PIR_OK;
  if (!packed || !ray_o || !ray_d || !min_dis || !max_dis || !work_mask || !convergent || !points || !sdf || !distance)
    return PIR_ERR_ARG;
  if (sphere_tracing_iters < 0 || sphere_tracing_iters > 30 || n_steps < 2 || max_bisect < 1 || max_bisect > 64)
    return PIR_ERR_ARG;
  return PIR_OK;
}

// ---- weight-gradient GEMM: K slices ---------------------------------------------------------------------------------
// 256 x 256 tiles: job j gets slices first[j] .. first[j+1]-1 in proportion to its cost per 32-point step, as many as
// fill one round of `cus` workgroups (capped by the slab space), at least 8 K steps per slice.  Returns the number of
// workgroups; first[] has n_jobs + 1 meaningful entries, the rest repeat the total.
inline int wgrad256_split(int n_jobs, const int* cost, int steps, long cus, long slab_cap, int (&first)[12]) {
  int total = 0;
  for (int j = 0; j < n_jobs; ++j) total += cost[j];
  long slots = cus < slab_cap ? cus : slab_cap;
  int n_wg = 0;
  for (int j = 0; j < n_jobs; ++j) {
    int k = total > 0 ? (int)(slots * cost[j] / total) : 1;
    const int kmax = steps / 8 > 0 ? steps / 8 : 1;
    if (k > kmax) k = kmax;
    if (k < 1) k = 1;
    first[j] = n_wg;
    n_wg += k;
  }
  for (int j = n_jobs; j < 12; ++j) first[j] = n_wg;
  return n_wg;
}

// 128 x 128 tiles: one split for all jobs, chosen so that jobs * ksplit fills whole rounds of `slots` workgroups;
// slices of at least 8 steps, at most ksplit_max.
inline int wgrad128_ksplit(int n_jobs, int n_rows, int k_pts, long slots, int ksplit_max) {
  int ksplit = 1;
  const int kmax = n_rows / k_pts / 8 < ksplit_max ? n_rows / k_pts / 8 : ksplit_max;
  double best = -1.0;
  for (int k = 1; k <= kmax; ++k) {
    const long wgs = (long)n_jobs * k, rounds = (wgs + slots - 1) / slots;
    const double eff = (double)wgs / (double)(rounds * slots);
    if (eff >= best - 1e-9) { best = eff; ksplit = k; }   // ties: the finer split
  }
  return ksplit;
}

// ---- SDF backward: points per pass for a scratch of scratch_bytes (0: too small) ------------------------------------
inline long bwd_chunk_cap(size_t scratch_bytes, size_t fixed_bytes, size_t row_bytes, long cap_max) {
  if (scratch_bytes < fixed_bytes + 32 * row_bytes + 256) return 0;
  long cap = (long)((scratch_bytes - fixed_bytes - 256) / row_bytes) / 32 * 32;
  return cap > cap_max ? cap_max : cap;
}

}  // namespace pir_host
