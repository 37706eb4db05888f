// Shared host-side helpers for the C-ABI entry points.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/pir_hip.h"

static inline int pir_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PIR_OK : PIR_ERR_LAUNCH;
}

static inline int pir_num_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

// kernel families for the optional event profiler (prof.hip)
extern "C" int pir_prof_begin(int id, double units, void* stream);
extern "C" void pir_prof_end(int token, void* stream);
extern "C" int pir_prof_enabled(void);
extern "C" int pir_trace_profile_read(double* evals3);
struct PirProfScope {
  int tok; void* st;
  PirProfScope(int id, double units, void* stream) : tok(pir_prof_begin(id, units, stream)), st(stream) {}
  ~PirProfScope() { pir_prof_end(tok, st); }
};
