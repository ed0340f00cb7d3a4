// ABI bookkeeping for libcdfo_hip.so.
#include "common.h"

extern "C" int cdfo_abi_version(void) { return 1; }
static thread_local int g_cu_limit = 0;
int cdfo_cu_limit_value() { return g_cu_limit; }
// Limit the CUs that persistent kernels launched by THIS host thread fill (0 = all); returns the previous value.
extern "C" int cdfo_set_cu_limit(int n) {
  const int prev = g_cu_limit;
  g_cu_limit = n > 0 ? n : 0;
  return prev;
}
extern "C" int cdfo_sizeof_conv_args(void) { return (int)sizeof(cdfo_conv_args); }
extern "C" const char* cdfo_build_info(void) { return "libcdfo_hip gfx950 (CDNA4) " __DATE__ " " __TIME__; }

#include "prof.h"
#include <stdlib.h>

CdfoProfState& cdfo_prof_state() {
  static CdfoProfState s;
  return s;
}

// Start recording one (start, stop) event pair around every kernel launch of this library (up to max_records).
extern "C" int cdfo_prof_begin(int max_records) {
  CdfoProfState& p = cdfo_prof_state();
  if (max_records <= 0) return CDFO_EINVAL;
  if (p.cap < max_records) {
    for (int i = 0; i < 2 * p.cap; ++i) (void)hipEventDestroy(p.ev[i]);
    free(p.ev); free(p.kid); free(p.flops); free(p.bytes);
    p.ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * 2 * max_records);
    p.kid = (int*)malloc(sizeof(int) * max_records);
    p.flops = (double*)malloc(sizeof(double) * max_records);
    p.bytes = (double*)malloc(sizeof(double) * max_records);
    for (int i = 0; i < 2 * max_records; ++i) {
      hipError_t e = hipEventCreate(&p.ev[i]);
      if (e != hipSuccess) return (int)e;
    }
    p.cap = max_records;
  }
  p.n = 0;
  p.enabled = true;
  return 0;
}

// Stop recording, wait for the events, and accumulate per kernel family (arrays of KID_COUNT entries):
// launches, total milliseconds, total algorithmic FLOPs, total algorithmic bytes.  Returns #records or <0.
extern "C" int cdfo_prof_end(int* launches, double* ms, double* flops, double* bytes, int nkid) {
  CdfoProfState& p = cdfo_prof_state();
  p.enabled = false;
  if (nkid < KID_COUNT) return CDFO_EINVAL;
  for (int k = 0; k < nkid; ++k) { launches[k] = 0; ms[k] = 0; flops[k] = 0; bytes[k] = 0; }
  for (int i = 0; i < p.n; ++i) {
    if (hipEventSynchronize(p.ev[2 * i + 1]) != hipSuccess) return -3;
    float t = 0.f;
    if (hipEventElapsedTime(&t, p.ev[2 * i], p.ev[2 * i + 1]) != hipSuccess) return -3;
    const int k = p.kid[i];
    launches[k] += 1; ms[k] += t; flops[k] += p.flops[i]; bytes[k] += p.bytes[i];
  }
  const int n = p.n;
  p.n = 0;
  return n;
}

extern "C" int cdfo_prof_kid_count(void) { return KID_COUNT; }
