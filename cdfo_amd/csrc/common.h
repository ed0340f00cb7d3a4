// Shared device/host helpers for the gfx950 kernels of libcdfo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cdfo_hip.h"
#include "prof.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

#define CDFO_LAUNCH_CHECK()                         \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

// Per-device host-side caches.  One process normally drives one GPU, but a caller may hold models on several devices:
// everything cached on the host (CU count, "this kernel's LDS attribute is set") is keyed by the CURRENT device ordinal
// (the Python wrappers make the operands' device current before every call).
constexpr int CDFO_MAXDEV = 64;
static inline int cdfo_cur_device() {
  int d = 0;
  return hipGetDevice(&d) == hipSuccess && d >= 0 && d < CDFO_MAXDEV ? d : -1;
}
// compute units the persistent kernels launched by the calling thread may fill: the current device's count (0 on failure), or
// the smaller value set by cdfo_set_cu_limit() (abi.hip) -- a caller that runs two schedules beside each other on two streams
// gives each a share of the chip (a persistent workgroup takes a CU's whole LDS, so shares do not overlap)
int cdfo_cu_limit_value();
static inline int cdfo_num_cus() {
  static int cus[CDFO_MAXDEV] = {0};
  const int d = cdfo_cur_device();
  if (d < 0) return 0;
  if (!cus[d]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) != hipSuccess) return 0;
    cus[d] = prop.multiProcessorCount;
  }
  const int lim = cdfo_cu_limit_value();
  return (lim > 0 && lim < cus[d]) ? lim : cus[d];
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (call site, device); `done` is the call site's static table
struct CdfoAttrOnce { bool done[CDFO_MAXDEV] = {false}; };
static inline hipError_t cdfo_set_max_lds(CdfoAttrOnce& once, const void* func, int bytes) {
  const int d = cdfo_cur_device();
  if (d < 0) return hipErrorInvalidDevice;
  if (once.done[d]) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) once.done[d] = true;
  return e;
}

// the same for kernels whose dynamic LDS size depends on the shapes: the attribute only ever grows (a larger value than
// needed can cost residency: the runtime may budget the declared maximum)
struct CdfoAttrGrow { int cur[CDFO_MAXDEV] = {0}; };
static inline hipError_t cdfo_grow_max_lds(CdfoAttrGrow& g, const void* func, int bytes) {
  const int d = cdfo_cur_device();
  if (d < 0) return hipErrorInvalidDevice;
  if (bytes <= g.cur[d]) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) g.cur[d] = bytes;
  return e;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case CDFO_ACT_LRELU: return v > 0.f ? v : 0.1f * v;
    case CDFO_ACT_RELU: return v > 0.f ? v : 0.f;
    case CDFO_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
