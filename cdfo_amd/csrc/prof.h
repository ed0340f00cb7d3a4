// Optional per-launch timing with HIP events on the launch stream (used by bench.py for the live roofline
// numbers).  Disabled by default: zero events are recorded unless cdfo_prof_begin() was called.
#pragma once
#include <hip/hip_runtime.h>

enum CdfoKid {
  KID_CONV3_WIDE = 0, KID_CONV3_NARROW, KID_CONV1, KID_CONV3_S2, KID_STEM, KID_LAYERNORM, KID_DWCONV, KID_FLOW_WARP,
  KID_RESAMPLE, KID_SCALE, KID_CONV_LAST, KID_SMALL_CONV, KID_SPATIAL_GATE, KID_CHAN_SUM, KID_GRAM, KID_FOLD,
  KID_RDAB_PREP, KID_COLCONV9, KID_ATTN_ROW, KID_ATTN_COL, KID_ATTN_WIN, KID_LAYOUT, KID_PACK, KID_DCN, KID_CONV3_WS,
  KID_CONV3_RING, KID_CONV3_RING4, KID_CONV3_WS_RES, KID_DCN_BWD, KID_CONV3_WINO, KID_COUNT
};

struct CdfoProfState {
  bool enabled = false;
  int cap = 0, n = 0;
  hipEvent_t* ev = nullptr;   // 2 per record
  int* kid = nullptr;
  double* flops = nullptr;
  double* bytes = nullptr;
};
CdfoProfState& cdfo_prof_state();

struct CdfoProfScope {
  hipStream_t st;
  int slot = -1;
  CdfoProfScope(hipStream_t s, int kid, double flops, double bytes) : st(s) {
    CdfoProfState& p = cdfo_prof_state();
    if (!p.enabled || p.n >= p.cap) return;
    slot = p.n++;
    p.kid[slot] = kid; p.flops[slot] = flops; p.bytes[slot] = bytes;
    (void)hipEventRecord(p.ev[2 * slot], st);
  }
  ~CdfoProfScope() {
    if (slot >= 0) (void)hipEventRecord(cdfo_prof_state().ev[2 * slot + 1], st);
  }
};
