// 3x3 stride-1 convolution with 64 input channels as a ROW-STREAMING Winograd F(2,3) product along x (fp16 MFMA, fp32 accumulate).
//
// Block_.body[0] (arch/SIDECVSR_our.py:383-387: Conv2d(64, 256, 3, 1, 1) + LeakyReLU) is a third of the forward's time and its direct
// form (conv3x3_ws.hip) is matrix-pipe / power bound: 1.05-1.18 PFLOP/s whatever the schedule.  The lever left is the number of
// MFMAs.  The one-dimensional minimal filtering algorithm F(2,3) along x computes two adjacent outputs of a row from four
// transformed inputs instead of six products:
//     d = x[2t-1 .. 2t+2],   V0 = d0 - d2,  V1 = d1 + d2,  V2 = d2 - d1,  V3 = d1 - d3                       (per input row, channel)
//     U0 = g0,  U1 = (g0 + g1 + g2) / 2,  U2 = (g0 - g1 + g2) / 2,  U3 = g2                                 (per kernel row dy)
//     M_xi = sum over dy, channel of U_xi[dy] * V_xi[row + dy - 1]
//     y[2t] = M0 + M1 + M2,   y[2t+1] = M1 - M2 - M3
// i.e. four independent "3x1 vertical convolutions" with K = 3 x 64 on a half-width grid: 2/3 of the direct form's MACs.  The
// two-dimensional F(2x2,3x3) would cut them to 4/9 but its 24-add output transform per output channel does not hide beside the
// MFMAs of a 64-deep contraction, and the kernel would sit on the HBM roof anyway (the fp16 result is 4x the input).
// Numerics: V and U are rounded to fp16 once (v_pk_add_f16 rounds the exact difference; U is formed in fp32 on the host); measured
// against an exact convolution the result's error is ~1.3x that of the direct product on fp16-rounded operands.  |V| <= 2 |x|: the
// model's fp16 range guard budgets for the factor two.
//
// Structure (no LDS-DMA, no weights in LDS):
//   * persistent 512-thread workgroup per CU = 128 output channels x a sequence of UNITS (image, 32-pixel column strip, row
//     segment); the Cout / 128 workgroups of a unit sit on one XCD and run the same unit sequence (shared L2);
//   * wave w owns 16 output channels whose 24 weight fragments (3 dy x 4 xi x 2 K halves, `cdfo_pack_conv3x3_wino`) stay in its
//     REGISTERS for the whole launch (96 VGPRs): the matrix cores' A operand never touches LDS;
//   * the unit is walked down one input row at a time: row i adds U[0] V[i] to output row i + 1, U[1] V[i] to row i and U[2] V[i]
//     to row i - 1, which is then complete (three rolling accumulator sets of 4 xi x 16 channels x 16 column pairs): 24
//     v_mfma_f32_16x16x32_f16 per row and wave, no vertical halo recompute inside a segment;
//   * V is computed ONCE per workgroup: per batch of three input rows wave w = (xi, K half) loads its two raw fragments per row
//     straight from global memory into registers (lane = column pair x 8-channel group = the MFMA B-fragment lane), forms V with
//     four v_pk_add_f16 and writes one 1 KiB fragment per row into a two-slot LDS ring; all eight waves read the eight fragments of
//     a row back in lane order (conflict-free).  One workgroup barrier per batch; the loads of batch n + 2 are in flight while
//     batch n computes;
//   * the epilogue of a completed row is lane-local: 4 adds per output channel (the output transform), activation, fp16, two
//     8-byte stores (columns 2t, 2t + 1) into the chunk-planar result [B][Cout/16][H][W][16] or its space-to-depth form.
#include "common.h"
#include <type_traits>

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int WN_THREADS = 512;
constexpr int WN_ROWS = 3;                       // input rows per batch (= the period of the rolling accumulator sets)
constexpr int WN_VROW = 8 * 1024;                // V of one input row: fragment xi * 2 + K half, 1 KiB each
constexpr int WN_SLOT = WN_ROWS * WN_VROW;
#ifndef WN_BSYNC
#define WN_BSYNC 1      // the row of a batch behind which waves 4-7 pass its barrier (their V fragments are written behind row 0)
#endif
constexpr int WN_NSLOT = 3;                      // V ring: batch n lives in slot n % 3 (see the barrier's place in `batch`)
constexpr int WN_LDS = WN_NSLOT * WN_SLOT;       // 72 KiB
// UP (on-the-fly bilinear x2 of a low-resolution source): the three low-resolution rows a batch interpolates from, staged for the
// workgroup as 432 16-byte units [row 3][plane 4][pixel 18][half 2] of a 512-unit slot; two slots behind the V ring
constexpr int WN_RAW_SLOT = 8192, WN_RAW_OFF = WN_LDS, WN_LDS_UP = WN_LDS + 2 * WN_RAW_SLOT;

struct wino_args {
  const unsigned char* src; int B, H, W;
  const f16x8_t* w; const float* bias; int Cout, act;
  _Float16* out; int s2d, halfsplit;
  int seg_h, nseg, nstrips, nb;                  // rows per segment, segments per strip, strips per image row, batches per unit
  unsigned long long* clk;                       // dbg 512: s_memtime stamps [workgroup][wave][8] of one steady-state batch
};

typedef int i32x4 __attribute__((ext_vector_type(4)));

// The pixel loads are compiler-visible buffer loads: their data crosses basic blocks in REGISTERS (issued in one batch, consumed in
// the next), and hipcc may copy such registers at a control-flow merge -- with inline-assembly loads it did so while the data was
// still in flight (stale V fragments in the first batches of a unit, timing dependent).  The price of visible loads is hipcc's
// wait-count pass, which merges the states of a loop's pre-header and back edge pessimistically: the loops below are arranged so
// that every path into a batch has issued the SAME sequence [row-0 store, six loads, row-1 store, row-2 store] before it.
// DBG (developer ablations, wrong results): 1 = no MFMAs, 2 = no global loads, 4 = no stores, 8 = no epilogue arithmetic,
// 16 = no barrier / V exchange wait, 32 / 64 = loads / stores with lane-contiguous addresses (one cache line per four lanes)
// UP: a.src is the LOW-resolution tensor [B][4][H/2][W/2][16] and the convolution runs on its bilinear x2 (align_corners = False:
// out[2i] = 1/4 in[i-1] + 3/4 in[i], out[2i+1] = 3/4 in[i] + 1/4 in[i+1], clamped taps) WITHOUT that image ever existing.  The
// transformed inputs are linear in the low-resolution rows:  V(y) = 3/4 Vh(i) + 1/4 Vh(i -+ 1)  with, per column pair t (= low-
// resolution column t),  Vh_0 = 3/4 L[t-1] - 1/2 L[t] - 1/4 L[t+1],  Vh_1 = 1/4 L[t-1] + 3/2 L[t] + 1/4 L[t+1],
// Vh_2 = 1/4 (L[t+1] - L[t-1]),  Vh_3 = 1/4 L[t-1] + 1/2 L[t] - 3/4 L[t+1]  (clamped columns); two exceptions where the convolution's
// ZERO padding of the x2 image meets the interpolation's clamping: Vh_0 = -(3/4 L[0] + 1/4 L[1]) at t = 0 and Vh_3 = 1/4 L[t-1] +
// 3/4 L[t] at the last column.  Rows -1 and H of the x2 image are zero.  Data path: every lane fetches ONE 16-byte unit of the
// batch's three low-resolution rows (18 pixels x 4 planes: contiguous 576-byte runs, against 16 bytes of every second record in the
// plain form), the rows go through a two-slot LDS stage, wave (xi, K half) reads its three shifted fragments per row from there.
template <bool S2D, int DBG, bool UP = false>
__global__ __launch_bounds__(WN_THREADS) void conv3x3_c64_wino_kernel(wino_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, t16 = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, NB = a.nb;
  const int nhalf = a.Cout >> 7;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int npairs = nslots / nhalf;
  const int half = slot % nhalf, pair = slot / nhalf;
  if (pair >= npairs) return;
  const int units_img = a.nstrips * a.nseg, n_units = a.B * units_img;
  const int stride_u = npairs * 8, first_u = pair * 8 + xcd;
  const int my_units = first_u < n_units ? (n_units - first_u + stride_u - 1) / stride_u : 0;
  if (my_units == 0) return;
  const int T = my_units * NB;

  // ---- this wave's weights and bias: 16 output channels cb * 16 .. + 15; accumulator rows 4 kg .. 4 kg + 3 of column t16
  const int cb = half * 8 + wave;
  f16x8_t wf[3][4][2];
  {
    const f16x8_t* wp = a.w + (long long)cb * 24 * 64 + lane;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) wf[dy][xi][sc] = wp[((dy * 4 + xi) * 2 + sc) * 64];
    // a use right here: the compiler's wait-count bookkeeping then has no weight load pending at the loop entry (it would merge
    // such a pending load into the loop's back edge and wait for the PREFETCHED pixel loads inside the MFMA blocks)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) asm volatile("" : "+v"(wf[dy][xi][sc]));
  }
  f32x4 binit = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
#pragma unroll
    for (int k = 0; k < 4; ++k) binit[k] = a.bias[cb * 16 + 4 * kg + k];
  }
  asm volatile("" : "+v"(binit));      // waited for here, not inside the loop (see the weights)
  const f32x4 zinit = {0.f, 0.f, 0.f, 0.f};
  const _Float16 slope1 = a.act == CDFO_ACT_NONE ? (_Float16)1.f : (a.act == CDFO_ACT_LRELU ? (_Float16)0.1f : (_Float16)0.f);
  const f16x4_t slope_h = {slope1, slope1, slope1, slope1};

  // ---- V production role of this wave: fragment wave = xi_p * 2 + sc_p; V = sA dA + sB dB with
  //      xi 0: d0 - d2, xi 1: d1 + d2, xi 2: d2 - d1, xi 3: d1 - d3   (d_j = pixel x0 + 2t + j - 1)
  const int xi_p = wave >> 1, sc_p = wave & 1;
  const int jA = xi_p == 0 ? 0 : (xi_p == 2 ? 2 : 1), jB = xi_p == 0 ? 2 : (xi_p == 1 ? 2 : (xi_p == 2 ? 1 : 3));
  const _Float16 vsg = xi_p == 1 ? (_Float16)1.f : (_Float16)-1.f;
  const f16x8_t vsgn = {vsg, vsg, vsg, vsg, vsg, vsg, vsg, vsg};
  const unsigned img_bytes = UP ? (unsigned)((H >> 1) * (W >> 1)) * 128u : (unsigned)(H * W) * 128u;      // 4 planes x 32 bytes per pixel
  const unsigned lane_src = (unsigned)(sc_p * 2 + (kg >> 1)) * (unsigned)(H * W) * 32u + (unsigned)(kg & 1) * 16u;
  const int nck = a.Cout >> 4;
  const unsigned out_img_bytes = (unsigned)nck * (unsigned)(H * W) * 32u;
  const unsigned lane16 = (unsigned)lane * 16u;
  const bool grp_b = wave >= 4;        // the SIMD partners of waves 0-3 (see STAGGER below)
  if ((DBG & 128) && grp_b) __builtin_amdgcn_s_setprio(1);      // experiment: static priority for the second-dispatched half
  if ((DBG & 256) && !grp_b) __builtin_amdgcn_s_setprio(1);     // experiment: ... or for the first half

  auto unit_coords = [&](int ord, int& b, int& x0, int& y0, int& y1) {
    const int u = (ord * npairs + pair) * 8 + xcd;
    b = u / units_img;
    const int r = u - b * units_img;
    const int sy = r / a.nstrips;
    x0 = (r - sy * a.nstrips) * 32; y0 = sy * a.seg_h; y1 = min(y0 + a.seg_h, H);
  };
  // per-lane address parts (they depend on the unit's x0 only): the row part of every access is a scalar offset, a row outside
  // the image / segment gets a zero-sized buffer descriptor (hardware zero fill) -- no vector instruction per load or store
  auto src_lane_offsets = [&](int x0, unsigned& va, unsigned& vb) {
    const int xa = x0 + 2 * t16 + jA - 1, xb = x0 + 2 * t16 + jB - 1;
    va = (xa >= 0 && xa < W) ? lane_src + (unsigned)xa * 32u : 0x80000000u;
    vb = (xb >= 0 && xb < W) ? lane_src + (unsigned)xb * 32u : 0x80000000u;
    if (DBG & 32) { va = (unsigned)x0 * 32u + lane16; vb = va + 1024u; }      // ablation: line-contiguous loads (wrong data)
  };

  u32x4 raw[WN_ROWS][2];
  // (batches past the end re-load / re-write the last one: no branch in the steady-state loop)
  auto issue_loads = [&](int n, unsigned va, unsigned vb, int b, int y0, int y1, int i0) {
    const unsigned char* base = a.src + (long long)b * img_bytes;
#pragma unroll
    for (int j = 0; j < WN_ROWS; ++j) {
      const int y = i0 + j;
      const int rowok = -(int)(y >= 0 && y < H && y <= y1);      // all ones / zero: scalar masks, no branch
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, (int)img_bytes & rowok, 0x00020000);
      const int so = (y * W * 32) & rowok;
      if (DBG & 2) {
        raw[j][0] = u32x4{va, vb, (unsigned)so, 0u}; raw[j][1] = u32x4{vb, va, 0u, (unsigned)so};
      } else {
        raw[j][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)va, so, 0));
        raw[j][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vb, so, 0));
      }
    }
  };
  auto write_v = [&](int n) {
    unsigned uoff = (unsigned)((n % WN_NSLOT) * WN_SLOT + wave * 1024);
    asm volatile("" : "+s"(uoff));        // formed per batch from the lane part every LDS access shares (a hoisted copy gets spilled)
    unsigned char* dst = smem + lane16 + uoff;
#pragma unroll
    for (int j = 0; j < WN_ROWS; ++j) {
      const f16x8_t va = __builtin_bit_cast(f16x8_t, raw[j][0]), vb = __builtin_bit_cast(f16x8_t, raw[j][1]);
      *reinterpret_cast<f16x8_t*>(dst + j * WN_VROW) = __builtin_elementwise_fma(vb, vsgn, va);      // one rounding: exact sum, rounded
    }
  };

  // ---- UP: staging role of this lane (unit su of a raw slot), the staged load of a batch, and the V fragments formed from a slot
  const int Hl = H >> 1, Wl = W >> 1;
  const int su = wave * 64 + lane;
  const int s_px = (su >> 1) % 18, s_pl = ((su >> 1) / 18) & 3, s_r = (su >> 1) / 72;
  const unsigned s_const = (unsigned)s_pl * (unsigned)(Hl * Wl) * 32u + (unsigned)(su & 1) * 16u;
  u32x4 sreg = {0u, 0u, 0u, 0u};      // the staged unit in flight (requested one batch before it is written to LDS)
  auto stage_load = [&](int b, int x0, int i0) {
    const int base = (i0 - 1) >> 1;       // rows base .. base + 2 of the low-resolution image (clamped: the interpolation's edge rule)
    const int row = min(max(base + s_r, 0), Hl - 1), col = min(max((x0 >> 1) - 1 + s_px, 0), Wl - 1);
    const unsigned vo = su < 432 ? s_const + (unsigned)(row * Wl + col) * 32u : 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.src) + (long long)b * img_bytes, 0,
                                                                        (int)img_bytes, 0x00020000);
    if (DBG & 2) sreg = u32x4{vo, vo, vo, vo};
    else sreg = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, 0, 0));
  };
  auto stage_write = [&](int n) {
    unsigned roff = (unsigned)(WN_RAW_OFF + (n & 1) * WN_RAW_SLOT);
    asm volatile("" : "+s"(roff));
    *reinterpret_cast<u32x4*>(smem + roff + (unsigned)su * 16u) = sreg;
  };
  auto form_v = [&](int n, int x0, int i0) {
    // this lane's column pair = low-resolution column tg; coefficients of its xi (all exact in fp16), with the two edge exceptions
    const int tg = (x0 >> 1) + t16;
    float ca = xi_p == 0 ? 0.75f : (xi_p == 2 ? -0.25f : 0.25f);
    float cb_ = xi_p == 0 ? -0.5f : (xi_p == 1 ? 1.5f : (xi_p == 2 ? 0.f : 0.5f));
    float cc = xi_p == 0 ? -0.25f : (xi_p == 3 ? -0.75f : 0.25f);
    if (xi_p == 0 && tg == 0) { ca = 0.f; cb_ = -0.75f; cc = -0.25f; }
    if (xi_p == 3 && tg == Wl - 1) { ca = 0.25f; cb_ = 0.75f; cc = 0.f; }
    const _Float16 ha = (_Float16)ca, hb = (_Float16)cb_, hc = (_Float16)cc;
    const f16x8_t va = {ha, ha, ha, ha, ha, ha, ha, ha}, vb = {hb, hb, hb, hb, hb, hb, hb, hb}, vc = {hc, hc, hc, hc, hc, hc, hc, hc};
    unsigned roff = (unsigned)(WN_RAW_OFF + (n & 1) * WN_RAW_SLOT);
    asm volatile("" : "+s"(roff));
    const unsigned char* rb = smem + roff + (unsigned)(((sc_p * 2 + (kg >> 1)) * 18 + t16) * 32 + (kg & 1) * 16);
    f16x8_t vh[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const f16x8_t f0 = *reinterpret_cast<const f16x8_t*>(rb + r * 2304), f1 = *reinterpret_cast<const f16x8_t*>(rb + r * 2304 + 32),
                    f2 = *reinterpret_cast<const f16x8_t*>(rb + r * 2304 + 64);
      vh[r] = __builtin_elementwise_fma(va, f0, __builtin_elementwise_fma(vb, f1, vc * f2));
    }
    unsigned uoff = (unsigned)((n % WN_NSLOT) * WN_SLOT + wave * 1024);
    asm volatile("" : "+s"(uoff));
    unsigned char* dst = smem + lane16 + uoff;
    auto put = [&](int j, const f16x8_t& pa, const f16x8_t& pb) {      // row i0 + j of the x2 image: 3/4 pa + 1/4 pb, or zero padding
      const int y = i0 + j;
      const _Float16 wa = (y >= 0 && y < H) ? (_Float16)0.75f : (_Float16)0.f, wb = (y >= 0 && y < H) ? (_Float16)0.25f : (_Float16)0.f;
      const f16x8_t qa = {wa, wa, wa, wa, wa, wa, wa, wa}, qb = {wb, wb, wb, wb, wb, wb, wb, wb};
      *reinterpret_cast<f16x8_t*>(dst + j * WN_VROW) = __builtin_elementwise_fma(qa, pa, qb * pb);
    };
    if (i0 & 1) { put(0, vh[0], vh[1]); put(1, vh[1], vh[0]); put(2, vh[1], vh[2]); }
    else { put(0, vh[1], vh[0]); put(1, vh[1], vh[2]); put(2, vh[2], vh[1]); }
  };

  f32x4 acc[3][4];      // [row slot][xi]
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) acc[s][xi] = zinit;

  // The workgroup barrier of a batch sits behind its SECOND row for waves 4-7 (V of batch n + 1, written by every wave after its first
  // row, is then visible during their third row, whose MFMAs cover the first reads of the next batch) and behind the THIRD row for waves
  // 0-3.  Either way a wave has written its V fragments of batch n + 1 before it arrives, reads them only after it has passed, and
  // the slot a wave writes in batch n + 1 (V of n + 2, slot (n + 2) % 3) was last read in batch n - 1, which every wave has left when
  // any wave passes barrier n; a wave that is still in batch n's third row reads slot n % 3.
  auto sync = [&]() {
    if (DBG & 16) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  f16x8_t fv[2][4];     // V fragments [K half][xi] of the row in flight (steady state; carried from batch to batch)
  // epilogue of a completed accumulator set: y[2t] = M0 + M1 + M2, y[2t+1] = M1 - M2 - M3 (4 output channels each), activation, fp16;
  // lanes kg and kg ^ 1 then trade halves (v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second) so
  // that an even kg holds channels 8 (kg >> 1) .. + 7 of column 2t and an odd kg those of column 2t + 1: ONE 16-byte store per lane
  auto epilogue_math = [&](const f32x4 (&m)[4], u32x2& x, u32x2& y) {
    f16x4_t h0, h1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float ye, yo;
      if (DBG & 8) { ye = m[0][k] + m[2][k]; yo = m[1][k] + m[3][k]; }
      else { ye = m[0][k] + (m[1][k] + m[2][k]); yo = (m[1][k] - m[2][k]) - m[3][k]; }
      h0[k] = (_Float16)ye;
      h1[k] = (_Float16)yo;
    }
    if (!(DBG & 8)) {
      // activation on the packed fp16 values (8 vector instructions instead of 16): max(h, slope h).  The slope's own fp16 rounding
      // (0.1 -> 0.09998) and the second rounding touch negative results only, whose magnitude is a tenth of their pre-activation's:
      // their absolute error stays below that of the positive results
      h0 = __builtin_elementwise_max(h0, h0 * slope_h);
      h1 = __builtin_elementwise_max(h1, h1 * slope_h);
    }
    x = __builtin_bit_cast(u32x2, h0); y = __builtin_bit_cast(u32x2, h1);
  };
  auto epilogue_store = [&](u32x2 x, u32x2 y, __amdgpu_buffer_rsrc_t ro, unsigned vo, int so) {
    const auto s0 = __builtin_amdgcn_permlane16_swap(x[0], y[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(x[1], y[1], false, false);
    const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
    // Non-temporal stores for the plain (1x resolution) form: measured 0.221 -> 0.192 ms at 8 x 272x480, 0.034 -> 0.032 at one clip,
    // forward 104.52 / 104.27 -> 104.28 / 103.91 ms (same box, alternating).  The space-to-depth form from a materialised x2 source
    // measured 0.962 -> 1.040 ms stand-alone at eight clips of 544x960 (0.465 -> 0.406 at one clip of 1088x1920) and is left alone;
    // the UP form is launched with dbg 2048 (wino_launch_up).  dbg 2048 forces them, 8192 forbids them.
    constexpr bool NT = (DBG & 2048) || (!S2D && !UP && !(DBG & 8192));
    if (DBG & 4096) __builtin_amdgcn_raw_buffer_store_b128(v, ro, (int)vo, so, 1);        // experiment: sc0
    else if (!(DBG & 4)) __builtin_amdgcn_raw_buffer_store_b128(v, ro, (int)vo, so, NT ? 2 : 0);
    else if (v[0] == 0x12345678u && v[3] == 0x9abcdef0u) __builtin_amdgcn_raw_buffer_store_b128(v, ro, (int)vo, so, 0);
    // HAZARD (measured, gfx950): a vector instruction that overwrites a data register of a 128-bit buffer store within two wait states of
    // it reaches the store in some lanes.  hipcc pads that case only for stores WITHOUT a scalar offset register (it takes the offset's
    // extra issue cycle to cover it), this store has one, and the compiler did put a v_cndmask of the first data register right behind
    // it in one path: the first store of a unit, issued into an idle memory pipeline, then wrote that 0 / 1 in lanes 12-15 of every
    // row (tests/probe_wino_case.py; tools/check_store_hazard.py scans the assembly for the pattern).  The registers stay live, and two
    // wait states pass, through this statement.
    asm volatile("s_nop 1" : : "v"(v));
  };
  auto epilogue = [&](const f32x4 (&m)[4], __amdgpu_buffer_rsrc_t ro, unsigned vo, int so) {
    u32x2 x, y;
    epilogue_math(m, x, y);
    epilogue_store(x, y, ro, vo, so);
  };
  // scalar (row) part of a store address; the lane part `vo` carries the column, the channel half and (S2D) the x phase's plane
  auto row_offset = [&](int r) {
    if constexpr (S2D) return (((r & 1) * 2) * nck + cb) * ((H >> 1) * (W >> 1) * 32) + (r >> 1) * (W >> 1) * 32;
    else return (cb * (H * W) + r * W) * 32;
  };

  // one batch of three input rows.  ALL = every output row the batch touches is inside the segment (the steady state: straight-line
  // code, 72 MFMAs); otherwise the rows are guarded one by one (first / last batches of a unit)
  // chain: a steady-state batch of the same unit follows (this batch then prefetches its first fragments and, in waves 4-7, leaves
  // its last epilogue to it); fv_ready: the previous batch did so
  auto batch = [&](auto all_tag, int n, int b, int y0, int y1, int i0, unsigned vo, bool chain, bool fv_ready, auto&& mid) {
    constexpr bool ALL = decltype(all_tag)::value;
    unsigned voff = (unsigned)((n % WN_NSLOT) * WN_SLOT), voff_next = (unsigned)(((n + 1) % WN_NSLOT) * WN_SLOT);
    asm volatile("" : "+s"(voff), "+s"(voff_next));
    const unsigned char* vbase = smem + lane16 + voff;
    const unsigned char* vnext = smem + lane16 + voff_next;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.out) + (long long)b * out_img_bytes,
                                                                        0, (int)out_img_bytes, 0x00020000);
    if constexpr (ALL) {
      // steady state.  The eight V fragments of a row are requested in one go -- those of row j + 1 right behind row j's MFMAs, i.e.
      // BEFORE row j's epilogue, whose vector work covers their latency (with reads only two fragments ahead of their MFMAs every
      // triple of MFMAs waited ~100 cycles for the LDS: the first build of this kernel spent 2.3x its MFMA time per batch)
      // fragments of a row as two K halves of four (xi): fv[sc][xi].  Half `sc` of row j + 1 is requested right behind the MFMAs of
      // half `sc` of row j (its registers are free then): every read runs twelve MFMAs ahead of its use without a second buffer
      auto read_half = [&](const unsigned char* base, int jj, int sc) {
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) fv[sc][xi] = *reinterpret_cast<const f16x8_t*>(base + jj * WN_VROW + (xi * 2 + sc) * 1024);
      };
      unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // dbg 512: batch entry, row 0 MFMAs issued, its epilogue done, mid done,
      auto stamp = [&](int k) {                                  // row 1 + epilogue done, barrier passed, row 2 MFMAs issued, batch end
        if (DBG & 512) asm volatile("s_memtime %0" : "=s"(ts[k]) : : "memory");
      };
      stamp(0);
      if (!fv_ready || !grp_b) {      // (waves 0-3 pass the batch's barrier only at its end: nothing of this batch was theirs to prefetch)
        read_half(vbase, 0, 0);
        read_half(vbase, 0, 1);
      }
      // STAGGER (MI355X_MICROARCH.md, "two waves that run the same program with one barrier per block"): waves w and w + 4 share a
      // SIMD and would otherwise reach their MFMA runs and their vector-only epilogues together.  Waves 4-7 keep the batch's last
      // completed row in its accumulators across the barrier and convert / store it HERE, at the start of the next batch (the
      // accumulator set is re-opened by this batch's first row only after that), while their SIMD partner is in its MFMAs.
      if (DBG & 1024) __builtin_amdgcn_s_setprio(1);
      if (grp_b) epilogue(acc[1], ro, vo, row_offset(i0 - 2));
      __builtin_amdgcn_sched_barrier(0);
      auto row = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int s2 = (j + 2) % 3, s1 = j, s0 = (j + 1) % 3;
        // dbg 1024 (experiment): MFMA runs at LOW issue priority, the vector / memory work behind them at HIGH priority -- the SIMD's
        // arbiter serves priority, then age, so without this the older wave of a SIMD wins every slot and its partner trails it
        if (DBG & 1024) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) {
          // dy = 2 first: it completes output row i0 + j - 1, whose results have then landed when the epilogue starts
#pragma unroll
          for (int xi = 0; xi < 4; ++xi) {
            if (DBG & 1) { asm volatile("" :: "v"(fv[sc][xi])); continue; }
            acc[s2][xi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[2][xi][sc], fv[sc][xi], acc[s2][xi], 0, 0, 0);
          }
#pragma unroll
          for (int xi = 0; xi < 4; ++xi) {
            if (DBG & 1) { if (sc == 0) acc[s0][xi] = xi == 1 ? binit : zinit; continue; }
            acc[s1][xi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][xi][sc], fv[sc][xi], acc[s1][xi], 0, 0, 0);
            acc[s0][xi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][xi][sc], fv[sc][xi], sc == 0 ? (xi == 1 ? binit : zinit) : acc[s0][xi], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (j + 1 < WN_ROWS) read_half(vbase, j + 1, sc);
          else if (chain && grp_b) read_half(vnext, 0, sc);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j == 0) stamp(1);
        if (j == 2) stamp(6);
        if (DBG & 1024) __builtin_amdgcn_s_setprio(1);
        if (j < 2 || !(grp_b && chain)) epilogue(acc[s2], ro, vo, row_offset(i0 + j - 1));
        if (j == 0) stamp(2);
        if (j == 0) mid();
        if (j == 0) stamp(3);
        if (j == 1) stamp(4);
        // The batch's barrier sits at a DIFFERENT place in the two halves of the workgroup (in-kernel timeline, tools/wino_timeline.py:
        // the SIMD's arbiter serves the older wave first, waves 0-3 reached a common barrier ~1 000 cycles ahead of their partners and
        // idled there): waves 4-7 pass it behind their second row (and prefetch the next batch's first fragments in the third), waves 0-3
        // only at the end of the batch -- their third row needs nothing the barrier guards.  Slot arithmetic: see `sync`.
        if ((j == WN_BSYNC && grp_b) || (j == 2 && !grp_b)) sync();
        if (j == 1) stamp(5);
        __builtin_amdgcn_sched_barrier(0);
      };
      row(std::integral_constant<int, 0>{});
      row(std::integral_constant<int, 1>{});
      row(std::integral_constant<int, 2>{});
      if (DBG & 512) {
        stamp(7);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (a.clk && n == 8 && lane == 0) {
          unsigned long long* c = a.clk + ((long long)blockIdx.x * 8 + wave) * 8;
#pragma unroll
          for (int k = 0; k < 8; ++k) c[k] = ts[k];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < WN_ROWS; ++j) {
        // input row i0 + j: dy = 2 completes output row i0 + j - 1 (slot (j + 2) % 3), dy = 1 -> row i0 + j (slot j), dy = 0 opens
        // row i0 + j + 1 (slot (j + 1) % 3)
        const int r2 = i0 + j - 1, r1 = i0 + j, r0 = i0 + j + 1;
        const bool v2 = r2 >= y0 && r2 < y1, v1 = r1 >= y0 && r1 < y1, v0 = r0 >= y0 && r0 < y1;
        const int s2 = (j + 2) % 3, s1 = j, s0 = (j + 1) % 3;
        if (v0 || v1 || v2) {
#pragma unroll
          for (int sc = 0; sc < 2; ++sc) {
            f16x8_t fv[4];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) fv[xi] = *reinterpret_cast<const f16x8_t*>(vbase + j * WN_VROW + (xi * 2 + sc) * 1024);
            if (v2) {
#pragma unroll
              for (int xi = 0; xi < 4; ++xi) acc[s2][xi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[2][xi][sc], fv[xi], acc[s2][xi], 0, 0, 0);
            }
            if (v1) {
#pragma unroll
              for (int xi = 0; xi < 4; ++xi) acc[s1][xi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][xi][sc], fv[xi], acc[s1][xi], 0, 0, 0);
            }
            if (v0) {
#pragma unroll
              for (int xi = 0; xi < 4; ++xi)
                acc[s0][xi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][xi][sc], fv[xi], sc == 0 ? (xi == 1 ? binit : zinit) : acc[s0][xi], 0, 0, 0);
            }
          }
        }
        // (a row that is not stored still issues its store, out of range: every path then has the same vector-memory sequence)
        if (j < 2 || !(grp_b && chain)) epilogue(acc[s2], ro, v2 ? vo : 0x80000000u, row_offset(r2));
        if (j == 0) mid();
        if ((j == WN_BSYNC && grp_b) || (j == 2 && !grp_b)) sync();
      }
    }
  };

  // the next batch's place in the unit sequence (clamped at the end: the last batch is re-loaded, harmlessly)
  auto next_coords = [&](int n, int& b, int& x0, int& y0, int& y1, int& i0) {
    n = min(n, T - 1);
    const int ord = n / NB, m = n - ord * NB;
    unit_coords(ord, b, x0, y0, y1);
    i0 = y0 - 1 + m * WN_ROWS;
  };

  {
    int b, x0, y0, y1, i0;
    if constexpr (UP) {
      // raw rows of batches 0 and 1 staged, V of batch 0 formed, the unit of batch 2 in flight: the state `mid` expects
      next_coords(0, b, x0, y0, y1, i0);
      const int x00 = x0, i00 = i0;
      stage_load(b, x0, i0);
      stage_write(0);
      next_coords(1, b, x0, y0, y1, i0);
      stage_load(b, x0, i0);
      stage_write(1);
      sync();
      form_v(0, x00, i00);
      next_coords(2, b, x0, y0, y1, i0);
      stage_load(b, x0, i0);
    } else {
      unsigned va, vb;
      next_coords(0, b, x0, y0, y1, i0);
      src_lane_offsets(x0, va, vb);
      issue_loads(0, va, vb, b, y0, y1, i0);
      write_v(0);
      next_coords(1, b, x0, y0, y1, i0);
      src_lane_offsets(x0, va, vb);
      issue_loads(1, va, vb, b, y0, y1, i0);
    }
    if (!(DBG & 4)) {      // two out-of-range (dropped) stores stand for "the previous batch's rows 1 and 2" in the wait counts
      const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.out), 0, 0, 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, r0, (int)0x80000000u, 0, 0);
    }
    sync();
  }
  int n = 0;
  for (int ord = 0; ord < my_units; ++ord) {
    int b, x0, y0, y1;
    unit_coords(ord, b, x0, y0, y1);
    // lane parts of the store addresses (columns 2t, 2t + 1 of the strip) and of the NEXT batches' loads within this unit
    const int xo = x0 + 2 * t16 + (kg & 1);       // this lane stores column 2t + (kg & 1), channels 8 (kg >> 1) .. + 7 of the wave's 16
    unsigned vo, va, vb;
    // (half-split rows: a row's first 8-channel halves, then its second halves -- 16 lanes then store 256 contiguous bytes)
    if constexpr (S2D) vo = (a.halfsplit ? (unsigned)(xo >> 1) * 16u + (unsigned)(kg >> 1) * (unsigned)((W >> 1) * 16)
                                         : (unsigned)(xo >> 1) * 32u + (unsigned)(kg >> 1) * 16u) +
                            (unsigned)(kg & 1) * (unsigned)(nck * (H >> 1) * (W >> 1) * 32);
    else vo = (unsigned)xo * 32u + (unsigned)(kg >> 1) * 16u;
    if (xo >= W) vo = 0x80000000u;
    if (DBG & 64) vo = (unsigned)x0 * 32u + lane16;                             // ablation: line-contiguous stores (wrong layout)
    src_lane_offsets(x0, va, vb);
    // batch m covers input rows y0 - 1 + 3m ..+2; every output row it touches is inside the segment for 1 <= m < mf
    const int mf = (y1 - y0) / 3;
    // the V fragments of batch n + 1 and the loads of batch n + 2, issued after the batch's first row; inside the unit they need
    // nothing but the row number
    auto mid_same = [&](int nn, int m) {
      return [&, nn, m]() {
        if constexpr (UP) {
          // the unit of batch nn + 2 goes to its LDS slot, the one of batch nn + 3 is requested, V of batch nn + 1 is formed from
          // the slot written one batch ago (visible since the last barrier)
          stage_write(nn + 2);
          int b3, x3, y03, y13, i03;
          next_coords(nn + 3, b3, x3, y03, y13, i03);
          stage_load(b3, x3, i03);
          form_v(nn + 1, x0, y0 - 1 + (m + 1) * WN_ROWS);
        } else {
          write_v(nn + 1);
          issue_loads(nn + 2, va, vb, b, y0, y1, y0 - 1 + (m + 2) * WN_ROWS);
        }
      };
    };
    auto mid_any = [&](int nn) {
      return [&, nn]() {
        int b2, x2, y02, y12, i02;
        if constexpr (UP) {
          stage_write(nn + 2);
          next_coords(nn + 3, b2, x2, y02, y12, i02);
          stage_load(b2, x2, i02);
          next_coords(nn + 1, b2, x2, y02, y12, i02);
          form_v(nn + 1, x2, i02);
        } else {
          write_v(nn + 1);
          unsigned va2, vb2;
          next_coords(nn + 2, b2, x2, y02, y12, i02);
          src_lane_offsets(x2, va2, vb2);
          issue_loads(nn + 2, va2, vb2, b2, y02, y12, i02);
        }
      };
    };
    // steady-state batches: 1 <= m < mfast (the last two batches of a unit prefetch across the unit boundary through the general
    // path).  A batch defers its last row's epilogue (waves 4-7) exactly when a steady-state batch of the same unit follows it.
    const int mfast = min(mf, NB - 2);
    batch(std::false_type{}, n, b, y0, y1, y0 - 1, vo, 1 < mfast, false, mid_any(n));
    ++n;
    int m = 1;
    for (; m < mfast; ++m, ++n)
      batch(std::true_type{}, n, b, y0, y1, y0 - 1 + m * WN_ROWS, vo, m + 1 < mfast, m > 1, mid_same(n, m));
    for (; m < NB; ++m, ++n)
      batch(std::false_type{}, n, b, y0, y1, y0 - 1 + m * WN_ROWS, vo, false, false, mid_any(n));
  }
}

// [Cout][64][3][3] fp32 -> [Cout/16][dy][xi][K half][lane = kg * 16 + i][8] fp16 (the MFMA A fragments of the kernel above)
__global__ __launch_bounds__(256) void pack_wino_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int Cout) {
  const int total = Cout * 64 * 12;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 7, ln = (idx >> 3) & 63, f = (idx >> 9) % 24, cb = idx / (512 * 24);
    const int sc = f & 1, xi = (f >> 1) & 3, dy = f >> 3;
    const int o = cb * 16 + (ln & 15), c = sc * 32 + (ln >> 4) * 8 + e;
    const float* g = w + ((long long)o * 64 + c) * 9 + dy * 3;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    const float u = xi == 0 ? g0 : (xi == 1 ? 0.5f * (g0 + g1 + g2) : (xi == 2 ? 0.5f * (g0 - g1 + g2) : g2));
    out[idx] = (_Float16)u;
  }
}

}  // namespace

extern "C" int cdfo_pack_conv3x3_wino(const float* w_oihw, void* packed, int Cout, void* stream) {
  if (Cout <= 0 || Cout % 16) return CDFO_EINVAL;
  hipLaunchKernelGGL(pack_wino_kernel, dim3((Cout * 64 * 12 + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w_oihw,
                     static_cast<_Float16*>(packed), Cout);
  CDFO_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Host side.  Row segments are chosen so that the unit count fills the workgroup lanes evenly: a unit costs
// 3 * ceil((seg_h + 2) / 3) row steps (+ one barrier per batch).
namespace {
int wino_launch_up(const wino_args& a, int grid, hipStream_t st) {
  static CdfoAttrOnce once;
  // (dbg 2048 = non-temporal stores: the 2.1 GB intermediate of the x2 branch is read back once, by a kernel that starts after this one
  // has finished; same-box A/B of the forward 104.65 / 104.54 -> 103.98 / 104.11 ms)
  const hipError_t e = cdfo_set_max_lds(once, reinterpret_cast<const void*>(conv3x3_c64_wino_kernel<true, 2048, true>), WN_LDS_UP);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((conv3x3_c64_wino_kernel<true, 2048, true>), dim3(grid), dim3(WN_THREADS), WN_LDS_UP, st, a);
  return 0;
}

template <int DBG>
int wino_launch(const wino_args& a, int grid, hipStream_t st) {
  static CdfoAttrOnce once_s, once_p;       // 72 KiB of dynamic LDS: above the 64 KiB a kernel gets without asking
  if (a.s2d) {
    const hipError_t e = cdfo_set_max_lds(once_s, reinterpret_cast<const void*>(conv3x3_c64_wino_kernel<true, DBG>), WN_LDS);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv3x3_c64_wino_kernel<true, DBG>), dim3(grid), dim3(WN_THREADS), WN_LDS, st, a);
  } else {
    const hipError_t e = cdfo_set_max_lds(once_p, reinterpret_cast<const void*>(conv3x3_c64_wino_kernel<false, DBG>), WN_LDS);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv3x3_c64_wino_kernel<false, DBG>), dim3(grid), dim3(WN_THREADS), WN_LDS, st, a);
  }
  return 0;
}
}  // namespace

// dbg: 0 (developer ablation bits otherwise, WRONG results: 1 no MFMAs, 2 no global loads, 4 no stores, 8 no epilogue arithmetic,
// 16 no barrier)
extern "C" int cdfo_conv3x3_c64_wino_dbg(const void* src_cp16, int B, int H, int W, const void* w_wino, const float* bias, int Cout, int act,
                                         void* out_cp16, int store_mode, int dbg, void* clk_probe, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (B <= 0 || H <= 0 || W <= 0 || (W & 1) || Cout <= 0 || Cout % 128) return CDFO_EINVAL;
  if (act == CDFO_ACT_SIGMOID || (store_mode != CDFO_STORE_PLAIN && store_mode != CDFO_STORE_S2D && store_mode != CDFO_STORE_S2D_HS)) return CDFO_EINVAL;
  if (store_mode != CDFO_STORE_PLAIN && (H & 1)) return CDFO_EINVAL;
  if ((long long)H * W * 128 >= (1ll << 31) || (long long)(Cout / 16) * H * W * 32 >= (1ll << 31)) return CDFO_EINVAL;   // per-image 32-bit offsets
  if (!aligned16(src_cp16) || !aligned16(w_wino) || !aligned16(out_cp16)) return CDFO_EALIGN;
  const int cus = cdfo_num_cus();
  const int nhalf = Cout / 128;
  if (cus < 8 || cus / 8 < nhalf) return CDFO_EINVAL;
  const int nslots = cus / 8, npairs = nslots / nhalf, lanes = npairs * 8;
  const int nstrips = (W + 31) / 32;
  int best_seg = H;
  double best_cost = 1e30;
  for (int nseg = 1; nseg <= (H + 7) / 8; ++nseg) {
    const int seg_h = (H + nseg - 1) / nseg;
    if ((seg_h * (nseg - 1)) >= H) continue;         // an empty last segment
    const long long units = (long long)B * nstrips * nseg;
    const long long rounds = (units + lanes - 1) / lanes;
    const double cost = (double)rounds * (3.0 * ((seg_h + 2 + 2) / 3) + 1.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best_seg = seg_h; }
  }
  wino_args a;
  a.src = static_cast<const unsigned char*>(src_cp16); a.B = B; a.H = H; a.W = W;
  a.w = static_cast<const f16x8_t*>(w_wino); a.bias = bias; a.Cout = Cout; a.act = act;
  a.out = static_cast<_Float16*>(out_cp16); a.s2d = store_mode != CDFO_STORE_PLAIN; a.halfsplit = store_mode == CDFO_STORE_S2D_HS;
  a.clk = static_cast<unsigned long long*>(clk_probe);
  a.seg_h = best_seg; a.nseg = (H + best_seg - 1) / best_seg; a.nstrips = nstrips; a.nb = (best_seg + 2 + 2) / 3;
  const double px = (double)B * H * W;
  CdfoProfScope prof(st, KID_CONV3_WINO, 2.0 * px * Cout * 64 * 9, 2.0 * (px * Cout + px * 64) + 2.0 * 12 * 64 * Cout);
  int rc = 0;
  if (dbg == -2) {       // developer timeline of the UP form (tools/wino_timeline.py --up): dbg 512 | 2048
    static CdfoAttrOnce once_tl;
    const hipError_t e = cdfo_set_max_lds(once_tl, reinterpret_cast<const void*>(conv3x3_c64_wino_kernel<true, 2560, true>), WN_LDS_UP);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv3x3_c64_wino_kernel<true, 2560, true>), dim3(nslots * 8), dim3(WN_THREADS), WN_LDS_UP, st, a);
    CDFO_LAUNCH_CHECK();
    return 0;
  }
  if (dbg == -1) {       // cdfo_conv3x3_c64_wino_up2
    rc = wino_launch_up(a, nslots * 8, st);
    if (rc) return rc;
    CDFO_LAUNCH_CHECK();
    return 0;
  }
  switch (dbg) {
    case 0: rc = wino_launch<0>(a, nslots * 8, st); break;
    case 1: rc = wino_launch<1>(a, nslots * 8, st); break;
    case 2: rc = wino_launch<2>(a, nslots * 8, st); break;
    case 4: rc = wino_launch<4>(a, nslots * 8, st); break;
    case 8: rc = wino_launch<8>(a, nslots * 8, st); break;
    case 12: rc = wino_launch<12>(a, nslots * 8, st); break;
    case 16: rc = wino_launch<16>(a, nslots * 8, st); break;
    case 6: rc = wino_launch<6>(a, nslots * 8, st); break;
    case 14: rc = wino_launch<14>(a, nslots * 8, st); break;
    case 30: rc = wino_launch<30>(a, nslots * 8, st); break;
    case 31: rc = wino_launch<31>(a, nslots * 8, st); break;
    case 32: rc = wino_launch<32>(a, nslots * 8, st); break;
    case 64: rc = wino_launch<64>(a, nslots * 8, st); break;
    case 96: rc = wino_launch<96>(a, nslots * 8, st); break;
    case 128: rc = wino_launch<128>(a, nslots * 8, st); break;
    case 256: rc = wino_launch<256>(a, nslots * 8, st); break;
    case 512: rc = wino_launch<512>(a, nslots * 8, st); break;
    case 1024: rc = wino_launch<1024>(a, nslots * 8, st); break;
    case 1536: rc = wino_launch<1536>(a, nslots * 8, st); break;
    case 2048: rc = wino_launch<2048>(a, nslots * 8, st); break;
    case 4096: rc = wino_launch<4096>(a, nslots * 8, st); break;
    case 8192: rc = wino_launch<8192>(a, nslots * 8, st); break;
    default: return CDFO_EINVAL;
  }
  if (rc) return rc;
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_conv3x3_c64_wino(const void* src_cp16, int B, int H, int W, const void* w_wino, const float* bias, int Cout, int act,
                                     void* out_cp16, int store_mode, void* stream) {
  return cdfo_conv3x3_c64_wino_dbg(src_cp16, B, H, W, w_wino, bias, Cout, act, out_cp16, store_mode, 0, nullptr, stream);
}

// Block_'s double-resolution branch without its double-resolution source: src_lr_cp16 [B][4][H/2][W/2][16] = up.0(x) (cdfo_block_prologue2's
// t16), H x W = the x2 image's size; result = cdfo_conv3x3_c64_wino(bilinear_x2(src), ..., CDFO_STORE_S2D) up to fp16 rounding points
extern "C" int cdfo_conv3x3_c64_wino_up2(const void* src_lr_cp16, int B, int H, int W, const void* w_wino, const float* bias, int Cout, int act,
                                         void* out_cp16, int store_mode, void* stream) {
  if ((H & 3) || (W & 3) || (store_mode != CDFO_STORE_S2D && store_mode != CDFO_STORE_S2D_HS)) return CDFO_EINVAL;
  return cdfo_conv3x3_c64_wino_dbg(src_lr_cp16, B, H, W, w_wino, bias, Cout, act, out_cp16, store_mode, -1, nullptr, stream);
}
