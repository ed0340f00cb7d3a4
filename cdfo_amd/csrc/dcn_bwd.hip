// Deformable convolution BACKWARD (DCNv1 and modulated DCNv2) for gfx950 -- SURVEY section 8f n2.
//
// Replaces, without the `columns` [C*kh*kw, Ho*Wo] HBM buffer the reference materialises twice per image,
//   columns = W^T x grad_output                                            (ops/dcn/src/deform_conv_cuda.cpp:617-620, 331-336)
//   modulated_deformable_col2im_coord_gpu_kernel / deformable_col2im_coord  (cu:700-766, 373-430)  -> grad_offset, grad_mask
//   modulated_deformable_col2im_gpu_kernel / deformable_col2im              (cu:634-698, 283-330)  -> grad_input
//   im2col + grad_weight += grad_output x columns^T, grad_bias += rowsum    (cpp:637-664, 447-476)
//
// Three kernels:
//   dcn_bwd_data_kernel    one thread = (image, deformable group, tap, output pixel).  The sample position and its
//                          bilinear / coordinate weights depend on exactly that tuple, so they are computed once and the
//                          thread walks the C/dg channels of its deformable group four at a time: column gradient =
//                          dot(W[:, c, tap], grad_output[:, pixel]) (weight operand is wave-uniform -> scalar loads;
//                          grad_output reads are coalesced along the pixel axis and shared by the four channels),
//                          grad_offset / grad_mask are reduced in registers and
//                          ASSIGNED (deterministic, as in the reference), grad_input takes 4 hardware fp32 atomics.
//   dcn_bwd_weight_kernel  one workgroup = one input channel x 64 output channels x a range of (image, pixel) positions,
//                          walked in chunks of 64: sampled column values [T][64] and the grad_output tile [64][64] are
//                          staged in LDS, each thread owns up to 16 (cout, tap) pairs; one atomic per pair per workgroup.
//   dcn_bwd_bias_kernel    (output channel, slice of positions) workgroups, one atomic each.
// grad_input, grad_weight and grad_bias are accumulated into (callers zero them: ops/dcn/deform_conv.py:71-72, 85, 154-158).
#include "common.h"

namespace {

struct DcnBwdArgs {
  const float* in; const float* offset; const float* mask; const float* w; const float* gout;
  float* gin; float* goff; float* gmask; float* gw; float* gbias;
  int B, C, H, W, Co, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg;
  float scale;
  const _Float16* gt;       // GW form: grad_output per 8 x 32 tile as fp16 [B][tile][hi|lo][64 o][256 pixels] (dcn_bwd_gprep_kernel)
  const float* gt_inv;      //          and the power of two that undoes each tile's scaling
  float* col;               //          masked column values [B][dg][tile][4 * 9][256 pixels] written by the data kernel
};

struct Sample {             // bilinear footprint of one sample position (cu:466-496, 498-567)
  int o1, o2, o3, o4;       // plane offsets of the 4 corners; -1 = outside the image (contributes 0)
  float hh, hw, lh, lw;
  bool valid;
};

__device__ __forceinline__ Sample make_sample(float h, float w, int H, int W) {
  Sample s;
  s.valid = h > -1.f && w > -1.f && h < (float)H && w < (float)W;     // false for NaN / inf offsets as well
  if (!s.valid) h = w = 0.f;                                           // no float -> int conversion of a non-finite value
  const int hl = (int)floorf(h), wl = (int)floorf(w), hhi = hl + 1, whi = wl + 1;
  s.lh = h - (float)hl; s.lw = w - (float)wl; s.hh = 1.f - s.lh; s.hw = 1.f - s.lw;
  s.o1 = (s.valid && hl >= 0 && wl >= 0) ? hl * W + wl : -1;
  s.o2 = (s.valid && hl >= 0 && whi <= W - 1) ? hl * W + whi : -1;
  s.o3 = (s.valid && hhi <= H - 1 && wl >= 0) ? hhi * W + wl : -1;
  s.o4 = (s.valid && hhi <= H - 1 && whi <= W - 1) ? hhi * W + whi : -1;
  return s;
}

__global__ __launch_bounds__(256) void dcn_bwd_data_kernel(DcnBwdArgs a) {
  const int T = a.kh * a.kw, P = a.Ho * a.Wo;
  const int Cg = a.C / a.groups, Cog = a.Co / a.groups, Cdg = a.C / a.dg;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int d = blockIdx.y / T, t = blockIdx.y - d * T, b = blockIdx.z;
  if (p >= P) return;
  const int ho = p / a.Wo, wo = p - ho * a.Wo, ki = t / a.kw, kj = t - ki * a.kw;
  const long long ob = ((long long)(b * a.dg + d) * T + t) * 2 * P + p;
  const float h_im = (float)(ho * a.sh - a.ph + ki * a.dh) + a.offset[ob];
  const float w_im = (float)(wo * a.sw - a.pw + kj * a.dw) + a.offset[ob + P];
  const long long mb = ((long long)(b * a.dg + d) * T + t) * P + p;
  const float m = a.mask ? a.mask[mb] : 1.f;
  const Sample s = make_sample(h_im, w_im, a.H, a.W);
  const float w1 = s.hh * s.hw, w2 = s.hh * s.lw, w3 = s.lh * s.hw, w4 = s.lh * s.lw;
  float vh = 0.f, vw = 0.f, mv = 0.f;
  if (s.valid) {
    for (int cc0 = 0; cc0 < Cdg; cc0 += 4) {
      // column gradients of up to 4 channels at once: each grad_output value is loaded once per chunk, the weights are
      // wave-uniform (scalar loads)
      float cg4[4] = {0.f, 0.f, 0.f, 0.f};
      const int c_first = d * Cdg + cc0, g = c_first / Cg;
      const int nc = (Cdg - cc0) < 4 ? (Cdg - cc0) : 4;
      const bool one_group = (c_first + nc - 1) / Cg == g;          // the chunk lies inside one conv group (the usual case)
      if (one_group) {
        const int cl = c_first - g * Cg;
        const float* wp = a.w + ((long long)(g * Cog) * Cg + cl) * T + t;        // + o * Cg * T + e * T   (wave-uniform)
        const float* gp = a.gout + ((long long)b * a.Co + g * Cog) * P + p;      // + o * P                (coalesced)
        for (int o = 0; o < Cog; ++o) {
          const float gv = gp[(long long)o * P];
          const float* wo = wp + (long long)o * Cg * T;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nc) cg4[e] = fmaf(wo[e * T], gv, cg4[e]);
        }
      } else {
        for (int e = 0; e < nc; ++e) {
          const int c = c_first + e, ge = c / Cg, cl = c - ge * Cg;
          const float* wp = a.w + ((long long)(ge * Cog) * Cg + cl) * T + t;
          const float* gp = a.gout + ((long long)b * a.Co + ge * Cog) * P + p;
          for (int o = 0; o < Cog; ++o) cg4[e] = fmaf(wp[(long long)o * Cg * T], gp[(long long)o * P], cg4[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (e >= nc) continue;
        const int c = c_first + e;
        const float cg = cg4[e];
        const long long pl = ((long long)b * a.C + c) * a.H * a.W;
        const float* im = a.in + pl;
        const float v1 = s.o1 >= 0 ? im[s.o1] : 0.f, v2 = s.o2 >= 0 ? im[s.o2] : 0.f;
        const float v3 = s.o3 >= 0 ? im[s.o3] : 0.f, v4 = s.o4 >= 0 ? im[s.o4] : 0.f;
        mv = fmaf(cg, w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4, mv);
        const float tg = cg * m;
        vh = fmaf(s.hw * (v3 - v1) + s.lw * (v4 - v2), tg, vh);                    // d sample / d h
        vw = fmaf(s.hh * (v2 - v1) + s.lh * (v4 - v3), tg, vw);                    // d sample / d w
        if (a.gin) {
          float* gi = a.gin + pl;
          if (s.o1 >= 0) unsafeAtomicAdd(gi + s.o1, w1 * tg);
          if (s.o2 >= 0) unsafeAtomicAdd(gi + s.o2, w2 * tg);
          if (s.o3 >= 0) unsafeAtomicAdd(gi + s.o3, w3 * tg);
          if (s.o4 >= 0) unsafeAtomicAdd(gi + s.o4, w4 * tg);
        }
      }
    }
  }
  if (a.goff) { a.goff[ob] = vh; a.goff[ob + P] = vw; }
  if (a.gmask) a.gmask[mb] = mv;
}

// ---- data gradients, tiled form (groups == 1, C/dg == 4, kh*kw <= 9: the alignment module's shape) ----------------------
// One workgroup = an 8 x 32 tile of output pixels x one deformable group; one thread = one pixel.
//   * the thread's 64 grad_output values are read once (coalesced along x) and turned into all 4 x kh*kw column gradients
//     of its deformable group with wave-uniform (scalar-loaded) weights: Co * 4 * kh*kw FMAs per pixel, no re-reads;
//   * with 4 channels per group the thread is the only contributor to grad_offset / grad_mask of (group, tap, pixel):
//     plain coalesced stores, deterministic;
//   * grad_input is scattered into an LDS window covering the tile's sampling footprint + a 5-pixel margin, as 64-bit
//     FIXED-POINT integer atomics: ds_add_f32 retires one wave-instruction per ~80 ns per CU on gfx950, ds_add_u64 one per
//     3-5 ns (tools/probe/lds_atomic_probe.hip) -- the float form was 9.4 of the kernel's 12.3 ms.  The scale is a power
//     of two chosen per workgroup from max |column gradient x mask| (every contribution is at most that; 2^12 of them
//     per cell cannot overflow), values keep >= 38 significant bits, and integer sums do not depend on the order of the
//     adds, so the window -- unlike a float-atomic one -- is bit-reproducible.  Only samples that leave the window go to
//     global memory one by one.  The window is flushed once, skipping zeros: ~3.7 k global atomics per workgroup instead
//     of 256 * kh*kw * 16 (a 16 x 32 tile was no faster).
//   * GW form (grad_weight requested, Co <= 64, workspace given): the sampled, masked column values the thread has in
//     hand anyway are stored ([image][group][tile][4 * kh*kw][256 pixels] fp32, 256 contiguous bytes per wave-instruction),
//     and dcn_bwd_gw_kernel contracts them with grad_output on the matrix cores -- the weight-gradient kernel's own
//     sampling pass (5.9 ms at the alignment shape) disappears.  (Contracting inside this kernel -- column image in LDS,
//     sums in registers over a run of tiles -- was tried: 213 VGPRs and 66 KB of LDS took this latency-bound kernel from
//     12 to 8 waves per CU and it ran 3x slower than the two kernels do together.)
constexpr int DT_Y = 8, DT_X = 32, DT_MARGIN = 5, DT_MAXT = 9, DT_THREADS = DT_Y * DT_X;
constexpr int DT_COLROWS = 4 * DT_MAXT;                          // (e, t) rows of the column image

typedef _Float16 dcnb_f16x8 __attribute__((ext_vector_type(8)));

// power of two s with amax * s in [2^13, 2^14): the scaled values sit well inside fp16's range, hi + lo keep 22 bits
__device__ __forceinline__ float dcnb_pow2_scale(float amax, float& inv) {
  int ex = 0;
  if (amax > 0.f && amax < INFINITY) frexpf(amax, &ex);          // amax = m * 2^ex, m in [0.5, 1)
  ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
  inv = ldexpf(1.f, ex - 14);
  return ldexpf(1.f, 14 - ex);
}

__device__ __forceinline__ void dcnb_split8(const f32x4& v0, const f32x4& v1, float sc, dcnb_f16x8& hi, dcnb_f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = v0[j] * sc, x1 = v1[j] * sc;
    hi[j] = (_Float16)x0;     lo[j] = (_Float16)(x0 - (float)hi[j]);
    hi[4 + j] = (_Float16)x1; lo[4 + j] = (_Float16)(x1 - (float)hi[4 + j]);
  }
}

// grad_output -> per-tile transposed, scaled, split copy for the GW form: grid (tiles, B), one thread per pixel of the
// 8 x 32 tile (row-major index = the thread index = the pixel index the consumer uses), Co <= 64 (absent channels: zeros).
__global__ __launch_bounds__(DT_THREADS) void dcn_bwd_gprep_kernel(const float* __restrict__ gout, _Float16* __restrict__ gt,
                                                                  float* __restrict__ gt_inv, int Co, int Ho, int Wo,
                                                                  int tiles_x, int ntiles) {
  __shared__ float s_max[DT_THREADS / 64];
  const int tid = threadIdx.x, tile = blockIdx.x, b = blockIdx.y, P = Ho * Wo;
  const int ho = (tile / tiles_x) * DT_Y + (tid >> 5), wo = (tile % tiles_x) * DT_X + (tid & 31);
  const bool pvalid = ho < Ho && wo < Wo;
  const float* gp = gout + (long long)b * Co * P + (pvalid ? ho * Wo + wo : 0);
  float g[64];
  float amax = 0.f;
#pragma unroll
  for (int o = 0; o < 64; ++o) {
    g[o] = (pvalid && o < Co) ? gp[(long long)o * P] : 0.f;
    amax = fmaxf(amax, fabsf(g[o]));
  }
  amax = wave_max(amax);
  if ((tid & 63) == 0) s_max[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
  float inv;
  const float sc = dcnb_pow2_scale(amax, inv);
  if (tid == 0) gt_inv[(long long)b * ntiles + tile] = inv;
  _Float16* base = gt + ((long long)b * ntiles + tile) * (2 * 64 * DT_THREADS) + tid;
#pragma unroll
  for (int o = 0; o < 64; ++o) {
    const float x = g[o] * sc;
    const _Float16 hi = (_Float16)x;
    base[o * DT_THREADS] = hi;
    base[(64 + o) * DT_THREADS] = (_Float16)(x - (float)hi);
  }
}

template <bool T9, bool GW>
__global__ __launch_bounds__(DT_THREADS) void dcn_bwd_data_tile_kernel(DcnBwdArgs a, int WH, int WW, int tiles_x, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) long long win[];      // [4][WH][WW], fixed point (see above)
  __shared__ float s_amax[DT_THREADS / 64];
  const int tid = threadIdx.x;
  const int T = T9 ? 9 : a.kh * a.kw, P = a.Ho * a.Wo;
  const int d = blockIdx.y, b = blockIdx.z, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * DT_Y, tx0 = (tile % tiles_x) * DT_X;
  const int ho = ty0 + (tid >> 5), wo = tx0 + (tid & 31);
  const bool pvalid = ho < a.Ho && wo < a.Wo;
  const int p = pvalid ? ho * a.Wo + wo : 0;
  const int wy0 = ty0 * a.sh - a.ph - DT_MARGIN, wx0 = tx0 * a.sw - a.pw - DT_MARGIN;
  const int wsize = 4 * WH * WW;
  // GW: this (image, group, tile)'s column image [4 * 9][256 pixels] in the workspace; thread = pixel = column index
  float* const colp = GW ? a.col + (((long long)b * a.dg + d) * ntiles + tile) * (DT_COLROWS * DT_THREADS) + tid : nullptr;
  if (a.gin)
    for (int i = tid; i < wsize; i += DT_THREADS) win[i] = 0;
  float cg[4][DT_MAXT];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int t = 0; t < DT_MAXT; ++t) cg[e][t] = 0.f;
  if (pvalid) {
    const float* gp = a.gout + (long long)b * a.Co * P + p;
    const float* wd = a.w + (long long)(4 * d) * T;                    // + o * C * T + e * T + t   (wave-uniform: scalar loads;
    for (int o = 0; o < a.Co; ++o) {                                   //  staging them in LDS instead was 9 % slower)
      const float gv = gp[(long long)o * P];
      const float* wo_ = wd + (long long)o * a.C * T;
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < DT_MAXT; ++t)
          if (T9 || t < T) cg[e][t] = fmaf(wo_[e * T + t], gv, cg[e][t]);
    }
  }
  // masks of the nine taps (also needed for the fixed-point scale below)
  float mk[DT_MAXT];
#pragma unroll
  for (int t = 0; t < DT_MAXT; ++t)
    mk[t] = (pvalid && (T9 || t < T) && a.mask) ? a.mask[((long long)(b * a.dg + d) * T + t) * P + p] : 1.f;
  float fx_scale = 1.f, fx_inv = 1.f;
  bool fx_ok = true;
  if (a.gin) {
    // workgroup maximum of |column gradient x mask| -> power-of-two scale 2^(48 - e), 2^e >= max: a contribution is below
    // 2^48 in magnitude, 2^12 of them (256 pixels x 9 taps, bilinear weights <= 1) stay inside 63 bits
    float amax = 0.f;
#pragma unroll
    for (int t = 0; t < DT_MAXT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = fabsf(cg[e][t] * mk[t]);
        amax = v <= 3.0e38f ? fmaxf(amax, v) : INFINITY;               // NaN / infinity: no fixed point for this workgroup
      }
    amax = wave_max(amax);
    if ((tid & 63) == 0) s_amax[tid >> 6] = amax;
    __syncthreads();                                                   // (also: the zeroed window is visible)
    amax = s_amax[0];
#pragma unroll
    for (int i = 1; i < DT_THREADS / 64; ++i) amax = fmaxf(amax, s_amax[i]);
    int ex = 0;
    if (amax > 0.f && amax < INFINITY) frexpf(amax, &ex);              // amax < 2^ex
    // fixed point only while 2^(48 - ex) is a representable scale: a finite maximum in [2^120, 2^128) would be scaled by the
    // CLAMPED exponent and overflow the 63-bit sums, one below 2^-60 would keep fewer than 48 bits (measured: 1.7 % error at
    // |grad_output| ~ 1e-30) -> such a workgroup takes the float path like a non-finite one
    fx_ok = amax < INFINITY && ex <= 120 && ex >= -60;
    ex = ex < -60 ? -60 : (ex > 120 ? 120 : ex);
    fx_scale = ldexpf(1.f, 48 - ex);
    fx_inv = ldexpf(1.f, ex - 48);     // !fx_ok: every sample goes straight to global memory as a float (NaN / inf propagate)
  }
#pragma unroll
  for (int t = 0; t < DT_MAXT; ++t) {
    if (!T9 && t >= T) {                                               // (GW: absent taps are zero rows of the image)
      if (GW)
#pragma unroll
        for (int e = 0; e < 4; ++e) colp[(e * DT_MAXT + t) * DT_THREADS] = 0.f;
      continue;
    }
    float colv[4] = {0.f, 0.f, 0.f, 0.f};
    if (pvalid) {
      const int ki = t / a.kw, kj = t - ki * a.kw;
      const long long ob = ((long long)(b * a.dg + d) * T + t) * 2 * P + p;
      const long long mb = ((long long)(b * a.dg + d) * T + t) * P + p;
      const float h_im = (float)(ho * a.sh - a.ph + ki * a.dh) + a.offset[ob];
      const float w_im = (float)(wo * a.sw - a.pw + kj * a.dw) + a.offset[ob + P];
      const float m = mk[t];
      float vh = 0.f, vw = 0.f, mv = 0.f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W) {
        const int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
        const float lh = h_im - (float)hl, lw = w_im - (float)wl, hh = 1.f - lh, hw = 1.f - lw;
        const bool r0 = hl >= 0, r1 = hl + 1 <= a.H - 1, c0 = wl >= 0, c1 = wl + 1 <= a.W - 1;
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const int o00 = hl * a.W + wl;
        const int ly = hl - wy0, lx = wl - wx0;                        // window coordinates of the top-left corner
        const bool inwin = fx_ok && ly >= 0 && ly + 1 < WH && lx >= 0 && lx + 1 < WW;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const long long pl = ((long long)b * a.C + 4 * d + e) * a.H * a.W;
          const float* im = a.in + pl;
          const float v1 = (r0 && c0) ? im[o00] : 0.f, v2 = (r0 && c1) ? im[o00 + 1] : 0.f;
          const float v3 = (r1 && c0) ? im[o00 + a.W] : 0.f, v4 = (r1 && c1) ? im[o00 + a.W + 1] : 0.f;
          const float cgv = cg[e][t];
          const float sv = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
          colv[e] = sv * m;
          mv = fmaf(cgv, sv, mv);
          const float tg = cgv * m;
          vh = fmaf(hw * (v3 - v1) + lw * (v4 - v2), tg, vh);
          vw = fmaf(hh * (v2 - v1) + lh * (v4 - v3), tg, vw);
          if (a.gin) {
            if (inwin) {
              unsigned long long* wp = reinterpret_cast<unsigned long long*>(win) + (e * WH + ly) * WW + lx;
              const float ts = tg * fx_scale;                          // |ts| < 2^48
              if (r0 && c0) atomicAdd(wp, (unsigned long long)(long long)(w1 * ts));
              if (r0 && c1) atomicAdd(wp + 1, (unsigned long long)(long long)(w2 * ts));
              if (r1 && c0) atomicAdd(wp + WW, (unsigned long long)(long long)(w3 * ts));
              if (r1 && c1) atomicAdd(wp + WW + 1, (unsigned long long)(long long)(w4 * ts));
            } else {
              float* gi = a.gin + pl + o00;
              if (r0 && c0) unsafeAtomicAdd(gi, w1 * tg);
              if (r0 && c1) unsafeAtomicAdd(gi + 1, w2 * tg);
              if (r1 && c0) unsafeAtomicAdd(gi + a.W, w3 * tg);
              if (r1 && c1) unsafeAtomicAdd(gi + a.W + 1, w4 * tg);
            }
          }
        }
      }
      if (a.goff) { a.goff[ob] = vh; a.goff[ob + P] = vw; }
      if (a.gmask) a.gmask[mb] = mv;
    }
    if (GW)
#pragma unroll
      for (int e = 0; e < 4; ++e) colp[(e * DT_MAXT + t) * DT_THREADS] = colv[e];     // 256 contiguous bytes per wave
  }
  if (a.gin) {
    __syncthreads();
    const int plane = WH * WW;
    for (int i = tid; i < wsize; i += DT_THREADS) {
      const long long vi = win[i];
      if (vi == 0) continue;
      const float v = (float)vi * fx_inv;
      const int e = i / plane, rr = i - e * plane, ly = rr / WW, lx = rr - ly * WW;
      const int yy = wy0 + ly, xx = wx0 + lx;
      if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
        unsafeAtomicAdd(a.gin + (((long long)b * a.C + 4 * d + e) * a.H + yy) * a.W + xx, v);
    }
  }
}

// ---- GW form, second half: gw[o][4d + e][t] += scale * sum over pixels of gout[o][p] col[(e,t)][p] on the matrix cores.
// Workgroup = one deformable group x a run of tiles of one image; wave w takes pixels 64w .. 64w+63 of every tile (K = 4 steps
// of 16).  A fragment (lane = output channel, 8 consecutive pixels) = one 16-byte load per fp16 half from the transposed
// copy; B fragment (lane = (e,t) row, the same 8 pixels) = two 16-byte loads of fp32 column values, scaled by the wave's
// power of two for this tile (max over its 36 x 64 values) and split hi + lo.  The running sums stay in registers, in units
// of the current tile's scales (a change multiplies them by a power of two: exact); at the end the four waves are added in
// a fixed order through LDS and leave as Co * 4 * kh*kw global atomics per workgroup.
__global__ __launch_bounds__(DT_THREADS) void dcn_bwd_gw_kernel(DcnBwdArgs a, int ntiles, int tpw) {
  __shared__ float red[4 * 64 * DT_COLROWS];                           // [wave][o][(e,t)]: 36,864 bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r = lane & 31;
  const int d = blockIdx.y, b = blockIdx.z, T = a.kh * a.kw;
  f32x16 gacc[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) gacc[i >> 1][i & 1][q] = 0.f;
  // the accumulators' unit as a binary EXPONENT (a float product of two tile scales can leave fp32's range); it only ever
  // grows: a tile whose own unit is smaller has its column values scaled down to the running one instead (a power of two,
  // possibly flushing what is below fp32's resolution of the sums anyway), so no rescale ever multiplies by an infinity
  int unit_e = -100000;
  for (int ti = 0; ti < tpw; ++ti) {
    const int tile = blockIdx.x * tpw + ti;
    if (tile >= ntiles) break;
    const _Float16* gtile = a.gt + ((long long)b * ntiles + tile) * (2 * 64 * DT_THREADS) + 64 * wave;
    const float* ctile = a.col + (((long long)b * a.dg + d) * ntiles + tile) * (DT_COLROWS * DT_THREADS) + 64 * wave;
    // this lane's column values of the tile: rows r and 32 + r (the latter only for r < 4), 4 k-steps x 8 pixels
    f32x4 cv[2][4][2];
    float cmax = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int et = 32 * nt + r;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        if (et < DT_COLROWS) {
          v0 = *reinterpret_cast<const f32x4*>(ctile + et * DT_THREADS + 16 * k + 8 * h);
          v1 = *reinterpret_cast<const f32x4*>(ctile + et * DT_THREADS + 16 * k + 8 * h + 4);
        }
        cv[nt][k][0] = v0; cv[nt][k][1] = v1;
#pragma unroll
        for (int j = 0; j < 4; ++j) cmax = fmaxf(cmax, fmaxf(fabsf(v0[j]), fabsf(v1[j])));
      }
    float cinv;
    float csc = dcnb_pow2_scale(wave_max(cmax), cinv);
    const int ue = ilogbf(a.gt_inv[(long long)b * ntiles + tile]) + ilogbf(cinv);     // both exact, normal powers of two
    if (ue > unit_e) {
      const int de = unit_e - ue;
      const float f = ldexpf(1.f, de < -200 ? -200 : de);                             // <= 1; 0 for the first tile (sums are 0)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) gacc[i >> 1][i & 1][q] *= f;
      unit_e = ue;
    } else if (ue < unit_e) {
      const int de = ue - unit_e;
      csc *= ldexpf(1.f, de < -200 ? -200 : de);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int l0 = 16 * k + 8 * h;
      dcnb_f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        ah[mt] = *reinterpret_cast<const dcnb_f16x8*>(gtile + (32 * mt + r) * DT_THREADS + l0);
        al[mt] = *reinterpret_cast<const dcnb_f16x8*>(gtile + (64 + 32 * mt + r) * DT_THREADS + l0);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) dcnb_split8(cv[nt][k][0], cv[nt][k][1], csc, bh[nt], bl[nt]);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          gacc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], gacc[mt][nt], 0, 0, 0);
          gacc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], gacc[mt][nt], 0, 0, 0);
          gacc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], gacc[mt][nt], 0, 0, 0);
        }
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int et = 32 * nt + r;
      if (et < DT_COLROWS)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int o = 32 * mt + (q & 3) + 8 * (q >> 2) + 4 * h;
          red[(wave * 64 + o) * DT_COLROWS + et] = ldexpf(gacc[mt][nt][q], unit_e < -1000 ? 0 : unit_e);
        }
    }
  __syncthreads();
  for (int i = tid; i < 64 * DT_COLROWS; i += DT_THREADS) {
    const int o = i / DT_COLROWS, et = i - o * DT_COLROWS, e = et / DT_MAXT, t = et - e * DT_MAXT;
    if (o >= a.Co || t >= T) continue;
    const float v = (red[i] + red[64 * DT_COLROWS + i]) + (red[2 * 64 * DT_COLROWS + i] + red[3 * 64 * DT_COLROWS + i]);
    unsafeAtomicAdd(a.gw + ((long long)o * a.C + 4 * d + e) * T + t, a.scale * v);
  }
}

constexpr int WPX = 64;     // positions per chunk
constexpr int WOB = 64;     // output channels per workgroup
constexpr int WPAIRS = 16;  // (cout, tap) pairs per thread  =>  kh*kw <= 64

__global__ __launch_bounds__(256) void dcn_bwd_weight_kernel(DcnBwdArgs a, int span /* positions per workgroup */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* gl = smem;                      // [WOB][WPX + 1]
  float* vl = smem + WOB * (WPX + 1);    // [T][WPX]
  const int tid = threadIdx.x;
  const int T = a.kh * a.kw, P = a.Ho * a.Wo;
  const int Cg = a.C / a.groups, Cog = a.Co / a.groups, Cdg = a.C / a.dg;
  const int c = blockIdx.y, g = c / Cg, cl = c - g * Cg, d = c / Cdg;
  const int o0 = blockIdx.z * WOB, no = (Cog - o0) < WOB ? (Cog - o0) : WOB;
  const long long total = (long long)a.B * P;
  const long long q0 = (long long)blockIdx.x * span;
  const long long q1 = (q0 + span) < total ? (q0 + span) : total;
  float acc[WPAIRS];
#pragma unroll
  for (int k = 0; k < WPAIRS; ++k) acc[k] = 0.f;

  for (long long qc = q0; qc < q1; qc += WPX) {
    __syncthreads();
    for (int item = tid; item < T * WPX; item += 256) {
      const int t = item >> 6, px = item & (WPX - 1);
      const long long q = qc + px;
      float val = 0.f;
      if (q < q1) {
        const int b = (int)(q / P), p = (int)(q - (long long)b * P);
        const int ho = p / a.Wo, wo = p - ho * a.Wo, ki = t / a.kw, kj = t - ki * a.kw;
        const long long ob = ((long long)(b * a.dg + d) * T + t) * 2 * P + p;
        const float h_im = (float)(ho * a.sh - a.ph + ki * a.dh) + a.offset[ob];
        const float w_im = (float)(wo * a.sw - a.pw + kj * a.dw) + a.offset[ob + P];
        const Sample s = make_sample(h_im, w_im, a.H, a.W);
        if (s.valid) {
          const float* im = a.in + ((long long)b * a.C + c) * a.H * a.W;
          const float v1 = s.o1 >= 0 ? im[s.o1] : 0.f, v2 = s.o2 >= 0 ? im[s.o2] : 0.f;
          const float v3 = s.o3 >= 0 ? im[s.o3] : 0.f, v4 = s.o4 >= 0 ? im[s.o4] : 0.f;
          val = (s.hh * s.hw) * v1 + (s.hh * s.lw) * v2 + (s.lh * s.hw) * v3 + (s.lh * s.lw) * v4;
          if (a.mask) val *= a.mask[((long long)(b * a.dg + d) * T + t) * P + p];
        }
      }
      vl[t * WPX + px] = val;
    }
    for (int item = tid; item < WOB * WPX; item += 256) {
      const int o = item >> 6, px = item & (WPX - 1);
      const long long q = qc + px;
      float v = 0.f;
      if (o < no && q < q1) {
        const int b = (int)(q / P), p = (int)(q - (long long)b * P);
        v = a.gout[((long long)b * a.Co + g * Cog + o0 + o) * P + p];
      }
      gl[o * (WPX + 1) + px] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < WPAIRS; ++k) {
      const int pair = tid + k * 256;
      if (pair < WOB * T) {
        const int o = pair & (WOB - 1), t = pair >> 6;
        const float* gr = gl + o * (WPX + 1);
        const float* vr = vl + t * WPX;
        float s = 0.f;
#pragma unroll 16
        for (int px = 0; px < WPX; ++px) s = fmaf(gr[px], vr[px], s);
        acc[k] += s;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < WPAIRS; ++k) {
    const int pair = tid + k * 256;
    if (pair < WOB * T) {
      const int o = pair & (WOB - 1), t = pair >> 6;
      if (o < no) unsafeAtomicAdd(a.gw + ((long long)(g * Cog + o0 + o) * Cg + cl) * T + t, a.scale * acc[k]);
    }
  }
}

// The same contraction for four input channels of one deformable group at a time (groups == 1, (C/dg) % 4 == 0,
// kh*kw <= 9): the sample position, its bilinear weights and the offset / mask reads are shared by the four channels and
// the grad_output tile is staged once for all of them -- a quarter of the latency-bound sampling work per channel.
constexpr int W4PAIRS = 9;   // (cout, channel, tap) triples per thread: 64 * 4 * T / 256  =>  T <= 9

__global__ __launch_bounds__(256) void dcn_bwd_weight4_kernel(DcnBwdArgs a, int span /* positions per workgroup */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* gl = smem;                      // [WOB][WPX + 1]
  float* vl = smem + WOB * (WPX + 1);    // [4][T][WPX]
  const int tid = threadIdx.x;
  const int T = a.kh * a.kw, P = a.Ho * a.Wo, Cdg = a.C / a.dg;
  const int c0 = blockIdx.y * 4, d = c0 / Cdg;
  const int o0 = blockIdx.z * WOB, no = (a.Co - o0) < WOB ? (a.Co - o0) : WOB;
  const long long total = (long long)a.B * P;
  const long long q0 = (long long)blockIdx.x * span;
  const long long q1 = (q0 + span) < total ? (q0 + span) : total;
  float acc[W4PAIRS];
#pragma unroll
  for (int k = 0; k < W4PAIRS; ++k) acc[k] = 0.f;

  for (long long qc = q0; qc < q1; qc += WPX) {
    __syncthreads();
    for (int item = tid; item < T * WPX; item += 256) {
      const int t = item >> 6, px = item & (WPX - 1);
      const long long q = qc + px;
      float val[4] = {0.f, 0.f, 0.f, 0.f};
      if (q < q1) {
        const int b = (int)(q / P), p = (int)(q - (long long)b * P);
        const int ho = p / a.Wo, wo = p - ho * a.Wo, ki = t / a.kw, kj = t - ki * a.kw;
        const long long ob = ((long long)(b * a.dg + d) * T + t) * 2 * P + p;
        const float h_im = (float)(ho * a.sh - a.ph + ki * a.dh) + a.offset[ob];
        const float w_im = (float)(wo * a.sw - a.pw + kj * a.dw) + a.offset[ob + P];
        const Sample s = make_sample(h_im, w_im, a.H, a.W);
        if (s.valid) {
          const float m = a.mask ? a.mask[((long long)(b * a.dg + d) * T + t) * P + p] : 1.f;
          const float w1 = s.hh * s.hw, w2 = s.hh * s.lw, w3 = s.lh * s.hw, w4 = s.lh * s.lw;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float* im = a.in + ((long long)b * a.C + c0 + e) * a.H * a.W;
            const float v1 = s.o1 >= 0 ? im[s.o1] : 0.f, v2 = s.o2 >= 0 ? im[s.o2] : 0.f;
            const float v3 = s.o3 >= 0 ? im[s.o3] : 0.f, v4 = s.o4 >= 0 ? im[s.o4] : 0.f;
            val[e] = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * m;
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) vl[(e * T + t) * WPX + px] = val[e];
    }
    for (int item = tid; item < WOB * WPX; item += 256) {
      const int o = item >> 6, px = item & (WPX - 1);
      const long long q = qc + px;
      float v = 0.f;
      if (o < no && q < q1) {
        const int b = (int)(q / P), p = (int)(q - (long long)b * P);
        v = a.gout[((long long)b * a.Co + o0 + o) * P + p];
      }
      gl[o * (WPX + 1) + px] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < W4PAIRS; ++k) {
      const int pair = tid + k * 256;
      if (pair < WOB * 4 * T) {
        const int o = pair & (WOB - 1), et = pair >> 6;          // et = e * T + t
        const float* gr = gl + o * (WPX + 1);
        const float* vr = vl + et * WPX;
        float s = 0.f;
#pragma unroll 16
        for (int px = 0; px < WPX; ++px) s = fmaf(gr[px], vr[px], s);
        acc[k] += s;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < W4PAIRS; ++k) {
    const int pair = tid + k * 256;
    if (pair < WOB * 4 * T) {
      const int o = pair & (WOB - 1), et = pair >> 6;
      if (o < no) unsafeAtomicAdd(a.gw + ((long long)(o0 + o) * a.C + c0) * T + et, a.scale * acc[k]);   // [o][c0 + e][t]
    }
  }
}

// grid = (Co, slices): every workgroup sums one slice of one output channel's B*P values; one atomic per workgroup
__global__ __launch_bounds__(256) void dcn_bwd_bias_kernel(const float* __restrict__ gout, float* __restrict__ gbias, int B,
                                                           int Co, int P) {
  __shared__ float part[4];
  const int o = blockIdx.x, tid = threadIdx.x;
  const long long total = (long long)B * P;
  float s = 0.f;
  for (long long q = (long long)blockIdx.y * 256 + tid; q < total; q += (long long)gridDim.y * 256) {
    const int b = (int)(q / P), p = (int)(q - (long long)b * P);
    s += gout[((long long)b * Co + o) * P + p];
  }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh, 64);
  if ((tid & 63) == 0) part[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) unsafeAtomicAdd(gbias + o, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace

static long long dcnb_gt_bytes(long long nt) { return nt * (2 * 64 * DT_THREADS) * (long long)sizeof(_Float16); }
static long long dcnb_inv_bytes(long long nt) { return ((nt * 4 + 255) / 256) * 256; }

// Workspace of cdfo_dcn_backward_ws: room for the per-tile transposed fp16 copy of grad_output (+ one float per tile) and
// the column values (4 * 9 floats per pixel and deformable group) when the fused weight-gradient form applies (groups == 1, C/dg == 4, kh*kw <= 9, Co <= 64), else 0.  -1 on bad shapes.
extern "C" long long cdfo_dcn_backward_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                                                       int ph, int pw, int dh, int dw, int groups, int deformable_groups) {
  if (B <= 0 || C <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || groups <= 0 ||
      deformable_groups <= 0 || C % groups || Co % groups || C % deformable_groups)
    return -1;
  const int Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1, Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (Ho <= 0 || Wo <= 0) return -1;
  if (groups != 1 || C / deformable_groups != 4 || kh * kw > DT_MAXT || Co > 64) return 0;
  const long long nt = (long long)B * cdiv(Wo, DT_X) * cdiv(Ho, DT_Y);
  return dcnb_gt_bytes(nt) + dcnb_inv_bytes(nt) + nt * deformable_groups * (DT_COLROWS * DT_THREADS) * (long long)sizeof(float);
}

extern "C" int cdfo_dcn_backward_ws(const float* in, const float* offset, const float* mask, const float* weight,
                                    const float* grad_out, float* grad_in, float* grad_offset, float* grad_mask,
                                    float* grad_weight, float* grad_bias, int B, int C, int H, int W, int Co, int kh, int kw,
                                    int sh, int sw, int ph, int pw, int dh, int dw, int groups, int deformable_groups,
                                    float scale, void* workspace, long long workspace_bytes, void* stream);

extern "C" int cdfo_dcn_backward(const float* in, const float* offset, const float* mask, const float* weight,
                                 const float* grad_out, float* grad_in, float* grad_offset, float* grad_mask,
                                 float* grad_weight, float* grad_bias, int B, int C, int H, int W, int Co, int kh, int kw,
                                 int sh, int sw, int ph, int pw, int dh, int dw, int groups, int deformable_groups,
                                 float scale, void* stream) {
  return cdfo_dcn_backward_ws(in, offset, mask, weight, grad_out, grad_in, grad_offset, grad_mask, grad_weight, grad_bias, B, C,
                              H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, deformable_groups, scale, nullptr, 0, stream);
}

// The same with a workspace (cdfo_dcn_backward_workspace_bytes, 16-byte aligned; NULL / too small: the weight gradient
// runs as its own kernel with a second sampling pass -- same results up to summation order, slower).
extern "C" int cdfo_dcn_backward_ws(const float* in, const float* offset, const float* mask, const float* weight,
                                    const float* grad_out, float* grad_in, float* grad_offset, float* grad_mask,
                                    float* grad_weight, float* grad_bias, int B, int C, int H, int W, int Co, int kh, int kw,
                                    int sh, int sw, int ph, int pw, int dh, int dw, int groups, int deformable_groups,
                                    float scale, void* workspace, long long workspace_bytes, void* stream) {
  if (B <= 0 || C <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || groups <= 0 ||
      deformable_groups <= 0)
    return CDFO_EINVAL;
  if (C % groups || Co % groups || C % deformable_groups) return CDFO_EINVAL;
  if (!in || !offset || !weight || !grad_out) return CDFO_EINVAL;
  if (grad_mask && !mask) return CDFO_EINVAL;
  const int Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  const int Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (Ho <= 0 || Wo <= 0) return CDFO_EINVAL;
  const int T = kh * kw, P = Ho * Wo;
  if (T > (WPAIRS * 256) / WOB) return CDFO_EINVAL;
  if ((long long)deformable_groups * T > 65535 || B > 65535 || C > 65535) return CDFO_EINVAL;
  DcnBwdArgs a{in, offset, mask, weight, grad_out, grad_in, grad_offset, grad_mask, grad_weight, grad_bias,
               B, C, H, W, Co, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, groups, deformable_groups, scale, nullptr, nullptr, nullptr};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double px = (double)B * P;
  bool gw_done = false;
  if (grad_in || grad_offset || grad_mask) {
    CdfoProfScope prof(st, KID_DCN_BWD, 2.0 * px * Co * (C / groups) * T,
                       4.0 * (px * (Co + 6.0 * deformable_groups * T) + 2.0 * B * C * H * W + (double)Co * (C / groups) * T));
    const int WH = (DT_Y - 1) * sh + (kh - 1) * dh + 2 + 2 * DT_MARGIN, WW = (DT_X - 1) * sw + (kw - 1) * dw + 2 + 2 * DT_MARGIN;
    const size_t wlds = (size_t)4 * WH * WW * sizeof(long long);
    const int tiles_x = cdiv(Wo, DT_X), tiles_y = cdiv(Ho, DT_Y);
    // GW form: the data kernel also stores its column values, dcn_bwd_gw_kernel contracts them with grad_output
    const long long ws_need = cdfo_dcn_backward_workspace_bytes(B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups,
                                                                deformable_groups);
    const bool fuse_gw = grad_weight != nullptr && Co <= 64 && ws_need > 0 && workspace != nullptr &&
                         workspace_bytes >= ws_need && aligned16(workspace);
    if (groups == 1 && C / deformable_groups == 4 && T <= DT_MAXT && wlds <= 48 * 1024 &&
        (long long)tiles_x * tiles_y < (1ll << 31)) {
      const int ntiles = tiles_x * tiles_y;
      if (fuse_gw) {
        const long long nt = (long long)B * ntiles;
        char* wsp = static_cast<char*>(workspace);
        _Float16* gt = reinterpret_cast<_Float16*>(wsp);
        float* gt_inv = reinterpret_cast<float*>(wsp + dcnb_gt_bytes(nt));
        a.col = reinterpret_cast<float*>(wsp + dcnb_gt_bytes(nt) + dcnb_inv_bytes(nt));
        hipLaunchKernelGGL(dcn_bwd_gprep_kernel, dim3(ntiles, B), dim3(DT_THREADS), 0, st, grad_out, gt, gt_inv, Co, Ho, Wo,
                           tiles_x, ntiles);
        a.gt = gt; a.gt_inv = gt_inv;
      }
      dim3 grid(ntiles, deformable_groups, B);
      if (T == 9) {
        if (fuse_gw) hipLaunchKernelGGL((dcn_bwd_data_tile_kernel<true, true>), grid, dim3(DT_THREADS), wlds, st, a, WH, WW, tiles_x, ntiles);
        else hipLaunchKernelGGL((dcn_bwd_data_tile_kernel<true, false>), grid, dim3(DT_THREADS), wlds, st, a, WH, WW, tiles_x, ntiles);
      } else {
        if (fuse_gw) hipLaunchKernelGGL((dcn_bwd_data_tile_kernel<false, true>), grid, dim3(DT_THREADS), wlds, st, a, WH, WW, tiles_x, ntiles);
        else hipLaunchKernelGGL((dcn_bwd_data_tile_kernel<false, false>), grid, dim3(DT_THREADS), wlds, st, a, WH, WW, tiles_x, ntiles);
      }
      if (fuse_gw) {
        CDFO_LAUNCH_CHECK();
        // a workgroup walks a run of tiles and keeps the sums in registers (one set of atomics per run)
        int tpw = 16;
        while (tpw > 1 && (long long)cdiv(ntiles, tpw) * deformable_groups * B < 4096) tpw >>= 1;
        hipLaunchKernelGGL(dcn_bwd_gw_kernel, dim3(cdiv(ntiles, tpw), deformable_groups, B), dim3(DT_THREADS), 0, st, a, ntiles, tpw);
        gw_done = true;
      }
    } else {
      hipLaunchKernelGGL(dcn_bwd_data_kernel, dim3(cdiv(P, 256), deformable_groups * T, B), dim3(256), 0, st, a);
    }
    CDFO_LAUNCH_CHECK();
  }
  if (grad_weight && !gw_done) {
    const long long total = (long long)B * P;
    // enough workgroups to fill 256 CUs a few times over, each walking a whole number of 64-position chunks
    const int zb = cdiv(Co / groups, WOB);
    long long want = 2048 / ((long long)C * zb) + 1;
    long long span = (total + want - 1) / want;
    span = (span + WPX - 1) / WPX * WPX;
    const int nx = (int)((total + span - 1) / span);
    CdfoProfScope prof(st, KID_DCN_BWD, 2.0 * px * Co * (C / groups) * T,
                       4.0 * (px * (Co + 3.0 * deformable_groups * T) + (double)B * C * H * W + (double)Co * (C / groups) * T));
    if (groups == 1 && (C / deformable_groups) % 4 == 0 && T <= W4PAIRS) {
      const size_t lds4 = (size_t)(WOB * (WPX + 1) + 4 * T * WPX) * sizeof(float);
      long long want4 = 2048 / ((long long)(C / 4) * zb) + 1;
      long long span4 = (total + want4 - 1) / want4;
      span4 = (span4 + WPX - 1) / WPX * WPX;
      const int nx4 = (int)((total + span4 - 1) / span4);
      hipLaunchKernelGGL(dcn_bwd_weight4_kernel, dim3(nx4, C / 4, zb), dim3(256), lds4, st, a, (int)span4);
    } else {
      const size_t lds = (size_t)(WOB * (WPX + 1) + T * WPX) * sizeof(float);
      hipLaunchKernelGGL(dcn_bwd_weight_kernel, dim3(nx, C, zb), dim3(256), lds, st, a, (int)span);
    }
    CDFO_LAUNCH_CHECK();
  }
  if (grad_bias) {
    const long long slices = ((long long)B * P + 16383) / 16384;
    hipLaunchKernelGGL(dcn_bwd_bias_kernel, dim3(Co, (unsigned)(slices < 64 ? slices : 64)), dim3(256), 0, st, grad_out, grad_bias,
                       B, Co, P);
    CDFO_LAUNCH_CHECK();
  }
  return 0;
}
