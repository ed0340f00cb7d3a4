// Shared epilogue of the MFMA convolution kernels: accumulator tiles -> wave-private LDS transpose -> 16-byte rows.
//
// A 32x32 MFMA accumulator holds ONE output channel per lane (column) and 16 pixels in its registers, so storing it
// directly means 4-byte scattered stores (256 B per wave-instruction) -- store-issue bound.  Each wave instead parks a
// 32-pixel x (32*NT)-channel tile in its own LDS slice and reads it back as float4 along the channel axis: every
// global store / residual load then moves 1 KiB per wave-instruction (4 whole pixels when the pitch is 64 channels).
#pragma once
#include "common.h"

template <int NT>
struct ConvEpi {
  static constexpr int RS = 32 * NT + 4;          // floats per staged pixel row (+4: keeps b128 reads conflict-free)
  static constexpr int WAVE_FLOATS = 32 * RS;     // per-wave LDS slice
  static constexpr int BLOCK_BYTES = 4 * WAVE_FLOATS * 4;
};

// acc[ni][e]: row(pixel) = (e&3) + 8*(e>>2) + 4*(lane>>5), col(channel) = ni*32 + (lane&31).
// Pixel i of the tile sits at (oy_base + (i>>4), ox0 + (i&15)).  Epilogue: +bias -> act -> +res1 -> +res2 -> store.
template <int NT, int LOG2_ROW>   // the tile's 32 pixels are 32 >> LOG2_ROW rows of (1 << LOG2_ROW) pixels
__device__ __forceinline__ void conv_tile_epilogue_impl(const cdfo_conv_args& a, float* wl, const f32x16* acc, int lane,
                                                        int b, int oy_base, int ox0, int n0) {
  constexpr int RS = ConvEpi<NT>::RS;
  const int h = lane >> 5, r = lane & 31;
#pragma unroll
  for (int ni = 0; ni < NT; ++ni)
#pragma unroll
    for (int e = 0; e < 16; ++e) wl[((e & 3) + 8 * (e >> 2) + 4 * h) * RS + ni * 32 + r] = acc[ni][e];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr int C4 = 8 * NT, RPI = 64 / C4;       // float4 columns per row, rows per wave-instruction
  constexpr int ROWW = 1 << LOG2_ROW, NROWS = 32 >> LOG2_ROW;
  const int c4 = lane % C4, pr = lane / C4;
  const int n = n0 + c4 * 4;
  const bool nok = n < a.Cout;                    // Cout % 4 == 0 is enforced by the launcher
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (a.bias && nok) bias = *reinterpret_cast<const f32x4*>(a.bias + n);
  // activation as one branch-free form: v > 0 ? v : slope * v   (none: 1, LeakyReLU: 0.1, ReLU: 0)
  const float slope = a.act == CDFO_ACT_NONE ? 1.f : (a.act == CDFO_ACT_LRELU ? 0.1f : 0.f);
  const bool plain = a.store_mode == CDFO_STORE_PLAIN;
  const int cq = a.Cout >> 2, sub = a.store_mode == CDFO_STORE_SHUFFLE2 ? n / cq : 0, cc = n - sub * cq;
#pragma unroll
  for (int row = 0; row < NROWS; ++row) {
    const int oy = oy_base + row;
    if (oy >= a.Ho) break;
    // 64-bit base once per image row of the tile; 32-bit steps inside it
    const long long pixrow = (long long)(b * a.Ho + oy) * a.Wo + ox0;
    // element offset of this lane's first value in the row (64-bit once per image row of the tile; 32-bit steps inside)
    long long obase;
    if (plain) obase = pixrow * a.ldo + n;
    else if (a.store_mode == CDFO_STORE_SHUFFLE2)
      obase = ((long long)(b * 2 * a.Ho + 2 * oy + (sub >> 1)) * (2 * a.Wo) + 2 * ox0 + (sub & 1)) * a.ldo + cc;
    else  // space-to-depth: pixel (oy>>1, x>>1), phase (oy&1, x&1); ox0 is even, so x&1 == xi&1
      obase = ((long long)(b * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) + (ox0 >> 1)) * a.ldo + (oy & 1) * 2 * a.Cout + n;
    float* orow = a.out + obase;
    _Float16* orow16 = reinterpret_cast<_Float16*>(a.out) + obase;   // out_f16: same indexing, 2-byte elements
    const float* r1row = a.res1 ? a.res1 + pixrow * a.ldr1 + n : nullptr;
    const float* r2row = a.res2 ? a.res2 + pixrow * a.ldr2 + n : nullptr;
    const bool s2d = a.store_mode == CDFO_STORE_S2D;
    const int ostep = plain ? a.ldo : 2 * a.ldo;
#pragma unroll
    for (int it = 0; it < ROWW / RPI; ++it) {
      const int xi = it * RPI + pr;               // pixel inside this row of the tile
      f32x4 v = *reinterpret_cast<const f32x4*>(wl + (row * ROWW + xi) * RS + c4 * 4) + bias;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f) + slope * fminf(v[k], 0.f);
      if (!nok || ox0 + xi >= a.Wo) continue;
      if (r1row) v += *reinterpret_cast<const f32x4*>(r1row + xi * a.ldr1);
      if (r2row) v += *reinterpret_cast<const f32x4*>(r2row + xi * a.ldr2);
      const int eoff = s2d ? (xi >> 1) * a.ldo + (xi & 1) * a.Cout : xi * ostep;
      if (a.out_f16) {
        typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
        f16x4_t hv;
#pragma unroll
        for (int k = 0; k < 4; ++k) hv[k] = (_Float16)v[k];
        *reinterpret_cast<f16x4_t*>(orow16 + eoff) = hv;
      } else {
        *reinterpret_cast<f32x4*>(orow + eoff) = v;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int NT>
__device__ __forceinline__ void conv_tile_epilogue(const cdfo_conv_args& a, float* wl, const f32x16* acc, int lane,
                                                   int b, int oy_base, int ox0, int n0) {
  conv_tile_epilogue_impl<NT, 4>(a, wl, acc, lane, b, oy_base, ox0, n0);     // 2 rows x 16 pixels
}
template <int NT>
__device__ __forceinline__ void conv_tile_epilogue_row32(const cdfo_conv_args& a, float* wl, const f32x16* acc, int lane,
                                                         int b, int oy, int ox0, int n0) {
  conv_tile_epilogue_impl<NT, 5>(a, wl, acc, lane, b, oy, ox0, n0);          // 1 row x 32 pixels
}
