// HBM-bound pixel-major kernels of the CVSR_V8 path: prior/pixel stems, LayerNorm, depthwise 3x3, motion-vector
// warp, 2x2 mean / bilinear x2 resampling, channel gating, the final 64->1 conv fused with the bilinear x4 skip.
// One pixel's 64 channels are a contiguous 256-B run, so 16 consecutive lanes (a float4 each) cover a pixel and
// every wave-instruction moves 4 whole pixels = 1 KiB, fully coalesced.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------
// 3x3 conv, 1 input channel -> 64 output channels (+bias, +act) [+ second output = result + add]
// replaces conv_first / conv_second / conv_expand_ufs / conv_expand_rms (arch/SIDECVSR_our.py:4379-4384,4417-4418,
// 4446-4449 incl. `fea_com = fea_i + rms_prior`).
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ img, long long img_bstride,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        int B, int H, int W, int act, float* __restrict__ out, int ldo,
                                                        const float* __restrict__ add, int lda, float* __restrict__ out2,
                                                        int ldo2) {
  const int cg = threadIdx.x & 15;  // channel group: couts 4cg..4cg+3
  float wr[9][4], br[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    br[j] = bias ? bias[cg * 4 + j] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t][j] = w[(cg * 4 + j) * 9 + t];
  }
  const long long npix = (long long)B * H * W;
  for (long long p = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 4; p < npix;
       p += ((long long)gridDim.x * blockDim.x) >> 4) {
    const int x = p % W;
    const int y = (p / W) % H;
    const int b = p / ((long long)W * H);
    const float* ip = img + b * img_bstride;
    float v[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int yy = y + dy - 1, xx = x + dx - 1;
        v[dy * 3 + dx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? ip[(long long)yy * W + xx] : 0.f;
      }
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = br[j];
#pragma unroll
      for (int t = 0; t < 9; ++t) s += v[t] * wr[t][j];
      o[j] = act_apply(s, act);
    }
    *reinterpret_cast<f32x4*>(out + p * ldo + cg * 4) = o;
    if (out2) {
      const f32x4 ad = *reinterpret_cast<const f32x4*>(add + p * lda + cg * 4);
      *reinterpret_cast<f32x4*>(out2 + p * ldo2 + cg * 4) = o + ad;
    }
  }
}

// Two 3x3 convolutions 1 -> 64 of the SAME single-channel plane in one pass:
//     outA = conv(img; wA, bA) + add            (fea_com = fea_i + conv_expand_rms(rms), arch.py:4446-4449)
//     outB = actB(conv(img; wB, bB))            (relu(conv_du_re.0(conv_expand_rms(rms))): the 1x1 conv_du_re.0 of
//                                                LLongRangAttention, arch.py:2148-2152 / 2200, composed with
//                                                conv_expand_rms on the host -- a 1x1 after a 3x3, nothing in between)
// so that `rms_prior` itself (read only by those two consumers) is never written.
__global__ __launch_bounds__(256) void stem_conv2_kernel(const float* __restrict__ img, long long img_bstride,
                                                         const float* __restrict__ wA, const float* __restrict__ bA,
                                                         const float* __restrict__ add, int lda, float* __restrict__ outA,
                                                         int ldoA, const float* __restrict__ wB,
                                                         const float* __restrict__ bB, int actB, float* __restrict__ outB,
                                                         int ldoB, int B, int H, int W, int s2dB) {
  const int cg = threadIdx.x & 15;  // channel group: couts 4cg..4cg+3
  float wa[9][4], wb[9][4], ba[4], bb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ba[j] = bA ? bA[cg * 4 + j] : 0.f;
    bb[j] = bB ? bB[cg * 4 + j] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) { wa[t][j] = wA[(cg * 4 + j) * 9 + t]; wb[t][j] = wB[(cg * 4 + j) * 9 + t]; }
  }
  const long long npix = (long long)B * H * W;
  for (long long p = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 4; p < npix;
       p += ((long long)gridDim.x * blockDim.x) >> 4) {
    const int x = p % W;
    const int y = (p / W) % H;
    const int b = p / ((long long)W * H);
    const float* ip = img + b * img_bstride;
    float v[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int yy = y + dy - 1, xx = x + dx - 1;
        v[dy * 3 + dx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? ip[(long long)yy * W + xx] : 0.f;
      }
    f32x4 oa, ob;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float sa = ba[j], sb = bb[j];
#pragma unroll
      for (int t = 0; t < 9; ++t) { sa += v[t] * wa[t][j]; sb += v[t] * wb[t][j]; }
      oa[j] = sa;
      ob[j] = act_apply(sb, actB);
    }
    *reinterpret_cast<f32x4*>(outA + p * ldoA + cg * 4) = oa + *reinterpret_cast<const f32x4*>(add + p * lda + cg * 4);
    if (s2dB) {
      // space-to-depth form [B][H/2 + 1][W/2 + 1][4 * 64] (channel = (y & 1, x & 1) phase * 64 + c; the extra last row and
      // column are the caller's zero padding): the 3x3 stride-2 pad-2 convolution that follows (conv_du_re.2) is then a
      // stride-1 convolution with a 2x2 tap window over it
      const long long q = ((long long)b * ((H >> 1) + 1) + (y >> 1)) * ((W >> 1) + 1) + (x >> 1);
      *reinterpret_cast<f32x4*>(outB + q * ldoB + ((y & 1) * 2 + (x & 1)) * 64 + cg * 4) = ob;
    } else {
      *reinterpret_cast<f32x4*>(outB + p * ldoB + cg * 4) = ob;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// per-pixel LayerNorm over 64 channels, biased variance, eps 1e-5 (arch.py:1169-1185)
__global__ __launch_bounds__(256) void layernorm64_kernel(const float* __restrict__ in, int ldi,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, long long npix,
                                                          float* __restrict__ out, int ldo) {
  const int cg = threadIdx.x & 15;
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + cg * 4);
  const f32x4 be = *reinterpret_cast<const f32x4*>(beta + cg * 4);
  for (long long p = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 4; p < npix;
       p += ((long long)gridDim.x * blockDim.x) >> 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(in + p * ldi + cg * 4);
    float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mu = s * (1.f / 64.f);
    const f32x4 d = v - mu;
    float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.f / sqrtf(q * (1.f / 64.f) + 1e-5f);
    *reinterpret_cast<f32x4*>(out + p * ldo + cg * 4) = d * rstd * g + be;
  }
}

// the same, written as fp16 hi | lo planes in chunk-planar layout [B][8][P][16] (split-fp16 source of the ring kernel)
__global__ __launch_bounds__(256) void layernorm64_cp16hl_kernel(const float* __restrict__ in, int ldi,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, long long npix, long long P,
                                                                 _Float16* __restrict__ out) {
  typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
  const int cg = threadIdx.x & 15;
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + cg * 4);
  const f32x4 be = *reinterpret_cast<const f32x4*>(beta + cg * 4);
  for (long long p = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 4; p < npix;
       p += ((long long)gridDim.x * blockDim.x) >> 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(in + p * ldi + cg * 4);
    float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mu = s * (1.f / 64.f);
    const f32x4 d = v - mu;
    float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.f / sqrtf(q * (1.f / 64.f) + 1e-5f);
    const f32x4 y = d * rstd * g + be;
    f16x4_t hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) { hi[k] = (_Float16)y[k]; lo[k] = (_Float16)(y[k] - (float)hi[k]); }
    const long long b = p / P, pix = p - b * P;
    _Float16* o16 = out + ((b * 8 + (cg >> 2)) * P + pix) * 16 + (cg & 3) * 4;
    *reinterpret_cast<f16x4_t*>(o16) = hi;
    *reinterpret_cast<f16x4_t*>(o16 + 4 * P * 16) = lo;
  }
}

// ------------------------------------------------------------------------------------------------------------
// depthwise 3x3, pad 1, no bias (qkv_dwconv, arch.py:1552,1559). weights raw [C][1][3][3]; C % 4 == 0, C <= 256.
// One thread = 4 consecutive pixels of a row x 4 channels: an 18-load 3x6 window serves 4 outputs (4.5 loads per
// output instead of 9), the 36 weights of the channel group stay in registers across the grid-stride loop.
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ in, int ldi,
                                                        const float* __restrict__ w, int B, int H, int W, int C,
                                                        float* __restrict__ out, int ldo) {
  const int cgs = C >> 2;
  const int xq = (W + 3) >> 2;                          // 4-pixel groups per row
  const long long total = (long long)B * H * xq * cgs;
  const long long stride = (long long)gridDim.x * blockDim.x;   // launcher makes this a multiple of cgs
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int cg = i % cgs;
  f32x4 wr[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[t][j] = w[(cg * 4 + j) * 9 + t];
  for (; i < total; i += stride) {
    const long long g = i / cgs;
    const int x0 = (int)(g % xq) * 4;
    const int y = (g / xq) % H;
    const long long b = g / ((long long)xq * H);
    f32x4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y + dy - 1;
      if (yy < 0 || yy >= H) continue;
      const float* row = in + ((b * H + yy) * W) * ldi + cg * 4;
      f32x4 v[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int xx = x0 + k - 1;
        v[k] = (xx >= 0 && xx < W) ? *reinterpret_cast<const f32x4*>(row + (long long)xx * ldi) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) acc[k] += v[k + dx] * wr[dy * 3 + dx];
    }
    float* o = out + ((b * H + y) * W + x0) * ldo + cg * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (x0 + k < W) *reinterpret_cast<f32x4*>(o + (long long)k * ldo) = acc[k];
  }
}

// ------------------------------------------------------------------------------------------------------------
// flow_warp (arch.py:3068-3099): bilinear sample at (x + mv_x, y + mv_y), zeros outside, align_corners=True
// with the same normalise / un-normalise arithmetic as F.grid_sample.  mv: [B][2][H][W] planes.
__global__ __launch_bounds__(256) void flow_warp_kernel(const float* __restrict__ in, int ldi,
                                                        const float* __restrict__ mv, long long mv_bstride, int B,
                                                        int H, int W, int C, float* __restrict__ out, int ldo) {
  const int cgs = C >> 2;
  const long long total = (long long)B * H * W * cgs;
  const float wm = (float)(W - 1 > 1 ? W - 1 : 1), hm = (float)(H - 1 > 1 ? H - 1 : 1);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cg = i % cgs;
    const long long p = i / cgs;
    const int x = p % W;
    const int y = (p / W) % H;
    const long long b = p / ((long long)W * H);
    const float* m = mv + b * mv_bstride;
    const float fx = m[(long long)y * W + x], fy = m[(long long)(H + y) * W + x];
    const float nx = 2.0f * ((float)x + fx) / wm - 1.0f;
    const float ny = 2.0f * ((float)y + fy) / hm - 1.0f;
    const float sx = ((nx + 1.f) / 2.f) * (float)(W - 1);
    const float sy = ((ny + 1.f) / 2.f) * (float)(H - 1);
    // a sample at or beyond one pixel outside the image has no corner inside it: the result is 0 (grid_sample, zeros
    // padding).  Decided in floating point BEFORE any conversion to int -- the reference's mv2mvs leaves x / 0 = inf in the
    // motion field (test_LD_22_FPS.py:106-110) and float -> int of inf / NaN is undefined
    if (!(sx > -1.f && sx < (float)W && sy > -1.f && sy < (float)H)) {
      *reinterpret_cast<f32x4*>(out + p * ldo + cg * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    const float x0f = floorf(sx), y0f = floorf(sy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float tx = sx - x0f, ty = sy - y0f;
    const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* base = in + b * H * W * ldi + cg * 4;
    const bool xa = x0 >= 0 && x0 < W, xb = x0 + 1 >= 0 && x0 + 1 < W;
    if (y0 >= 0 && y0 < H) {
      if (xa) acc += *reinterpret_cast<const f32x4*>(base + ((long long)y0 * W + x0) * ldi) * w00;
      if (xb) acc += *reinterpret_cast<const f32x4*>(base + ((long long)y0 * W + x0 + 1) * ldi) * w01;
    }
    if (y0 + 1 >= 0 && y0 + 1 < H) {
      if (xa) acc += *reinterpret_cast<const f32x4*>(base + ((long long)(y0 + 1) * W + x0) * ldi) * w10;
      if (xb) acc += *reinterpret_cast<const f32x4*>(base + ((long long)(y0 + 1) * W + x0 + 1) * ldi) * w11;
    }
    *reinterpret_cast<f32x4*>(out + p * ldo + cg * 4) = acc;
  }
}

// ------------------------------------------------------------------------------------------------------------
// bilinear x0.5 (== 2x2 mean for even sizes) and x2, align_corners=False (Interpolate, arch.py:324-333)
__global__ __launch_bounds__(256) void down2_kernel(const float* __restrict__ in, int ldi, int B, int H, int W, int C,
                                                    float* __restrict__ out, int ldo, int accumulate) {
  const int Ho = H >> 1, Wo = W >> 1, cgs = C >> 2;
  const long long total = (long long)B * Ho * Wo * cgs;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cg = i % cgs;
    const long long p = i / cgs;
    const int x = p % Wo;
    const int y = (p / Wo) % Ho;
    const long long b = p / ((long long)Wo * Ho);
    const float* s = in + ((b * H + 2 * y) * W + 2 * x) * ldi + cg * 4;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(s), a1 = *reinterpret_cast<const f32x4*>(s + ldi);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(s + (long long)W * ldi);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(s + (long long)W * ldi + ldi);
    f32x4 v = 0.5f * (0.5f * a0 + 0.5f * a1) + 0.5f * (0.5f * b0 + 0.5f * b1);
    float* o = out + p * ldo + cg * 4;
    if (accumulate) v += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

__global__ __launch_bounds__(256) void up2_kernel(const float* __restrict__ in, int ldi, int B, int H, int W, int C,
                                                  float* __restrict__ out, int ldo, int accumulate, int out_f16) {
  const int Ho = H * 2, Wo = W * 2, cgs = C >> 2;
  const long long total = (long long)B * Ho * Wo * cgs;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cg = i % cgs;
    const long long p = i / cgs;
    const int x = p % Wo;
    const int y = (p / Wo) % Ho;
    const long long b = p / ((long long)Wo * Ho);
    // src = (dst + 0.5) * 0.5 - 0.5, clamped at 0
    float sy = ((float)y + 0.5f) * 0.5f - 0.5f, sx = ((float)x + 0.5f) * 0.5f - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float* base = in + b * H * W * ldi + cg * 4;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((long long)y0 * W + x0) * ldi);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((long long)y0 * W + x1) * ldi);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((long long)y1 * W + x0) * ldi);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((long long)y1 * W + x1) * ldi);
    f32x4 v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    if (out_f16) {   // fp16 result (feeds a single-pass fp16 convolution): ldo counts halves
      typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
      f16x4_t hv;
#pragma unroll
      for (int k = 0; k < 4; ++k) hv[k] = (_Float16)v[k];
      *reinterpret_cast<f16x4_t*>(reinterpret_cast<_Float16*>(out) + p * ldo + cg * 4) = hv;
      continue;
    }
    float* o = out + p * ldo + cg * 4;
    if (accumulate) v += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

// bilinear x2 with an fp16 "chunk-planar" result [B][C/16][2H][2W][16] (the source layout of cdfo_conv3x3_c64_ws).
// A thread owns 4 channels of the 2x2 output block (2q-1..2q, 2p-1..2p), which reads exactly the 2x2 source block
// (q-1..q, p-1..p) (clamped at the image edge): one 16-byte load per output pixel instead of four.  Threads walk along
// p inside one 16-channel plane, so the two output rows are written as runs of consecutive 32-byte pixel records.
__global__ __launch_bounds__(256) void up2_cp16_kernel(const float* __restrict__ in, int ldi, int B, int H, int W, int C,
                                                       _Float16* __restrict__ out) {
  const int Ho = H * 2, Wo = W * 2, nc = C >> 4;
  const long long total = (long long)B * nc * (H + 1) * (W + 1) * 4;
  typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = i & 3;
    long long t = i >> 2;
    const int p = t % (W + 1); t /= (W + 1);
    const int q = t % (H + 1); t /= (H + 1);
    const int c = t % nc;
    const long long b = t / nc;
    const int xa = p > 0 ? p - 1 : 0, xb = p < W ? p : W - 1;
    const int ya = q > 0 ? q - 1 : 0, yb = q < H ? q : H - 1;
    const float* base = in + b * H * W * ldi + c * 16 + g * 4;
    const f32x4 vaa = *reinterpret_cast<const f32x4*>(base + ((long long)ya * W + xa) * ldi);
    const f32x4 vab = *reinterpret_cast<const f32x4*>(base + ((long long)ya * W + xb) * ldi);
    const f32x4 vba = *reinterpret_cast<const f32x4*>(base + ((long long)yb * W + xa) * ldi);
    const f32x4 vbb = *reinterpret_cast<const f32x4*>(base + ((long long)yb * W + xb) * ldi);
    _Float16* oplane = out + ((b * nc + c) * Ho) * (long long)Wo * 16 + g * 4;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int Y = 2 * q - 1 + dy;
      if (Y < 0 || Y >= Ho) continue;
      const float ly = dy ? 0.75f : 0.25f;            // odd output rows sit 1/4 past source row q-1, even ones 3/4
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int X = 2 * p - 1 + dx;
        if (X < 0 || X >= Wo) continue;
        const float lx = dx ? 0.75f : 0.25f;
        const f32x4 v = (1.f - ly) * ((1.f - lx) * vaa + lx * vab) + ly * ((1.f - lx) * vba + lx * vbb);
        f16x4_t hv;
#pragma unroll
        for (int k = 0; k < 4; ++k) hv[k] = (_Float16)v[k];
        *reinterpret_cast<f16x4_t*>(oplane + ((long long)Y * Wo + X) * 16) = hv;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// out = x * gate[b][c]   (CALayer, arch.py:2041-2043)
__global__ __launch_bounds__(256) void scale_channels_kernel(const float* __restrict__ in, int ldi,
                                                             const float* __restrict__ gate, int B, long long P, int C,
                                                             float* __restrict__ out, int ldo,
                                                             _Float16* __restrict__ out16) {
  typedef _Float16 sc_f16x4 __attribute__((ext_vector_type(4)));
  const int cgs = C >> 2;
  const long long total = (long long)B * P * cgs;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cg = i % cgs;
    const long long p = i / cgs;
    const long long b = p / P;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gate + b * C + cg * 4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(in + p * ldi + cg * 4) * g;
    *reinterpret_cast<f32x4*>(out + p * ldo + cg * 4) = v;
    if (out16) {          // fp16 chunk-planar copy [B][C/16][P][16] (the source layout of cdfo_conv3x3_c64_ws)
      sc_f16x4 hv;
#pragma unroll
      for (int k = 0; k < 4; ++k) hv[k] = (_Float16)v[k];
      *reinterpret_cast<sc_f16x4*>(out16 + ((b * (C >> 4) + (cg >> 2)) * P + (p - b * P)) * 16 + (cg & 3) * 4) = hv;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// conv_last (3x3, 64 -> 1, +bias) fused with `out += bilinear_x4(x_center)` (arch.py:4476-4480).
// The 64-channel HR feature map (4.3 GB at c3) is read ONCE: a workgroup takes a 16 x 64 pixel tile, and for every
// pixel of its 18 x 66 halo region 16 lanes (a float4 of channels each) form the nine per-tap channel sums
// t_k(p) = sum_c w[c][k] in[p][c] (DPP row reduction, no LDS traffic), lane 0 parks them in LDS; an output pixel is
// then the sum of nine parked values at its shifted positions.  (First cut: every output pixel gathered its 3x3
// neighbourhood itself, 9 x 256 B per pixel through the caches: 1.3 TB/s.)
constexpr int CL_TY = 16, CL_TX = 64, CL_HX = CL_TX + 2, CL_NP = (CL_TY + 2) * CL_HX;    // 1188 halo pixels

__device__ __forceinline__ float row16_sum(float v) {     // sum over the 16 lanes of a DPP row, result in every lane
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));  // row_ror:8
  return v;
}

__global__ __launch_bounds__(256) void conv_last_kernel(const float* __restrict__ in, int ldi,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ xc, long long xc_bstride, int B,
                                                        int Hh, int Wh, float* __restrict__ out) {
  __shared__ float sT[9][CL_NP];
  const int tid = threadIdx.x, cg = tid & 15, grp = tid >> 4;
  f32x4 wr[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[t][j] = w[(cg * 4 + j) * 9 + t];
  const float b0 = bias[0];
  const int H = Hh >> 2, W = Wh >> 2;
  const int tiles_x = (Wh + CL_TX - 1) / CL_TX, tiles_y = (Hh + CL_TY - 1) / CL_TY;
  const int ntiles = B * tiles_y * tiles_x;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int tx = t % tiles_x, t2 = t / tiles_x;
    const int ty = t2 % tiles_y, b = t2 / tiles_y;
    const int y0 = ty * CL_TY, x0 = tx * CL_TX;
    // ---- per-tap channel sums of the halo pixels (zero outside the image = the conv's zero padding)
    for (int p = grp; p < CL_NP; p += 16) {
      const int iy = p / CL_HX, ix = p - iy * CL_HX;
      const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy >= 0 && gy < Hh && gx >= 0 && gx < Wh)
        v = *reinterpret_cast<const f32x4*>(in + (((long long)b * Hh + gy) * Wh + gx) * ldi + cg * 4);
      float tk[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const f32x4 pr = v * wr[k];
        tk[k] = row16_sum((pr[0] + pr[1]) + (pr[2] + pr[3]));
      }
      if (cg == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) sT[k][p] = tk[k];
      }
    }
    __syncthreads();
    // ---- outputs: 4 per thread
    for (int o = tid; o < CL_TY * CL_TX; o += 256) {
      const int oy = o / CL_TX, ox = o - oy * CL_TX;
      const int y = y0 + oy, x = x0 + ox;
      if (y >= Hh || x >= Wh) continue;
      float s = 0.f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) s += sT[dy * 3 + dx][(oy + dy) * CL_HX + ox + dx];
      float sy = ((float)y + 0.5f) * 0.25f - 0.5f, sx = ((float)x + 0.5f) * 0.25f - 0.5f;
      sy = sy < 0.f ? 0.f : sy;
      sx = sx < 0.f ? 0.f : sx;
      const int yy0 = (int)sy, xx0 = (int)sx;
      const int yy1 = yy0 + (yy0 < H - 1 ? 1 : 0), xx1 = xx0 + (xx0 < W - 1 ? 1 : 0);
      const float ly = sy - (float)yy0, lx = sx - (float)xx0;
      const float* c = xc + b * xc_bstride;
      const float base = (1.f - ly) * ((1.f - lx) * c[(long long)yy0 * W + xx0] + lx * c[(long long)yy0 * W + xx1]) +
                         ly * ((1.f - lx) * c[(long long)yy1 * W + xx0] + lx * c[(long long)yy1 * W + xx1]);
      out[((long long)b * Hh + y) * Wh + x] = (s + b0) + base;
    }
    __syncthreads();
  }
}

// second half of the fused upsampler tail: out = bias + sum_k taps[(y + dy - 1, x + dx - 1)][k] + bilinear_x4(x_center)
__global__ __launch_bounds__(256) void conv_last_taps_kernel(const float* __restrict__ taps, int ldt,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ xc, long long xc_bstride, int B,
                                                             int Hh, int Wh, float* __restrict__ out) {
  const float b0 = bias[0];
  const int H = Hh >> 2, W = Wh >> 2;
  const long long total = (long long)B * Hh * Wh;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = i % Wh;
    const int y = (i / Wh) % Hh;
    const long long b = i / ((long long)Wh * Hh);
    float s = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y + dy - 1;
      if (yy < 0 || yy >= Hh) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = x + dx - 1;
        if (xx < 0 || xx >= Wh) continue;
        s += taps[((b * Hh + yy) * Wh + xx) * ldt + dy * 3 + dx];
      }
    }
    float sy = ((float)y + 0.5f) * 0.25f - 0.5f, sx = ((float)x + 0.5f) * 0.25f - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int yy0 = (int)sy, xx0 = (int)sx;
    const int yy1 = yy0 + (yy0 < H - 1 ? 1 : 0), xx1 = xx0 + (xx0 < W - 1 ? 1 : 0);
    const float ly = sy - (float)yy0, lx = sx - (float)xx0;
    const float* c = xc + b * xc_bstride;
    const float base = (1.f - ly) * ((1.f - lx) * c[(long long)yy0 * W + xx0] + lx * c[(long long)yy0 * W + xx1]) +
                       ly * ((1.f - lx) * c[(long long)yy1 * W + xx0] + lx * c[(long long)yy1 * W + xx1]);
    out[i] = (s + b0) + base;
  }
}

inline int grid_for(long long threads) {
  long long blocks = (threads + 255) / 256;
  return (int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
}

}  // namespace

extern "C" int cdfo_stem_conv(const float* img, long long img_bstride, const float* w, const float* bias, int B, int H,
                              int W, int act, float* out, int ldo, const float* add, int lda, float* out2, int ldo2,
                              void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || ldo % 4 || (out2 && (!add || lda % 4 || ldo2 % 4))) return CDFO_EINVAL;
  if (!aligned16(out) || (out2 && (!aligned16(out2) || !aligned16(add)))) return CDFO_EALIGN;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_STEM, 2.0*9*64*(double)B*H*W, 4.0*(double)B*H*W*(1+64+(out2?128:0)));
  hipLaunchKernelGGL(stem_conv_kernel, dim3(grid_for((long long)B * H * W * 16)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), img, img_bstride, w, bias, B, H, W, act, out, ldo, add, lda, out2,
                     ldo2);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_stem_conv2(const float* img, long long img_bstride, const float* wA, const float* bA, const float* add,
                               int lda, float* outA, int ldoA, const float* wB, const float* bB, int actB, float* outB,
                               int ldoB, int s2dB, int B, int H, int W, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || !img || !wA || !wB || !add || lda % 4 || ldoA % 4 || ldoB % 4) return CDFO_EINVAL;
  if (s2dB && ((H & 1) || (W & 1) || ldoB < 256)) return CDFO_EINVAL;
  if (!aligned16(outA) || !aligned16(outB) || !aligned16(add)) return CDFO_EALIGN;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_STEM, 2.0*2*9*64*(double)B*H*W, 4.0*(double)B*H*W*(1+192));
  hipLaunchKernelGGL(stem_conv2_kernel, dim3(grid_for((long long)B * H * W * 16)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), img, img_bstride, wA, bA, add, lda, outA, ldoA, wB, bB, actB, outB,
                     ldoB, B, H, W, s2dB);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_layernorm64(const float* in, int ldi, const float* gamma, const float* beta, long long npix,
                                float* out, int ldo, void* stream) {
  if (npix <= 0 || ldi % 4 || ldo % 4) return CDFO_EINVAL;
  if (!aligned16(in) || !aligned16(out) || !aligned16(gamma) || !aligned16(beta)) return CDFO_EALIGN;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_LAYERNORM, 0, 4.0*128*(double)npix);
  hipLaunchKernelGGL(layernorm64_kernel, dim3(grid_for(npix * 16)), dim3(256), 0, static_cast<hipStream_t>(stream), in,
                     ldi, gamma, beta, npix, out, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_dwconv3x3(const float* in, int ldi, const float* w, int B, int H, int W, int C, float* out, int ldo,
                              void* stream) {
  if (B <= 0 || C % 4 || C > 256 || ldi % 4 || ldo % 4) return CDFO_EINVAL;
  if (!aligned16(in) || !aligned16(out)) return CDFO_EALIGN;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_DWCONV, 2.0*9*C*(double)B*H*W, 8.0*C*(double)B*H*W);
  const int cgs = C / 4;
  long long threads = (long long)B * H * ((W + 3) / 4) * cgs;
  long long blocks = (threads + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  // the grid-stride step must be a multiple of the channel-group count (a thread keeps its channel group): with 256
  // threads per block any block count that is a multiple of cgs / gcd(cgs, 256) works
  int g = cgs, t = 256;
  while (t) { const int r = g % t; g = t; t = r; }
  const int unit = cgs / g;
  blocks = (blocks + unit - 1) / unit * unit;
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), in, ldi, w,
                     B, H, W, C, out, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_flow_warp(const float* in, int ldi, const float* mv, long long mv_bstride, int B, int H, int W, int C,
                              float* out, int ldo, void* stream) {
  if (B <= 0 || C % 4 || ldi % 4 || ldo % 4) return CDFO_EINVAL;
  if (!aligned16(in) || !aligned16(out)) return CDFO_EALIGN;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_FLOW_WARP, 0, 4.0*(2*C+2)*(double)B*H*W);
  hipLaunchKernelGGL(flow_warp_kernel, dim3(grid_for((long long)B * H * W * (C / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, ldi, mv, mv_bstride, B, H, W, C, out, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_resample2(const float* in, int ldi, int B, int H, int W, int C, float* out, int ldo, int up,
                              int accumulate, int out_f16, void* stream) {
  if (B <= 0 || C % 4 || ldi % 4 || ldo % 4 || (!up && ((H | W) & 1))) return CDFO_EINVAL;
  if (out_f16 && (!up || accumulate)) return CDFO_EINVAL;
  if (out_f16 < 0 || out_f16 > 2 || (out_f16 == 2 && C % 16)) return CDFO_EINVAL;
  if (!aligned16(in) || !aligned16(out)) return CDFO_EALIGN;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_RESAMPLE, 0, 4.0*C*(double)B*H*W*(up?5.0:1.25));
  if (out_f16 == 2)
    hipLaunchKernelGGL(up2_cp16_kernel, dim3(grid_for((long long)B * (H + 1) * (W + 1) * (C / 4))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), in, ldi, B, H, W, C, reinterpret_cast<_Float16*>(out));
  else if (up)
    hipLaunchKernelGGL(up2_kernel, dim3(grid_for((long long)B * H * W * C)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), in, ldi, B, H, W, C, out, ldo, accumulate, out_f16);
  else
    hipLaunchKernelGGL(down2_kernel, dim3(grid_for((long long)B * H * W * C / 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), in, ldi, B, H, W, C, out, ldo, accumulate);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_scale_channels(const float* in, int ldi, const float* gate, int B, long long P, int C, float* out,
                                   int ldo, void* out_cp16, void* stream) {
  if (B <= 0 || C % 4 || ldi % 4 || ldo % 4 || (out_cp16 && C % 16)) return CDFO_EINVAL;
  if (!aligned16(in) || !aligned16(out) || !aligned16(gate) || !aligned16(out_cp16)) return CDFO_EALIGN;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_SCALE, 0, 8.0*C*(double)B*P);
  hipLaunchKernelGGL(scale_channels_kernel, dim3(grid_for((long long)B * P * (C / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, ldi, gate, B, P, C, out, ldo, static_cast<_Float16*>(out_cp16));
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_conv_last(const float* in, int ldi, const float* w, const float* bias, const float* xc,
                              long long xc_bstride, int B, int Hh, int Wh, float* out, void* stream) {
  if (B <= 0 || (Hh & 3) || (Wh & 3) || ldi % 4) return CDFO_EINVAL;
  if (!aligned16(in)) return CDFO_EALIGN;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_CONV_LAST, 2.0*9*64*(double)B*Hh*Wh, 4.0*65*(double)B*Hh*Wh);
  const long long ntiles = (long long)B * cdiv(Hh, CL_TY) * cdiv(Wh, CL_TX);
  hipLaunchKernelGGL(conv_last_kernel, dim3((unsigned)(ntiles < 4096 ? ntiles : 4096)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, ldi, w, bias, xc, xc_bstride, B, Hh, Wh, out);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_conv_last_taps(const float* taps, int ldt, const float* bias, const float* xc, long long xc_bstride,
                                   int B, int Hh, int Wh, float* out, void* stream) {
  if (B <= 0 || (Hh & 3) || (Wh & 3) || ldt < 9 || !taps || !bias || !xc || !out) return CDFO_EINVAL;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_CONV_LAST, 2.0*9*(double)B*Hh*Wh, 4.0*(ldt + 1)*(double)B*Hh*Wh);
  hipLaunchKernelGGL(conv_last_taps_kernel, dim3(grid_for((long long)B * Hh * Wh)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), taps, ldt, bias, xc, xc_bstride, B, Hh, Wh, out);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_layernorm64_cp16hl(const float* in, int ldi, const float* gamma, const float* beta, int B, long long P,
                                       void* out, void* stream) {
  if (B <= 0 || P <= 0 || ldi % 4 || ldi < 64) return CDFO_EINVAL;
  if (!aligned16(in) || !aligned16(out) || !aligned16(gamma) || !aligned16(beta)) return CDFO_EALIGN;
  const long long npix = (long long)B * P;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_LAYERNORM, 0, 4.0*128*(double)npix);
  hipLaunchKernelGGL(layernorm64_cp16hl_kernel, dim3(grid_for(npix * 16)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     in, ldi, gamma, beta, npix, P, static_cast<_Float16*>(out));
  CDFO_LAUNCH_CHECK();
  return 0;
}
