// Direct (VALU) convolutions for the thin 16-channel layers of the prior U-net `side_to_feaoneUDSA_2`
// (arch/SIDECVSR_our.py:1815-1834): 3x3 stride-2 pad-2 convs, the two stride-2 transposed convs, and the
// SpatialAttention gate (arch.py:2719-2730, ChannelPool 1883-1885).  They run at 1/4 and 1/16 of the LR pixel
// count and carry < 0.1 % of the FLOPs, so they stay off the matrix cores.
#include "common.h"

namespace {

// thread = (output pixel, 4 output channels); weights staged in LDS as [tap][cin][cout]
template <int CIN, int COUT, bool TRANSPOSED>
__global__ __launch_bounds__(256) void small_conv3x3_kernel(const float* __restrict__ in, int ldi,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias, int B, int H, int W, int Ho,
                                                            int Wo, int stride, int pad, int act,
                                                            float* __restrict__ out, int ldo, _Float16* __restrict__ out_hl) {
  __shared__ __attribute__((aligned(16))) float sw[9 * CIN * COUT];
  for (int i = threadIdx.x; i < 9 * CIN * COUT; i += blockDim.x) {
    const int co = i % COUT, ci = (i / COUT) % CIN, t = i / (COUT * CIN);
    // Conv2d weight is [co][ci][t]; ConvTranspose2d weight is [ci][co][t]
    sw[i] = TRANSPOSED ? w[(ci * COUT + co) * 9 + t] : w[(co * CIN + ci) * 9 + t];
  }
  __syncthreads();
  constexpr int CGS = COUT / 4;
  const long long total = (long long)B * Ho * Wo * CGS;
  // A stride-2 transposed convolution reaches an output pixel through the taps whose parity matches its own (1, 2 or 4 of the
  // 9): pixels are enumerated phase by phase ((oy & 1, ox & 1) major) so that a wave's 16 pixels share their tap set and the
  // `continue`s below are wave-uniform -- in raster order every wave walked all nine taps under partial masks (4x the work)
  const bool phase_major = TRANSPOSED && stride == 2 && !(Ho & 1) && !(Wo & 1);
  const int Wq = Wo >> 1;
  const long long quarter = (long long)(Ho >> 1) * Wq;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cg = i % CGS;
    long long p = i / CGS;
    int ox, oy;
    const long long b = p / ((long long)Wo * Ho);
    if (phase_major) {
      const long long rr = p - b * (long long)Wo * Ho;
      const int ph = (int)(rr / quarter);
      const long long cell = rr - ph * quarter;
      oy = 2 * (int)(cell / Wq) + (ph >> 1);
      ox = 2 * (int)(cell % Wq) + (ph & 1);
      p = (b * Ho + oy) * Wo + ox;
    } else {
      ox = p % Wo;
      oy = (p / Wo) % Ho;
    }
    f32x4 acc = bias ? *reinterpret_cast<const f32x4*>(bias + cg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      int iy;
      if (TRANSPOSED) {  // oy = iy*stride - pad + ky
        const int num = oy + pad - ky;
        if (num < 0 || num % stride) continue;
        iy = num / stride;
      } else {
        iy = oy * stride - pad + ky;
      }
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        int ix;
        if (TRANSPOSED) {
          const int num = ox + pad - kx;
          if (num < 0 || num % stride) continue;
          ix = num / stride;
        } else {
          ix = ox * stride - pad + kx;
        }
        if (ix < 0 || ix >= W) continue;
        const float* ip = in + ((b * H + iy) * W + ix) * ldi;
        const float* wp = sw + (ky * 3 + kx) * CIN * COUT + cg * 4;
#pragma unroll
        for (int c4 = 0; c4 < CIN / 4; ++c4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(ip + c4 * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc += v[e] * *reinterpret_cast<const f32x4*>(wp + (c4 * 4 + e) * COUT);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = act_apply(acc[e], act);
    if (out_hl) {        // fp16 hi | lo planes, chunk-planar [B][2][Ho*Wo][16]: the split-fp16 source of cdfo_conv3x3_ring
      typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
      f16x4_t hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { hi[e] = (_Float16)acc[e]; lo[e] = (_Float16)(acc[e] - (float)hi[e]); }
      const long long P = (long long)Ho * Wo, pix = p - b * P;
      _Float16* o16 = out_hl + ((b * 2) * P + pix) * 16 + cg * 4;
      *reinterpret_cast<f16x4_t*>(o16) = hi;
      *reinterpret_cast<f16x4_t*>(o16 + P * 16) = lo;
    } else {
      *reinterpret_cast<f32x4*>(out + p * ldo + cg * 4) = acc;
    }
  }
}

// SpatialAttention on a 16-channel map: x * sigmoid(conv7x7([max_c x, mean_c x]) + b)
__global__ __launch_bounds__(256) void spatial_gate16_kernel(const float* __restrict__ in, int ldi,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias, int B, int H, int W,
                                                             float* __restrict__ out, int ldo) {
  __shared__ float sw[98];
  if (threadIdx.x < 98) sw[threadIdx.x] = w[threadIdx.x];  // [1][2][7][7]: plane 0 = max, plane 1 = mean
  __syncthreads();
  const long long npix = (long long)B * H * W;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix;
       p += (long long)gridDim.x * blockDim.x) {
    const int x = p % W;
    const int y = (p / W) % H;
    const long long b = p / ((long long)W * H);
    float s = bias[0];
    for (int dy = 0; dy < 7; ++dy) {
      const int yy = y + dy - 3;
      if (yy < 0 || yy >= H) continue;
      for (int dx = 0; dx < 7; ++dx) {
        const int xx = x + dx - 3;
        if (xx < 0 || xx >= W) continue;
        const float* q = in + ((b * H + yy) * W + xx) * ldi;
        float mx = -INFINITY, sm = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(q + c4 * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { mx = fmaxf(mx, v[e]); sm += v[e]; }
        }
        s += mx * sw[dy * 7 + dx] + (sm * (1.f / 16.f)) * sw[49 + dy * 7 + dx];
      }
    }
    const float g = 1.f / (1.f + expf(-s));
    const float* q = in + p * ldi;
    float* o = out + p * ldo;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
      *reinterpret_cast<f32x4*>(o + c4 * 4) = *reinterpret_cast<const f32x4*>(q + c4 * 4) * g;
  }
}

inline int grid_for(long long threads) {
  long long blocks = (threads + 255) / 256;
  return (int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
}

// ---------------------------------------------------------------------------------------------------------------
// First layer of the prior U-net in the feature extractor's first round, straight from the one-channel prior image:
//     t = lrelu( body.0( conv_second(pms) ) )      (arch.py:4420 conv_second -- no activation -- then 1819-1820 body.0)
// Both are 3x3 zero-padded convolutions with nothing but their biases in between, and conv_second's 64-channel result
// feeds only this layer in round 0 (arch.py:1463-1468), so it never needs to exist: for every tap t of body.0 whose
// position p + t lies inside the image (outside it body.0 sees its zero padding, not conv_second's bias),
//     out[o](p) = b0[o] + sum_t [p + t inside] * ( bt[t][o] + sum_u wc[t][u][o] * pms(p + t + u) ),
//     wc[t][u][o] = sum_c W0[o][c][t] W2[c][u],   bt[t][o] = sum_c W0[o][c][t] b2[c]     (composed on the host in fp64)
// -- exact at the borders, 81 x 16 multiply-adds per pixel instead of a 64-channel tensor written and read back.
// thread = (pixel, 4 output channels).
__global__ __launch_bounds__(256) void udsa_head_kernel(const float* __restrict__ img, long long bstride,
                                                        const float* __restrict__ wc, const float* __restrict__ bt,
                                                        const float* __restrict__ b0, int B, int H, int W,
                                                        float* __restrict__ out, int ldo) {
  __shared__ __attribute__((aligned(16))) float sw[81 * 16 + 9 * 16 + 16];
  for (int i = threadIdx.x; i < 81 * 16; i += blockDim.x) sw[i] = wc[i];
  for (int i = threadIdx.x; i < 9 * 16; i += blockDim.x) sw[81 * 16 + i] = bt[i];
  for (int i = threadIdx.x; i < 16; i += blockDim.x) sw[81 * 16 + 9 * 16 + i] = b0[i];
  __syncthreads();
  const long long total = (long long)B * H * W * 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i & 3);
    const long long p = i >> 2;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const long long b = p / ((long long)W * H);
    const float* im = img + b * bstride;
    float win[5][5];
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        const int yy = y + dy - 2, xx = x + dx - 2;
        win[dy][dx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? im[(long long)yy * W + xx] : 0.f;
      }
    f32x4 acc = *reinterpret_cast<const f32x4*>(sw + 81 * 16 + 9 * 16 + cg * 4);
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int yy = y + ty - 1, xx = x + tx - 1;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const int t = ty * 3 + tx;
        f32x4 a = *reinterpret_cast<const f32x4*>(sw + 81 * 16 + t * 16 + cg * 4);
#pragma unroll
        for (int uy = 0; uy < 3; ++uy)
#pragma unroll
          for (int ux = 0; ux < 3; ++ux)
            a += *reinterpret_cast<const f32x4*>(sw + (t * 9 + uy * 3 + ux) * 16 + cg * 4) * win[ty + uy][tx + ux];
        acc += a;
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = acc[k] > 0.f ? acc[k] : 0.1f * acc[k];
    *reinterpret_cast<f32x4*>(out + p * ldo + cg * 4) = acc;
  }
}

// Round 5: the same convolutions with FOUR output pixels per thread.  The kernel above reads one f32x4 of weights from LDS per four
// multiply-adds (144 ds_read_b128 per output quad at nine taps): it is bound by the LDS pipe, 4x above its HBM time.  Here a wave owns
// 64 cells of one output row -- of one output PHASE for the stride-2 transposed form, so that the tap set (1, 2 or 4 of the 9) is the
// same for every cell -- and lane (g, cg) computes output channels 4 cg .. + 3 of cells g, g + 16, g + 32, g + 48: every weight
// vector read from LDS feeds sixteen multiply-adds, a store instruction covers sixteen consecutive pixels (1 KiB fp32, 512 B of an
// fp16 plane).  Row and tap-parity tests are wave-uniform (scalar branches); a cell outside the row or a tap outside the image loads
// from a clamped address and is zeroed by a select.  stride 1 or 2.
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void small_conv16_px4_kernel(const float* __restrict__ in, int ldi, const float* __restrict__ w,
                                                               const float* __restrict__ bias, int B, int H, int W, int Ho, int Wo,
                                                               int stride, int pad, int act, float* __restrict__ out, int ldo,
                                                               _Float16* __restrict__ out_hl) {
  constexpr int CIN = 16, COUT = 16;
  __shared__ __attribute__((aligned(16))) float sw[9 * CIN * COUT];
  for (int i = threadIdx.x; i < 9 * CIN * COUT; i += blockDim.x) {
    const int co = i % COUT, ci = (i / COUT) % CIN, t = i / (COUT * CIN);
    sw[i] = TRANSPOSED ? w[(ci * COUT + co) * 9 + t] : w[(co * CIN + ci) * 9 + t];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 2, cg = lane & 3;
  const bool ph2 = TRANSPOSED && stride == 2;
  const int Hr = ph2 ? (Ho + 1) >> 1 : Ho, Wr = ph2 ? (Wo + 1) >> 1 : Wo, nph = ph2 ? 4 : 1;
  const int wpr = (Wr + 63) >> 6;
  const long long nwu = (long long)B * nph * Hr * wpr;
  const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + cg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (long long wu = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); wu < nwu; wu += (long long)gridDim.x * 4) {
    const int xw = (int)(wu % wpr);
    long long t = wu / wpr;
    const int cy = (int)(t % Hr);
    t /= Hr;
    const int ph = (int)(t % nph);
    const long long b = t / nph;
    const int phy = ph >> 1, phx = ph & 1;
    const int oy = ph2 ? 2 * cy + phy : cy;
    if (oy >= Ho) continue;
    const int ncell = ph2 ? (Wo - phx + 1) >> 1 : Wo;
    const int cx0 = xw * 64 + g;                      // cells cx0 + 16 j, j = 0 .. 3
    const int ox0 = ph2 ? 2 * cx0 + phx : cx0, oxs = ph2 ? 32 : 16;     // output column of cell j: ox0 + oxs * j
    f32x4 acc[4] = {b4, b4, b4, b4};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      int iy;
      if (TRANSPOSED) {                               // oy = iy * stride - pad + ky
        const int num = oy + pad - ky;
        if (num < 0 || (stride == 2 && (num & 1))) continue;
        iy = stride == 2 ? num >> 1 : num;
      } else {
        iy = oy * stride - pad + ky;
      }
      if (iy < 0 || iy >= H) continue;
      const float* irow = in + (b * H + iy) * (long long)W * ldi;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        int ix0, ixs;                                 // input column of cell j: ix0 + ixs * j
        if (TRANSPOSED) {
          const int num = ox0 + pad - kx;             // (its parity is the phase's: the same in every cell of the wave)
          if (stride == 2 && ((phx + pad - kx) & 1)) continue;
          ix0 = stride == 2 ? num >> 1 : num;
          ixs = 16;
        } else {
          ix0 = ox0 * stride - pad + kx;
          ixs = 16 * stride;
        }
        const float* ip[4];
        bool ok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ix = ix0 + ixs * j;
          ok[j] = ix >= 0 && ix < W && cx0 + 16 * j < ncell;
          ip[j] = irow + (long long)(ok[j] ? ix : 0) * ldi;
        }
        const float* wp = sw + (ky * 3 + kx) * CIN * COUT + cg * 4;
#pragma unroll
        for (int c4 = 0; c4 < CIN / 4; ++c4) {
          f32x4 v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = *reinterpret_cast<const f32x4*>(ip[j] + c4 * 4);
            if (!ok[j]) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + (c4 * 4 + e) * COUT);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += v[j][e] * wv;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (cx0 + 16 * j >= ncell) continue;
      const long long p = (b * Ho + oy) * Wo + ox0 + oxs * j;
      f32x4 a4 = acc[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) a4[e] = act_apply(a4[e], act);
      if (out_hl) {        // fp16 hi | lo planes, chunk-planar [B][2][Ho*Wo][16]: the split-fp16 source of cdfo_conv3x3_ring
        typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
        f16x4_t hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = (_Float16)a4[e]; lo[e] = (_Float16)(a4[e] - (float)hi[e]); }
        const long long P = (long long)Ho * Wo, pix = p - b * P;
        _Float16* o16 = out_hl + ((b * 2) * P + pix) * 16 + cg * 4;
        *reinterpret_cast<f16x4_t*>(o16) = hi;
        *reinterpret_cast<f16x4_t*>(o16 + P * 16) = lo;
      } else {
        *reinterpret_cast<f32x4*>(out + p * ldo + cg * 4) = a4;
      }
    }
  }
}

}  // namespace

static int small_conv16_launch(const float* in, int ldi, const float* w, const float* bias, int B, int H, int W, int stride,
                               int pad, int out_pad, int transposed, int act, float* out, int ldo, void* out_hl, void* stream) {
  if (B <= 0 || ldi % 4 || (!out_hl && ldo % 4) || stride < 1) return CDFO_EINVAL;
  if (!aligned16(in) || (out && !aligned16(out)) || (out_hl && !aligned16(out_hl)) || (bias && !aligned16(bias))) return CDFO_EALIGN;
  if (!out && !out_hl) return CDFO_EINVAL;
  int Ho, Wo;
  if (transposed) {
    Ho = (H - 1) * stride - 2 * pad + 3 + out_pad;
    Wo = (W - 1) * stride - 2 * pad + 3 + out_pad;
  } else {
    Ho = (H + 2 * pad - 3) / stride + 1;
    Wo = (W + 2 * pad - 3) / stride + 1;
  }
  if (Ho <= 0 || Wo <= 0) return CDFO_EINVAL;
  const int grid = grid_for((long long)B * Ho * Wo * 4);
  hipStream_t st = static_cast<hipStream_t>(stream);
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_SMALL_CONV, 2.0*9*256*(double)B*Ho*Wo, 4.0*16*((double)B*Ho*Wo+(double)B*H*W));
  _Float16* hl = static_cast<_Float16*>(out_hl);
  // four output pixels per thread (round 5) for the transposed forms: 70x122 -> 137x241 0.162 -> 0.099 ms, 137x241 -> 272x480 (hi | lo
  // planes) 0.471 -> 0.309 ms at 56 frames; the plain stride-2 convolutions gain nothing from it (0.268 -> 0.297, 0.095 -> 0.083 ms:
  // they are bound by their nine strided input reads per output, not by the weight reads) and stay on the one-pixel kernel
  if (transposed && stride <= 2) {
    const bool ph2 = transposed && stride == 2;
    const long long waves = (long long)B * (ph2 ? 4 : 1) * (ph2 ? (Ho + 1) / 2 : Ho) * (((ph2 ? (Wo + 1) / 2 : Wo) + 63) / 64);
    const int g4 = grid_for(waves * 64);
    hipLaunchKernelGGL(small_conv16_px4_kernel<true>, dim3(g4), dim3(256), 0, st, in, ldi, w, bias, B, H, W, Ho, Wo, stride, pad, act,
                       out, ldo, hl);
    CDFO_LAUNCH_CHECK();
    return 0;
  }
  if (transposed)
    hipLaunchKernelGGL((small_conv3x3_kernel<16, 16, true>), dim3(grid), dim3(256), 0, st, in, ldi, w, bias, B, H, W, Ho,
                       Wo, stride, pad, act, out, ldo, hl);
  else
    hipLaunchKernelGGL((small_conv3x3_kernel<16, 16, false>), dim3(grid), dim3(256), 0, st, in, ldi, w, bias, B, H, W, Ho,
                       Wo, stride, pad, act, out, ldo, hl);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_small_conv16(const float* in, int ldi, const float* w, const float* bias, int B, int H, int W,
                                 int stride, int pad, int out_pad, int transposed, int act, float* out, int ldo,
                                 void* stream) {
  return small_conv16_launch(in, ldi, w, bias, B, H, W, stride, pad, out_pad, transposed, act, out, ldo, nullptr, stream);
}

// The same with the result written as fp16 hi | lo planes, chunk-planar [B][2][Ho*Wo][16] (hi = fp16(v), lo = fp16(v - hi)):
// the split-fp16 source of cdfo_conv3x3_ring (plane_wrap = 2) for the 16 -> 64 convolution that follows it in the prior U-net.
extern "C" int cdfo_small_conv16_hl(const float* in, int ldi, const float* w, const float* bias, int B, int H, int W,
                                    int stride, int pad, int out_pad, int transposed, int act, void* out_hl, void* stream) {
  return small_conv16_launch(in, ldi, w, bias, B, H, W, stride, pad, out_pad, transposed, act, nullptr, 0, out_hl, stream);
}

extern "C" int cdfo_spatial_gate16(const float* in, int ldi, const float* w, const float* bias, int B, int H, int W,
                                   float* out, int ldo, void* stream) {
  if (B <= 0 || ldi % 4 || ldo % 4) return CDFO_EINVAL;
  if (!aligned16(in) || !aligned16(out)) return CDFO_EALIGN;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_SPATIAL_GATE, 0, 4.0*32*(double)B*H*W);
  hipLaunchKernelGGL(spatial_gate16_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, ldi, w, bias, B, H, W, out, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}

// lrelu(body.0(conv_second(img))) of the prior U-net's first round from the one-channel image (see udsa_head_kernel).
// img element [b][y][x] at img + b*img_bstride + y*W + x; wc [9][9][16], bt [9][16], b0 [16]; out [B][H][W][16], pitch ldo.
extern "C" int cdfo_udsa_head(const float* img, long long img_bstride, const float* wc, const float* bt, const float* b0, int B,
                              int H, int W, float* out, int ldo, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || ldo % 4 || ldo < 16 || !img || !wc || !bt || !b0) return CDFO_EINVAL;
  if (!aligned16(out)) return CDFO_EALIGN;
  const long long total = (long long)B * H * W * 4;
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(udsa_head_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), img, img_bstride, wc,
                     bt, b0, B, H, W, out, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}
