// Backward kernels of the small NCHW operators (nchw_ops.hip) so that the two consumers of the deformable convolution --
// `DSTA` (ops/attentionlayer.py:117-156) and `MVDualAttAlignment` (arch/SIDECVSR_our.py:3303-3352) -- are trainable like
// their reference classes (cdfo_amd/nchw_autograd.py, cdfo_amd/mv_align.py).  Same design rule as the forward file: these
// maps are 16 channels wide at ~1/6 resolution (or feed straight into the NCHW DCN operator), so every kernel is a plain
// GATHER -- one thread (or one wave) per gradient element, fixed summation order, no atomics: gradients are bit-reproducible.
#include "common.h"

namespace {

inline int grid_for(long long threads) {
  long long blocks = (threads + 255) / 256;
  return (int)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
}

// gin[b][c][iy][ix] = sum_{co,ky,kx : oy*stride - pad + ky == iy, ox*stride - pad + kx == ix} gout[b][co][oy][ox] * w[co][c][ky][kx]
__global__ __launch_bounds__(256) void conv2d_nchw_bwd_input_kernel(const float* __restrict__ gout, const float* __restrict__ w,
                                                                    int B, int C, int H, int W, int Co, int kh, int kw,
                                                                    int stride, int pad, int Ho, int Wo, float* __restrict__ gin) {
  const long long total = (long long)B * C * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ix = i % W;
    const int iy = (i / W) % H;
    const int c = (i / ((long long)W * H)) % C;
    const long long b = i / ((long long)W * H * C);
    float s = 0.f;
    for (int co = 0; co < Co; ++co) {
      const float* gp = gout + (b * Co + co) * Ho * Wo;
      const float* wp = w + ((long long)co * C + c) * kh * kw;
      for (int ky = 0; ky < kh; ++ky) {
        const int ty = iy + pad - ky;
        if (ty < 0 || ty % stride) continue;
        const int oy = ty / stride;
        if (oy >= Ho) continue;
        for (int kx = 0; kx < kw; ++kx) {
          const int tx = ix + pad - kx;
          if (tx < 0 || tx % stride) continue;
          const int ox = tx / stride;
          if (ox >= Wo) continue;
          s = fmaf(gp[(long long)oy * Wo + ox], wp[ky * kw + kx], s);
        }
      }
    }
    gin[i] = s;
  }
}

// one wave per weight element (co, c, ky, kx): gw = sum_{b,oy,ox} gout[b][co][oy][ox] * in[b][c][oy*s-p+ky][ox*s-p+kx];
// waves Co*C*kh*kw .. + Co - 1: the bias gradient of channel co (sum of gout)
__global__ __launch_bounds__(64) void conv2d_nchw_bwd_weight_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                                    int B, int C, int H, int W, int Co, int kh, int kw,
                                                                    int stride, int pad, int Ho, int Wo,
                                                                    float* __restrict__ gw, float* __restrict__ gbias) {
  const long long nw = (long long)Co * C * kh * kw;
  const long long e = blockIdx.x;
  const long long npo = (long long)Ho * Wo;
  float s = 0.f;
  if (e < nw) {
    const int kx = e % kw;
    const int ky = (e / kw) % kh;
    const int c = (e / ((long long)kw * kh)) % C;
    const int co = e / ((long long)kw * kh * C);
    for (long long b = 0; b < B; ++b) {
      const float* gp = gout + (b * Co + co) * npo;
      const float* ip = in + (b * C + c) * H * W;
      for (long long p = threadIdx.x; p < npo; p += 64) {
        const int oy = p / Wo, ox = p - (long long)oy * Wo;
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) s = fmaf(gp[p], ip[(long long)iy * W + ix], s);
      }
    }
    s = wave_sum(s);
    if (threadIdx.x == 0) gw[e] = s;
  } else if (gbias) {
    const int co = (int)(e - nw);
    for (long long b = 0; b < B; ++b) {
      const float* gp = gout + (b * Co + co) * npo;
      for (long long p = threadIdx.x; p < npo; p += 64) s += gp[p];
    }
    s = wave_sum(s);
    if (threadIdx.x == 0) gbias[co] = s;
  }
}

// arg-max of each pooling window (first maximum in scan order, like ATen's max_pool2d), as a flat index into the plane
__global__ __launch_bounds__(256) void maxpool_nchw_argmax_kernel(const float* __restrict__ in, int BC, int H, int W, int k,
                                                                  int stride, int Ho, int Wo, int* __restrict__ idx) {
  const long long total = (long long)BC * Ho * Wo;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ox = i % Wo;
    const int oy = (i / Wo) % Ho;
    const long long bc = i / ((long long)Wo * Ho);
    const float* ip = in + bc * H * W;
    float m = -INFINITY;
    int am = (oy * stride) * W + ox * stride;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) {
        const int q = (oy * stride + ky) * W + ox * stride + kx;
        const float v = ip[q];
        if (v > m || v != v) { m = v; am = q; }
      }
    idx[i] = am;
  }
}

// gin[bc][iy][ix] = sum over the windows that contain (iy, ix) and whose arg-max it is
__global__ __launch_bounds__(256) void maxpool_nchw_bwd_kernel(const float* __restrict__ gout, const int* __restrict__ idx,
                                                               int BC, int H, int W, int k, int stride, int Ho, int Wo,
                                                               float* __restrict__ gin) {
  const long long total = (long long)BC * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ix = i % W;
    const int iy = (i / W) % H;
    const long long bc = i / ((long long)W * H);
    const int me = iy * W + ix;
    int oy0 = iy - k + 1;
    oy0 = oy0 <= 0 ? 0 : (oy0 + stride - 1) / stride;
    int ox0 = ix - k + 1;
    ox0 = ox0 <= 0 ? 0 : (ox0 + stride - 1) / stride;
    const int oy1 = min(Ho - 1, iy / stride), ox1 = min(Wo - 1, ix / stride);
    float s = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy)
      for (int ox = ox0; ox <= ox1; ++ox) {
        const long long o = (bc * Ho + oy) * Wo + ox;
        if (idx[o] == me) s += gout[o];
      }
    gin[i] = s;
  }
}

// adjoint of resize_bilinear_nchw_kernel: gin[bc][y][x] = sum over the outputs whose four taps include (y, x) of their weight
// on it * gout.  Every candidate output re-evaluates the forward's own tap arithmetic (clamps included), so the two agree exactly.
__global__ __launch_bounds__(256) void resize_bilinear_nchw_bwd_kernel(const float* __restrict__ gout, int BC, int H, int W,
                                                                       int Ho, int Wo, float* __restrict__ gin) {
  const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
  const long long total = (long long)BC * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = i % W;
    const int y = (i / W) % H;
    const long long bc = i / ((long long)W * H);
    int oy_lo = (int)floorf(((float)y - 1.5f) / sh - 0.5f) - 1, oy_hi = (int)ceilf(((float)y + 1.5f) / sh) + 1;
    int ox_lo = (int)floorf(((float)x - 1.5f) / sw - 0.5f) - 1, ox_hi = (int)ceilf(((float)x + 1.5f) / sw) + 1;
    if (y == 0) oy_lo = 0;
    if (y == H - 1) oy_hi = Ho - 1;
    if (x == 0) ox_lo = 0;
    if (x == W - 1) ox_hi = Wo - 1;
    oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, Ho - 1);
    ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, Wo - 1);
    const float* gp = gout + bc * Ho * Wo;
    float s = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      float sy = ((float)oy + 0.5f) * sh - 0.5f;
      sy = sy < 0.f ? 0.f : sy;
      const int y0 = (int)sy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
      const float ly = sy - (float)y0;
      const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        float sx = ((float)ox + 0.5f) * sw - 0.5f;
        sx = sx < 0.f ? 0.f : sx;
        const int x0 = (int)sx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float lx = sx - (float)x0;
        const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        if (wx != 0.f) s = fmaf(wy * wx, gp[(long long)oy * Wo + ox], s);
      }
    }
    gin[i] = s;
  }
}

// mode 1: ga = g * (y > 0)            (y = relu output)
//      2: ga = g * y * (1 - y)        (y = sigmoid output)
//      4: ga = g[bc] / P              (adjoint of the plane mean: g is [BC])
//      5: ga = g * x * yv[bc] * s(1-s), s = sigmoid(a)    (gate out = x * s * yv: gradient w.r.t. a)
//      6: gx = g * s * yv[bc]                             (gate: gradient w.r.t. x)
__global__ __launch_bounds__(256) void ew_nchw_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                          const float* __restrict__ a, const float* __restrict__ x,
                                                          const float* __restrict__ yv, long long n, long long P, int mode,
                                                          float* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v;
    if (mode == 1) v = y[i] > 0.f ? g[i] : 0.f;
    else if (mode == 2) v = g[i] * y[i] * (1.f - y[i]);
    else if (mode == 4) v = g[i / P] / (float)P;
    else {
      const float s = 1.f / (1.f + expf(-a[i]));
      v = mode == 5 ? g[i] * x[i] * yv[i / P] * s * (1.f - s) : g[i] * s * yv[i / P];
    }
    out[i] = v;
  }
}

// gate out = x * sigmoid(a) * yv[bc]: gyv[bc] = sum_p g * x * sigmoid(a)   (one wave per plane)
__global__ __launch_bounds__(64) void gate_nchw_bwd_y_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                                             const float* __restrict__ x, long long P, float* __restrict__ gy) {
  const long long o = (long long)blockIdx.x * P;
  float s = 0.f;
  for (long long p = threadIdx.x; p < P; p += 64) s = fmaf(g[o + p] * x[o + p], 1.f / (1.f + expf(-a[o + p])), s);
  s = wave_sum(s);
  if (threadIdx.x == 0) gy[blockIdx.x] = s;
}

// adjoint of mv_offset_mask_kernel (nchw_ops.hip) w.r.t. the two conv_offset outputs: o1, o2, g1, g2 pixel-major [B,P,3*third]
// (pitch ld), goff [B][2*third][P], gmask [B][third][P] NCHW.  For k < 2*third: g1[k] = goff[k] * mag * (1 - tanh(o1[k])^2)
// (g2 with o2); for the mask third: g1[k] = g2[k] = gmask[k'] * s * (1 - s), s = sigmoid(o1[k] + o2[k]).  The motion field is an
// input of the network (no gradient).  Same 48-channel slabs through LDS as the forward: NCHW reads and pixel-major writes coalesced.
__global__ __launch_bounds__(256) void mv_offset_mask_bwd_kernel(const float* __restrict__ o1, const float* __restrict__ o2, int ld,
                                                                 const float* __restrict__ goff, const float* __restrict__ gmask,
                                                                 int B, long long P, int third, float mag,
                                                                 float* __restrict__ g1, float* __restrict__ g2) {
  constexpr int SL = 48;
  __shared__ float t[SL][65];
  const int tid = threadIdx.x;
  const long long tiles = (P + 63) / 64;
  const long long b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * 64;
  const int nch = 3 * third;
  for (int k0 = 0; k0 < nch; k0 += SL) {
    const int ns = (nch - k0) < SL ? (nch - k0) : SL;
    __syncthreads();
    for (int idx = tid; idx < 64 * ns; idx += 256) {
      const int kl = idx >> 6, px = idx & 63;
      const long long p = p0 + px;
      if (p < P) {
        const int k = k0 + kl;
        t[kl][px] = k < 2 * third ? goff[(b * 2 * third + k) * P + p] : gmask[(b * third + (k - 2 * third)) * P + p];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * SL; idx += 256) {
      const int px = idx / SL, kl = idx - px * SL;
      const long long p = p0 + px;
      if (kl < ns && p < P) {
        const int k = k0 + kl;
        const long long q = (b * P + p) * ld + k;
        const float va = o1[q], vb = o2[q], g = t[kl][px];
        if (k < 2 * third) {
          const float ta = tanhf(va), tb = tanhf(vb);
          g1[q] = g * mag * (1.f - ta * ta);
          g2[q] = g * mag * (1.f - tb * tb);
        } else {
          const float s = 1.f / (1.f + expf(-(va + vb)));
          const float r = g * s * (1.f - s);
          g1[q] = r;
          g2[q] = r;
        }
      }
    }
  }
}

}  // namespace

extern "C" int cdfo_conv2d_nchw_bwd(const float* in, const float* w, const float* gout, int B, int C, int H, int W, int Co,
                                    int kh, int kw, int stride, int pad, float* gin, float* gw, float* gbias, void* stream) {
  if (B <= 0 || C <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 || !gout) return CDFO_EINVAL;
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  if (Ho <= 0 || Wo <= 0 || (gin && !w) || ((gw || gbias) && !in && gw)) return CDFO_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (gin)
    hipLaunchKernelGGL(conv2d_nchw_bwd_input_kernel, dim3(grid_for((long long)B * C * H * W)), dim3(256), 0, st, gout, w, B, C, H,
                       W, Co, kh, kw, stride, pad, Ho, Wo, gin);
  if (gw) {
    const long long waves = (long long)Co * C * kh * kw + (gbias ? Co : 0);
    if (waves > 0x7fffffffLL) return CDFO_EINVAL;
    hipLaunchKernelGGL(conv2d_nchw_bwd_weight_kernel, dim3((unsigned)waves), dim3(64), 0, st, in, gout, B, C, H, W, Co, kh, kw,
                       stride, pad, Ho, Wo, gw, gbias);
  }
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_maxpool_nchw_bwd(const float* in, const float* gout, int BC, int H, int W, int k, int stride, int* idx_scratch,
                                     float* gin, void* stream) {
  const int Ho = (H - k) / stride + 1, Wo = (W - k) / stride + 1;
  if (BC <= 0 || k <= 0 || stride <= 0 || Ho <= 0 || Wo <= 0 || !idx_scratch || (long long)H * W >= (1ll << 31)) return CDFO_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(maxpool_nchw_argmax_kernel, dim3(grid_for((long long)BC * Ho * Wo)), dim3(256), 0, st, in, BC, H, W, k, stride,
                     Ho, Wo, idx_scratch);
  hipLaunchKernelGGL(maxpool_nchw_bwd_kernel, dim3(grid_for((long long)BC * H * W)), dim3(256), 0, st, gout, idx_scratch, BC, H, W,
                     k, stride, Ho, Wo, gin);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_resize_bilinear_nchw_bwd(const float* gout, int BC, int H, int W, int Ho, int Wo, float* gin, void* stream) {
  if (BC <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return CDFO_EINVAL;
  hipLaunchKernelGGL(resize_bilinear_nchw_bwd_kernel, dim3(grid_for((long long)BC * H * W)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), gout, BC, H, W, Ho, Wo, gin);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_ew_nchw_bwd(const float* g, const float* y, const float* a, const float* x, const float* yv, long long n,
                                long long P, int mode, float* out, void* stream) {
  if (n <= 0 || !g || !out) return CDFO_EINVAL;
  if (((mode == 1 || mode == 2) && !y) || (mode == 4 && P <= 0) || ((mode == 5 || mode == 6) && (!a || !yv || P <= 0 || (mode == 5 && !x))))
    return CDFO_EINVAL;
  if (mode != 1 && mode != 2 && mode != 4 && mode != 5 && mode != 6) return CDFO_EINVAL;
  hipLaunchKernelGGL(ew_nchw_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), g, y, a, x, yv, n, P,
                     mode, out);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_gate_nchw_bwd_y(const float* g, const float* a, const float* x, int BC, long long P, float* gy, void* stream) {
  if (BC <= 0 || P <= 0) return CDFO_EINVAL;
  hipLaunchKernelGGL(gate_nchw_bwd_y_kernel, dim3(BC), dim3(64), 0, static_cast<hipStream_t>(stream), g, a, x, P, gy);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_mv_offset_mask_bwd(const float* o1, const float* o2, int ld, const float* goff, const float* gmask, int B,
                                       long long P, int third, float mag, float* g1, float* g2, void* stream) {
  if (B <= 0 || P <= 0 || third <= 0 || ld < 3 * third) return CDFO_EINVAL;
  const long long blocks = (long long)B * ((P + 63) / 64);
  if (blocks > 0x7fffffffLL) return CDFO_EINVAL;
  hipLaunchKernelGGL(mv_offset_mask_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), o1, o2, ld,
                     goff, gmask, B, P, third, mag, g1, g2);
  CDFO_LAUNCH_CHECK();
  return 0;
}
