// Small NCHW operators for the two consumers of the deformable convolution, `DSTA` (ops/attentionlayer.py:86-156) and
// `MVDualAttAlignment` (arch/SIDECVSR_our.py:3265-3352).  Both are module-level parity targets that sit OFF the CVSR_V8
// hot path: their maps are 16 channels wide at ~1/6 resolution (DSTA) or feed straight into the
// NCHW DCN operator, so these are plain one-thread-per-output kernels -- correctness and API completeness, not speed.
#include "common.h"

namespace {

inline int grid_for(long long threads) {
  long long blocks = (threads + 255) / 256;
  return (int)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
}

__global__ __launch_bounds__(256) void conv2d_nchw_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int B, int C, int H, int W,
                                                          int Co, int kh, int kw, int stride, int pad, int Ho, int Wo,
                                                          int act, float* __restrict__ out) {
  const long long total = (long long)B * Co * Ho * Wo;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ox = i % Wo;
    const int oy = (i / Wo) % Ho;
    const int co = (i / ((long long)Wo * Ho)) % Co;
    const long long b = i / ((long long)Wo * Ho * Co);
    float s = bias ? bias[co] : 0.f;
    for (int c = 0; c < C; ++c) {
      const float* ip = in + (b * C + c) * H * W;
      const float* wp = w + ((long long)co * C + c) * kh * kw;
      for (int ky = 0; ky < kh; ++ky) {
        const int iy = oy * stride - pad + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < kw; ++kx) {
          const int ix = ox * stride - pad + kx;
          if (ix < 0 || ix >= W) continue;
          s = fmaf(ip[(long long)iy * W + ix], wp[ky * kw + kx], s);
        }
      }
    }
    out[i] = act_apply(s, act);
  }
}

__global__ __launch_bounds__(256) void maxpool_nchw_kernel(const float* __restrict__ in, int BC, int H, int W, int k,
                                                           int stride, int Ho, int Wo, float* __restrict__ out) {
  const long long total = (long long)BC * Ho * Wo;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ox = i % Wo;
    const int oy = (i / Wo) % Ho;
    const long long bc = i / ((long long)Wo * Ho);
    const float* ip = in + bc * H * W;
    float m = -INFINITY;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) m = fmaxf(m, ip[(long long)(oy * stride + ky) * W + ox * stride + kx]);
    out[i] = m;
  }
}

// F.interpolate(mode='bilinear', align_corners=False, size=(Ho,Wo)); accumulate: out += result
__global__ __launch_bounds__(256) void resize_bilinear_nchw_kernel(const float* __restrict__ in, int BC, int H, int W,
                                                                   int Ho, int Wo, int accumulate,
                                                                   float* __restrict__ out) {
  const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
  const long long total = (long long)BC * Ho * Wo;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ox = i % Wo;
    const int oy = (i / Wo) % Ho;
    const long long bc = i / ((long long)Wo * Ho);
    float sy = ((float)oy + 0.5f) * sh - 0.5f, sx = ((float)ox + 0.5f) * sw - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float* ip = in + bc * H * W;
    float v = (1.f - ly) * ((1.f - lx) * ip[(long long)y0 * W + x0] + lx * ip[(long long)y0 * W + x1]) +
              ly * ((1.f - lx) * ip[(long long)y1 * W + x0] + lx * ip[(long long)y1 * W + x1]);
    if (accumulate) v += out[i];
    out[i] = v;
  }
}

// out[bc] = mean over H*W (one wave per plane)
__global__ __launch_bounds__(64) void avgpool_nchw_kernel(const float* __restrict__ in, long long P,
                                                          float* __restrict__ out) {
  const float* ip = in + (long long)blockIdx.x * P;
  float s = 0.f;
  for (long long p = threadIdx.x; p < P; p += 64) s += ip[p];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[blockIdx.x] = s / (float)P;
}

// mode 0: out = a + b;  1: out = relu(a);  2: out = sigmoid(a);  3: out = x * sigmoid(a) * y[b][c]   (DSTA gate)
__global__ __launch_bounds__(256) void ew_nchw_kernel(const float* __restrict__ a, const float* __restrict__ b2,
                                                      const float* __restrict__ x, const float* __restrict__ y, long long n,
                                                      long long P, int mode, float* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = a[i];
    if (mode == 0) v += b2[i];
    else if (mode == 1) v = fmaxf(v, 0.f);
    else if (mode == 2) v = 1.f / (1.f + expf(-v));
    else v = x[i] * (1.f / (1.f + expf(-v))) * y[i / P];
    out[i] = v;
  }
}

// MVDualAttAlignment offset / mask assembly (arch.py:3336-3350) from the two pixel-major conv_offset outputs
// o1, o2 [B,H,W,27*dg] (ch: o1 | o2 | mask thirds):  offset[b][k] = mag*tanh(oa[k]) + mag*tanh(ob[k]) + flow[b][1 - (k&1)],
// mask[b][k] = sigmoid(ma[k] + mb[k]); outputs NCHW for the DCN operator.
// One workgroup = 64 consecutive pixels of one image; the 27*dg channels are walked in slabs of 48: read along the
// channel axis of the pixel-major inputs (coalesced), transposed through LDS, written along the pixel axis of the NCHW
// outputs (coalesced).
__global__ __launch_bounds__(256) void mv_offset_mask_kernel(const float* __restrict__ o1, const float* __restrict__ o2,
                                                             int ld, const float* __restrict__ flow, long long flow_bstride,
                                                             int B, long long P, int third, float mag,
                                                             float* __restrict__ offset, float* __restrict__ mask) {
  constexpr int SL = 48;
  __shared__ float t[SL][65];
  const int tid = threadIdx.x;
  const long long tiles = (P + 63) / 64;
  const long long b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * 64;
  const int nch = 3 * third;
  for (int k0 = 0; k0 < nch; k0 += SL) {
    const int ns = (nch - k0) < SL ? (nch - k0) : SL;
    __syncthreads();
    for (int idx = tid; idx < 64 * SL; idx += 256) {
      const int px = idx / SL, kl = idx - px * SL;
      const long long p = p0 + px;
      if (kl < ns && p < P) {
        const int k = k0 + kl;
        const float va = o1[(b * P + p) * ld + k], vb = o2[(b * P + p) * ld + k];
        float r;
        if (k < 2 * third) r = mag * tanhf(va) + mag * tanhf(vb) + flow[b * flow_bstride + (long long)(1 - (k & 1)) * P + p];   // flip(1): (y, x) pairs
        else r = 1.f / (1.f + expf(-(va + vb)));
        t[kl][px] = r;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * ns; idx += 256) {
      const int kl = idx >> 6, px = idx & 63;
      const long long p = p0 + px;
      if (p < P) {
        const int k = k0 + kl;
        if (k < 2 * third) offset[(b * 2 * third + k) * P + p] = t[kl][px];
        else mask[(b * third + (k - 2 * third)) * P + p] = t[kl][px];
      }
    }
  }
}

}  // namespace

extern "C" int cdfo_conv2d_nchw(const float* in, const float* w, const float* bias, int B, int C, int H, int W, int Co,
                                int kh, int kw, int stride, int pad, int act, float* out, void* stream) {
  if (B <= 0 || C <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return CDFO_EINVAL;
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return CDFO_EINVAL;
  hipLaunchKernelGGL(conv2d_nchw_kernel, dim3(grid_for((long long)B * Co * Ho * Wo)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, w, bias, B, C, H, W, Co, kh, kw, stride, pad, Ho, Wo, act, out);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_maxpool_nchw(const float* in, int BC, int H, int W, int k, int stride, float* out, void* stream) {
  const int Ho = (H - k) / stride + 1, Wo = (W - k) / stride + 1;
  if (BC <= 0 || k <= 0 || stride <= 0 || Ho <= 0 || Wo <= 0) return CDFO_EINVAL;
  hipLaunchKernelGGL(maxpool_nchw_kernel, dim3(grid_for((long long)BC * Ho * Wo)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, BC, H, W, k, stride, Ho, Wo, out);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_resize_bilinear_nchw(const float* in, int BC, int H, int W, int Ho, int Wo, int accumulate,
                                         float* out, void* stream) {
  if (BC <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return CDFO_EINVAL;
  hipLaunchKernelGGL(resize_bilinear_nchw_kernel, dim3(grid_for((long long)BC * Ho * Wo)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, BC, H, W, Ho, Wo, accumulate, out);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_avgpool_nchw(const float* in, int BC, long long P, float* out, void* stream) {
  if (BC <= 0 || P <= 0) return CDFO_EINVAL;
  hipLaunchKernelGGL(avgpool_nchw_kernel, dim3(BC), dim3(64), 0, static_cast<hipStream_t>(stream), in, P, out);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_ew_nchw(const float* a, const float* b, const float* x, const float* y, long long n, long long P,
                            int mode, float* out, void* stream) {
  if (n <= 0 || mode < 0 || mode > 3 || (mode == 0 && !b) || (mode == 3 && (!x || !y || P <= 0))) return CDFO_EINVAL;
  hipLaunchKernelGGL(ew_nchw_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, x, y, n, P,
                     mode, out);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_mv_offset_mask(const float* o1, const float* o2, int ld, const float* flow, long long flow_bstride,
                                   int B, long long P, int third, float mag, float* offset, float* mask, void* stream) {
  if (B <= 0 || P <= 0 || third <= 0 || ld < 3 * third) return CDFO_EINVAL;
  const long long blocks = (long long)B * ((P + 63) / 64);
  if (blocks > 0x7fffffffLL) return CDFO_EINVAL;
  hipLaunchKernelGGL(mv_offset_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), o1, o2, ld,
                     flow, flow_bstride, B, P, third, mag, offset, mask);
  CDFO_LAUNCH_CHECK();
  return 0;
}
