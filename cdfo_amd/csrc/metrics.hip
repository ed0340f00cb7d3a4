// Evaluation metrics on the device: PSNR and SSIM of single-channel frames with the reference's
// semantics (metric/psnr_ssim.py:278-317 calculate_psnr, :320-351 _ssim, :353-399 calculate_ssim), so that the evaluation
// loop needs no D2H copy / PNG round trip per frame.  Inputs are [N][H][W] fp32 planes on a 0..255 scale (`scale` is
// applied first, e.g. 255 for network outputs in [0,1]; `clamp` != 0 clamps to [0,255] and `round8` != 0 rounds to
// integers like the PNG the reference writes); `crop` border pixels are dropped.  All sums are in fp64, as in the
// reference.
#include "common.h"

namespace {

__device__ __forceinline__ double load_px(const float* p, float scale, int clamp, int round8) {
  float v = *p * scale;
  if (clamp) v = fminf(fmaxf(v, 0.f), 255.f);
  if (round8) v = rintf(v);
  return (double)v;
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  __syncthreads();
  return t;   // valid in thread 0
}

// partial[n][blockIdx.x] = sum over this block's pixels of (a - b)^2 inside the cropped region
__global__ __launch_bounds__(256) void sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, int H, int W,
                                                     int crop, float scale, int clamp, int round8,
                                                     double* __restrict__ partial) {
  __shared__ double sh[4];
  const int n = blockIdx.y, Hc = H - 2 * crop, Wc = W - 2 * crop;
  const long long total = (long long)Hc * Wc;
  double s = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int y = (int)(i / Wc) + crop, x = (int)(i % Wc) + crop;
    const long long o = ((long long)n * H + y) * W + x;
    const double d = load_px(a + o, scale, clamp, round8) - load_px(b + o, scale, clamp, round8);
    s += d * d;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) partial[(long long)n * gridDim.x + blockIdx.x] = s;
}

// SSIM map sums: 11x11 Gaussian (sigma 1.5) window, 'valid' positions of the cropped image only.
// One thread per output position, 121 taps from global memory (the metric is not on the hot path).
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, int H, int W,
                                                   int crop, float scale, int clamp, int round8,
                                                   double* __restrict__ partial) {
  __shared__ double sh[4];
  __shared__ double g[11];
  if (threadIdx.x < 11) {
    double s = 0.0;
    for (int i = 0; i < 11; ++i) s += exp(-((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
    g[threadIdx.x] = exp(-((int)(threadIdx.x - 5) * (int)(threadIdx.x - 5)) / (2.0 * 1.5 * 1.5)) / s;   // cv2.getGaussianKernel(11, 1.5)
  }
  __syncthreads();
  const int n = blockIdx.y, Hc = H - 2 * crop, Wc = W - 2 * crop, Ho = Hc - 10, Wo = Wc - 10;
  const long long total = (long long)Ho * Wo;
  const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
  double acc = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int y0 = (int)(i / Wo) + crop, x0 = (int)(i % Wo) + crop;
    double m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
    for (int dy = 0; dy < 11; ++dy) {
      const long long row = ((long long)n * H + y0 + dy) * W + x0;
      for (int dx = 0; dx < 11; ++dx) {
        const double w = g[dy] * g[dx];
        const double p = load_px(a + row + dx, scale, clamp, round8), q = load_px(b + row + dx, scale, clamp, round8);
        m1 += w * p; m2 += w * q; s11 += w * p * p; s22 += w * q * q; s12 += w * p * q;
      }
    }
    const double v1 = s11 - m1 * m1, v2 = s22 - m2 * m2, c = s12 - m1 * m2;
    acc += ((2 * m1 * m2 + C1) * (2 * c + C2)) / ((m1 * m1 + m2 * m2 + C1) * (v1 + v2 + C2));
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[(long long)n * gridDim.x + blockIdx.x] = acc;
}

}  // namespace

// partial: [N][nblocks] doubles (nblocks <= 1024 chosen by the callee and returned through *nblocks_out); the caller sums
// them (fixed order => deterministic) and divides by the pixel count.
extern "C" int cdfo_metric_partials(const float* a, const float* b, int N, int H, int W, int crop, float scale, int clamp,
                                    int round8, int metric, double* partial, int partial_cap, int* nblocks_out,
                                    void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || crop < 0 || !a || !b || !partial || !nblocks_out) return CDFO_EINVAL;
  const int Hc = H - 2 * crop, Wc = W - 2 * crop;
  const int Ho = metric == 1 ? Hc - 10 : Hc, Wo = metric == 1 ? Wc - 10 : Wc;
  if (Ho <= 0 || Wo <= 0 || (metric != 0 && metric != 1)) return CDFO_EINVAL;
  long long blocks = ((long long)Ho * Wo + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if ((long long)N * blocks > partial_cap) return CDFO_EINVAL;
  *nblocks_out = (int)blocks;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (metric == 0)
    hipLaunchKernelGGL(sqdiff_kernel, dim3((unsigned)blocks, N), dim3(256), 0, st, a, b, H, W, crop, scale, clamp, round8, partial);
  else
    hipLaunchKernelGGL(ssim_kernel, dim3((unsigned)blocks, N), dim3(256), 0, st, a, b, H, W, crop, scale, clamp, round8, partial);
  CDFO_LAUNCH_CHECK();
  return 0;
}
