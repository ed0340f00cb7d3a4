// Pixel-local operators that only the reference's CVSR_V7 forward needs; fp32 pixel-major
// [B,H,W,64] activations like the rest of the library.  All HBM-bound, one pass over the data each.
//
//   cdfo_chan_pool      ChannelPool (arch/SIDECVSR_our.py:1883-1885): [max_c x, mean_c x] per pixel
//   cdfo_spatial_gate   SpatialAttention (arch.py:2719-2730): x * sigmoid(conv_kxk(pool(x)))  (7x7 in the feature extractor)
//   cdfo_rdab_mix       the mixing step of RDAB.forward (arch.py:2830-2847):
//                         x_f * ( softmax_c(v_max + Gumbel(u)) + sigmoid(conv3x3(pool(x_c))) )
//   cdfo_shrink_planes  F.interpolate(scale 0.5 / 0.25, bilinear, align_corners=False) / 2 resp. / 4 of the one- and
//                       two-channel priors (arch.py:4296-4303): the mean of the central 2x2 of each 2x2 / 4x4 block
//   cdfo_lincomb        out = ca*a + cb*b + cc*c  (the cross-scale sums of Block.forward, arch.py:367-375)
#include "common.h"

namespace {

// 16 lanes per pixel, one float4 of the 64 channels each
__global__ __launch_bounds__(256) void chan_pool_kernel(const float* __restrict__ x, int ld, long long npix,
                                                        float* __restrict__ out) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long p = gid >> 4;
  const int l = (int)(gid & 15);
  if (p >= npix) return;                      // whole 16-lane groups leave together (256 % 16 == 0)
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * ld + l * 4);
  float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
  float sm = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    sm += __shfl_xor(sm, o, 64);
  }
  if (l == 0) {
    out[p * 2] = mx;
    out[p * 2 + 1] = sm * (1.f / 64.f);
  }
}

__device__ __forceinline__ float pooled_conv(const float* __restrict__ pooled, const float* __restrict__ w, int H, int W,
                                             int y, int x, int ks) {
  // [2][ks][ks] kernel over the (max, mean) map, zero padding (ks-1)/2; nested loops: no per-tap integer division
  const int r = (ks - 1) / 2;
  float s = 0.f;
  for (int dy = 0; dy < ks; ++dy) {
    const int yy = y + dy - r;
    if (yy < 0 || yy >= H) continue;
    const float* row = pooled + (long long)yy * W * 2;
    const float* w0 = w + dy * ks;
    const float* w1 = w + ks * ks + dy * ks;
    for (int dx = 0; dx < ks; ++dx) {
      const int xx = x + dx - r;
      if (xx >= 0 && xx < W) {
        const float2 pv = *reinterpret_cast<const float2*>(row + xx * 2);
        s = fmaf(w0[dx], pv.x, fmaf(w1[dx], pv.y, s));
      }
    }
  }
  return s;
}

// gate[p] = sigmoid(conv_ksxks(pooled)[p] + bias): one thread per pixel (2 * ks * ks taps from the L2-resident pooled map)
__global__ __launch_bounds__(256) void gate_map_kernel(const float* __restrict__ pooled, const float* __restrict__ w,
                                                       const float* __restrict__ bias, int B, int H, int W, int ks,
                                                       float* __restrict__ gate) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x, npix = (long long)B * H * W;
  if (p >= npix) return;
  const int b = (int)(p / ((long long)H * W));
  const int rem = (int)(p - (long long)b * H * W), y = rem / W, x = rem - y * W;
  const float s = pooled_conv(pooled + (long long)b * H * W * 2, w, H, W, y, x, ks);
  gate[p] = 1.f / (1.f + __expf(-(s + bias[0])));
}

// The feature extractor applies ONE SpatialAttention (shared weights) to the same tensor once per round (arch.py:1350-1368):
// x_k = x_{k-1} * g_k,  g_k = sigmoid(conv(pool(x_{k-1}))).  Every g is positive, so x_k = x_0 * G_k with the per-pixel product
// G_k = g_1 ... g_k and pool(x_k) = G_k * pool(x_0) (max and mean commute with a positive per-pixel factor): the rounds only need
// pool(x_0) and the plane G -- the gated 64-channel tensors are never written.  cum_out[p] = G_k[p] from cum_in = G_{k-1} (null: 1).
__global__ __launch_bounds__(256) void gate_map_cum_kernel(const float* __restrict__ pooled, const float* __restrict__ cum_in,
                                                           const float* __restrict__ w, const float* __restrict__ bias, int B, int H,
                                                           int W, int ks, float* __restrict__ cum_out) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x, npix = (long long)B * H * W;
  if (p >= npix) return;
  const int b = (int)(p / ((long long)H * W));
  const int rem = (int)(p - (long long)b * H * W), y = rem / W, x = rem - y * W;
  const float* pl = pooled + (long long)b * H * W * 2;
  const float* cm = cum_in ? cum_in + (long long)b * H * W : nullptr;
  const int r = (ks - 1) / 2;
  float s = 0.f;
  for (int dy = 0; dy < ks; ++dy) {
    const int yy = y + dy - r;
    if (yy < 0 || yy >= H) continue;
    const float* w0 = w + dy * ks;
    const float* w1 = w + ks * ks + dy * ks;
    for (int dx = 0; dx < ks; ++dx) {
      const int xx = x + dx - r;
      if (xx >= 0 && xx < W) {
        const float2 pv = *reinterpret_cast<const float2*>(pl + ((long long)yy * W + xx) * 2);
        const float g = cm ? cm[(long long)yy * W + xx] : 1.f;
        s = fmaf(w0[dx], pv.x * g, fmaf(w1[dx], pv.y * g, s));
      }
    }
  }
  cum_out[p] = (cm ? cm[rem] : 1.f) * (1.f / (1.f + __expf(-(s + bias[0]))));
}

// out = x * gate[pixel]: 16 lanes per pixel, one float4 each -- a pure HBM stream
__global__ __launch_bounds__(256) void spatial_gate_kernel(const float* __restrict__ x, int ld, const float* __restrict__ gate,
                                                           long long npix, float* __restrict__ out, int ldo) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long p = gid >> 4;
  const int l = (int)(gid & 15);
  if (p >= npix) return;
  const float g = gate[p];
  f32x4 v = *reinterpret_cast<const f32x4*>(x + p * ld + l * 4);
  v[0] *= g; v[1] *= g; v[2] *= g; v[3] *= g;
  *reinterpret_cast<f32x4*>(out + p * ldo + l * 4) = v;
}

// One wave = 64 consecutive pixels of one image.  Phase 1 (lane = pixel): the 64 noise values of the pixel are read
// channel by channel (coalesced along x in the reference's NCHW noise tensor), e_c = exp(v_c - vmax) / (-log u_c)
// (== exp(v_c + g_c - vmax) with g = -log(-log u)) goes to LDS, the softmax denominator and the 3x3 spatial gate stay
// in registers -> LDS.  Phase 2 (16 lanes per pixel): x_f * (e / sum + att), written coalesced.
__global__ __launch_bounds__(128) void rdab_mix_kernel(const float* __restrict__ xf, int ld, const float* __restrict__ pooled,
                                                       const float* __restrict__ w3, const float* __restrict__ b3,
                                                       const float* __restrict__ v, const float* __restrict__ u, int B, int H,
                                                       int W, float* __restrict__ out, int ldo) {
  __shared__ float e_s[2][64][65];
  __shared__ float add_s[2][64], inv_s[2][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int P = H * W, tiles = (P + 63) / 64;
  const int tile = min(blockIdx.x * 2 + wave, B * tiles - 1);   // a surplus wave repeats the last tile (same values)
  const int b = tile / tiles, p0 = (tile - b * tiles) * 64;
  const int p = p0 + lane;
  const float* vb = v + b * 64;
  float vmax = vb[0];
  for (int c = 1; c < 64; ++c) vmax = fmaxf(vmax, vb[c]);
  if (p < P) {
    const float* ub = u + (long long)b * 64 * P + p;
    float sum = 0.f;
    for (int c = 0; c < 64; ++c) {
      const float e = __expf(vb[c] - vmax) / (-__logf(ub[(long long)c * P]));
      e_s[wave][c][lane] = e;
      sum += e;
    }
    const int y = p / W, x = p - y * W;
    const float s = pooled_conv(pooled + (long long)b * P * 2, w3, H, W, y, x, 3);
    add_s[wave][lane] = 1.f / (1.f + __expf(-(s + b3[0])));
    inv_s[wave][lane] = 1.f / sum;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int idx = i * 64 + lane, px = idx >> 4, c4 = (idx & 15) * 4;
    if (p0 + px >= P) continue;
    const long long q = (long long)b * P + p0 + px;
    f32x4 t = *reinterpret_cast<const f32x4*>(xf + q * ld + c4);
    const float a = add_s[wave][px], r = inv_s[wave][px];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] *= fmaf(e_s[wave][c4 + k][px], r, a);
    *reinterpret_cast<f32x4*>(out + q * ldo + c4) = t;
  }
}

__global__ __launch_bounds__(256) void shrink_planes_kernel(const float* __restrict__ src, long long bstride, int npl, int B,
                                                            int H, int W, int lv, float* __restrict__ dst) {
  const int Ho = H >> lv, Wo = W >> lv;
  const long long n = (long long)B * npl * Ho * Wo;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n) return;
  const int xo = (int)(gid % Wo), yo = (int)((gid / Wo) % Ho), pl = (int)((gid / ((long long)Wo * Ho)) % npl);
  const int b = (int)(gid / ((long long)Wo * Ho * npl));
  const int y0 = (yo << lv) + (lv == 2 ? 1 : 0), x0 = (xo << lv) + (lv == 2 ? 1 : 0);
  const float* s = src + (long long)b * bstride + (long long)pl * H * W;
  const float m = 0.25f * ((s[(long long)y0 * W + x0] + s[(long long)y0 * W + x0 + 1]) +
                           (s[(long long)(y0 + 1) * W + x0] + s[(long long)(y0 + 1) * W + x0 + 1]));
  dst[gid] = m * (lv == 2 ? 0.25f : 0.5f);
}

__global__ __launch_bounds__(256) void lincomb_kernel(float* out, const float* a, float ca, const float* b, float cb, const float* c,
                                                      float cc, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 r = reinterpret_cast<const f32x4*>(a)[i];
  r[0] *= ca; r[1] *= ca; r[2] *= ca; r[3] *= ca;
  if (b) {
    const f32x4 t = reinterpret_cast<const f32x4*>(b)[i];
    r[0] = fmaf(cb, t[0], r[0]); r[1] = fmaf(cb, t[1], r[1]); r[2] = fmaf(cb, t[2], r[2]); r[3] = fmaf(cb, t[3], r[3]);
  }
  if (c) {
    const f32x4 t = reinterpret_cast<const f32x4*>(c)[i];
    r[0] = fmaf(cc, t[0], r[0]); r[1] = fmaf(cc, t[1], r[1]); r[2] = fmaf(cc, t[2], r[2]); r[3] = fmaf(cc, t[3], r[3]);
  }
  reinterpret_cast<f32x4*>(out)[i] = r;
}

}  // namespace

extern "C" int cdfo_chan_pool(const float* x, int ld, long long npix, int C, float* out, void* stream) {
  if (C != 64 || npix <= 0 || ld < 64 || ld % 4) return CDFO_EINVAL;
  if (!aligned16(x)) return CDFO_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CdfoProfScope prof(st, KID_SPATIAL_GATE, 0.0, 4.0 * npix * 66.0);
  hipLaunchKernelGGL(chan_pool_kernel, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, st, x, ld, npix, out);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_spatial_gate(const float* x, int ld, const float* pooled, const float* w, const float* bias, int B, int H,
                                 int W, int C, int ks, float* gate_scratch, float* out, int ldo, void* stream) {
  if (C != 64 || B <= 0 || H <= 0 || W <= 0 || ks < 1 || !(ks & 1) || ld < 64 || ldo < 64 || ld % 4 || ldo % 4 || !gate_scratch)
    return CDFO_EINVAL;
  if (!aligned16(x) || !aligned16(out)) return CDFO_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long npix = (long long)B * H * W;
  CdfoProfScope prof(st, KID_SPATIAL_GATE, 2.0 * npix * 2 * ks * ks, 4.0 * npix * 132.0);
  hipLaunchKernelGGL(gate_map_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, pooled, w, bias, B, H, W, ks,
                     gate_scratch);
  hipLaunchKernelGGL(spatial_gate_kernel, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, st, x, ld, gate_scratch, npix,
                     out, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_gate_map_cumulative(const float* pooled, const float* cum_in, const float* w, const float* bias, int B, int H, int W,
                                        int ks, float* cum_out, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || ks < 1 || !(ks & 1) || !pooled || !w || !bias || !cum_out || cum_out == cum_in) return CDFO_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long npix = (long long)B * H * W;
  CdfoProfScope prof(st, KID_SPATIAL_GATE, 2.0 * npix * 2 * ks * ks, 4.0 * npix * 4.0);
  hipLaunchKernelGGL(gate_map_cum_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, pooled, cum_in, w, bias, B, H, W, ks,
                     cum_out);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_rdab_mix(const float* xf, int ld, const float* pooled, const float* w3, const float* b3, const float* vmax,
                             const float* noise, int B, int H, int W, float* out, int ldo, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || ld < 64 || ldo < 64 || ld % 4 || ldo % 4) return CDFO_EINVAL;
  if (!aligned16(xf) || !aligned16(out)) return CDFO_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long npix = (long long)B * H * W;
  const int tiles = B * ((H * W + 63) / 64);
  CdfoProfScope prof(st, KID_RDAB_PREP, 0.0, 4.0 * npix * (64.0 * 3 + 2));
  hipLaunchKernelGGL(rdab_mix_kernel, dim3((tiles + 1) / 2), dim3(128), 0, st, xf, ld, pooled, w3, b3, vmax, noise, B, H, W,
                     out, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_shrink_planes(const float* src, long long src_bstride, int planes, int B, int H, int W, int level,
                                  float* dst, void* stream) {
  if (B <= 0 || planes <= 0 || H <= 0 || W <= 0 || level < 1 || level > 2) return CDFO_EINVAL;
  if ((H | W) & ((1 << level) - 1)) return CDFO_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long n = (long long)B * planes * (H >> level) * (W >> level);
  CdfoProfScope prof(st, KID_RESAMPLE, 0.0, 4.0 * n * 5.0);
  hipLaunchKernelGGL(shrink_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, src_bstride, planes, B, H,
                     W, level, dst);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_lincomb(float* out, const float* a, float ca, const float* b, float cb, const float* c, float cc,
                            long long n, void* stream) {
  if (n <= 0 || n % 4 || !a || !out) return CDFO_EINVAL;
  if (!aligned16(out) || !aligned16(a) || (b && !aligned16(b)) || (c && !aligned16(c))) return CDFO_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CdfoProfScope prof(st, KID_SCALE, 0.0, 4.0 * n * (2.0 + (b != nullptr) + (c != nullptr)));
  hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, out, a, ca, b, cb, c, cc, n / 4);
  CDFO_LAUNCH_CHECK();
  return 0;
}
