// Deformable convolution for the reference's other two dtypes.
//
// The reference instantiates every DCN kernel for float, double and half (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// ops/dcn/src/deform_conv_cuda_kernel.cu:258,352,450,780,812,845).  float is the path everything else in this library is
// built for (dcn.hip, dcn_fast.hip, dcn_bwd.hip).  This file adds, behind one dtype-tagged pair of entry points:
//   * half:   fp16 tensors in and out, fp32 arithmetic -- the operands are widened into the workspace, the fp32 kernels
//             (including the fast split-fp16 forward) run on them, results are narrowed once (the reference rounds every
//             intermediate of its half instantiation to half; this is at least as accurate).  Accumulating gradient outputs
//             (grad_in, grad_weight, grad_bias) follow the reference's contract: the fp32 result is ADDED to the tensor;
//   * double: fp64 tensors, fp64 arithmetic in plain VALU kernels (gradient checking is what fp64 DCN is used for; these
//             are correctness-first: one 64-pixel tile per workgroup, 4-channel K chunks sampled into LDS, fp64 FMAs;
//             hardware fp64 atomics for the scattered gradients).
// Semantics (sampling rule, validity tests of forward / backward, offset and mask layouts, accumulate-vs-assign of the
// gradients) are those of dcn.hip / dcn_bwd.hip, i.e. deform_conv_cuda_kernel.cu:189-242,466-496,569-632 (forward) and
// :115-187,278-435,498-567,634-766 (backward).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ casts
template <typename S, typename D, bool ACC>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ s, D* __restrict__ d, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    if (ACC) d[i] = (D)((float)d[i] + (float)s[i]);
    else d[i] = (D)s[i];
  }
}
template <typename S, typename D, bool ACC>
void cast_launch(const S* s, D* d, long long n, hipStream_t st) {
  if (n <= 0) return;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((cast_kernel<S, D, ACC>), dim3((unsigned)blocks), dim3(256), 0, st, s, d, n);
}

// ------------------------------------------------------------------------------------------------ fp64 kernels
struct TArgs {
  const double* in; const double* offset; const double* mask; const double* weight; const double* bias; double* out;
  const double* gout; double* gin; double* goff; double* gmask; double* gw; double* gbias;
  int B, C, H, W, Co, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg;
  double scale;
};

__device__ __forceinline__ double bilinear64(const double* __restrict__ im, int H, int W, double h, double w) {
  const int hl = (int)floor(h), wl = (int)floor(w), hh_ = hl + 1, wh_ = wl + 1;
  const double lh = h - hl, lw = w - wl, hh = 1.0 - lh, hw = 1.0 - lw;
  double v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (hl >= 0 && wl >= 0) v1 = im[(long long)hl * W + wl];
  if (hl >= 0 && wh_ <= W - 1) v2 = im[(long long)hl * W + wh_];
  if (hh_ <= H - 1 && wl >= 0) v3 = im[(long long)hh_ * W + wl];
  if (hh_ <= H - 1 && wh_ <= W - 1) v4 = im[(long long)hh_ * W + wh_];
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

constexpr int TP = 64;        // pixels per workgroup
constexpr int TCC = 4;        // input channels per K chunk
constexpr int TOB = 16;       // output channels per thread and pass (4 waves x 16 = 64 per pass)

// forward: grid (ceil(P/64), conv groups, B), 256 threads = 64 pixels x 4 output-channel quarters
__global__ __launch_bounds__(256) void dcn64_fwd_kernel(TArgs a) {
  extern __shared__ __attribute__((aligned(16))) double col[];       // [TCC*T][64]
  const int tid = threadIdx.x, px = tid & 63, oq = tid >> 6;
  const int T = a.kh * a.kw, P = a.Ho * a.Wo, Cg = a.C / a.groups, Cog = a.Co / a.groups, Cdg = a.C / a.dg;
  const int b = blockIdx.z, g = blockIdx.y, p = blockIdx.x * TP + px;
  const bool pv = p < P;
  const int ho = pv ? p / a.Wo : 0, wo = pv ? p - ho * a.Wo : 0;
  for (int ob = 0; ob < Cog; ob += 4 * TOB) {                        // 64 output channels per pass
    double acc[TOB];
#pragma unroll
    for (int j = 0; j < TOB; ++j) acc[j] = 0.0;
    for (int c0 = 0; c0 < Cg; c0 += TCC) {
      const int ncc = min(TCC, Cg - c0);
      __syncthreads();
      for (int item = tid; item < ncc * T * TP; item += 256) {       // sample: item = (k = cc*T + t, pixel)
        const int i = item & (TP - 1), k = item >> 6, cc = k / T, t = k - cc * T;
        const int pp = blockIdx.x * TP + i;
        double v = 0.0;
        if (pp < P) {
          const int c = g * Cg + c0 + cc, d = c / Cdg, ki = t / a.kw, kj = t - ki * a.kw;
          const int h2 = pp / a.Wo, w2 = pp - h2 * a.Wo;
          const double* off = a.offset + ((long long)(b * a.dg + d) * 2 * T) * P;
          const double h_im = (double)(h2 * a.sh - a.ph + ki * a.dh) + off[(long long)(2 * t) * P + pp];
          const double w_im = (double)(w2 * a.sw - a.pw + kj * a.dw) + off[(long long)(2 * t + 1) * P + pp];
          if (h_im > -1 && w_im > -1 && h_im < a.H && w_im < a.W)
            v = bilinear64(a.in + ((long long)b * a.C + c) * a.H * a.W, a.H, a.W, h_im, w_im);
          if (a.mask) v *= a.mask[((long long)(b * a.dg + d) * T + t) * P + pp];
        }
        col[k * TP + i] = v;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < TOB; ++j) {
        const int o = ob + oq * TOB + j;
        if (o < Cog) {
          const double* wr = a.weight + ((long long)(g * Cog + o) * Cg + c0) * T;      // wave-uniform row
          double s = acc[j];
          for (int k = 0; k < ncc * T; ++k) s = fma(wr[k], col[k * TP + px], s);
          acc[j] = s;
        }
      }
    }
    if (pv) {
#pragma unroll
      for (int j = 0; j < TOB; ++j) {
        const int o = ob + oq * TOB + j;
        if (o < Cog) {
          const int oc = g * Cog + o;
          a.out[((long long)b * a.Co + oc) * P + p] = acc[j] + (a.bias ? a.bias[oc] : 0.0);
        }
      }
    }
  }
  (void)ho; (void)wo;
}

// backward w.r.t. input / offset / mask: one thread per (image, deformable group, tap, output pixel)
__global__ __launch_bounds__(256) void dcn64_bwd_data_kernel(TArgs a) {
  const int T = a.kh * a.kw, P = a.Ho * a.Wo, Cg = a.C / a.groups, Cog = a.Co / a.groups, Cdg = a.C / a.dg;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int dt = blockIdx.y, d = dt / T, t = dt - d * T, b = blockIdx.z;
  if (p >= P) return;
  const int ho = p / a.Wo, wo = p - ho * a.Wo, ki = t / a.kw, kj = t - ki * a.kw;
  const double* off = a.offset + ((long long)(b * a.dg + d) * 2 * T) * P;
  const double h_im = (double)(ho * a.sh - a.ph + ki * a.dh) + off[(long long)(2 * t) * P + p];
  const double w_im = (double)(wo * a.sw - a.pw + kj * a.dw) + off[(long long)(2 * t + 1) * P + p];
  const double m = a.mask ? a.mask[((long long)(b * a.dg + d) * T + t) * P + p] : 1.0;
  const bool valid = !(h_im <= -1 || w_im <= -1 || h_im >= a.H || w_im >= a.W);     // cu:143-150, 525-532
  double vh = 0, vw = 0, mv = 0;
  if (valid) {
    const int hl = (int)floor(h_im), wl = (int)floor(w_im), hhi = hl + 1, whi = wl + 1;
    const double lh = h_im - hl, lw = w_im - wl, hh = 1 - lh, hw = 1 - lw;
    const bool c1 = hl >= 0 && wl >= 0, c2 = hl >= 0 && whi <= a.W - 1, c3 = hhi <= a.H - 1 && wl >= 0,
               c4 = hhi <= a.H - 1 && whi <= a.W - 1;
    for (int cc = 0; cc < Cdg; ++cc) {
      const int c = d * Cdg + cc, g = c / Cg, cl = c - g * Cg;
      double cg = 0.0;                                                                 // (W^T x grad_out)[c, t, p]
      for (int o = 0; o < Cog; ++o)
        cg = fma(a.weight[((long long)(g * Cog + o) * Cg + cl) * T + t], a.gout[((long long)b * a.Co + g * Cog + o) * P + p], cg);
      const double* im = a.in + ((long long)b * a.C + c) * a.H * a.W;
      const double v1 = c1 ? im[(long long)hl * a.W + wl] : 0.0, v2 = c2 ? im[(long long)hl * a.W + whi] : 0.0,
                   v3 = c3 ? im[(long long)hhi * a.W + wl] : 0.0, v4 = c4 ? im[(long long)hhi * a.W + whi] : 0.0;
      mv += cg * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
      vh += (-hw * v1 - lw * v2 + hw * v3 + lw * v4) * cg * m;
      vw += (-hh * v1 + hh * v2 - lh * v3 + lh * v4) * cg * m;
      if (a.gin) {
        double* gi = a.gin + ((long long)b * a.C + c) * a.H * a.W;
        const double tg = cg * m;
        if (c1) atomicAdd(gi + (long long)hl * a.W + wl, hh * hw * tg);
        if (c2) atomicAdd(gi + (long long)hl * a.W + whi, hh * lw * tg);
        if (c3) atomicAdd(gi + (long long)hhi * a.W + wl, lh * hw * tg);
        if (c4) atomicAdd(gi + (long long)hhi * a.W + whi, lh * lw * tg);
      }
    }
  }
  if (a.goff) {
    a.goff[((long long)(b * a.dg + d) * 2 * T + 2 * t) * P + p] = vh;
    a.goff[((long long)(b * a.dg + d) * 2 * T + 2 * t + 1) * P + p] = vw;
  }
  if (a.gmask) a.gmask[((long long)(b * a.dg + d) * T + t) * P + p] = mv;
}

__device__ __forceinline__ double block_sum64(double v, double* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// grad_weight[oc][cl][t] += scale * sum_{b,p} grad_out[b][oc][p] * column[b][c][t][p]: one workgroup per weight element
__global__ __launch_bounds__(256) void dcn64_bwd_weight_kernel(TArgs a) {
  __shared__ double red[4];
  const int T = a.kh * a.kw, P = a.Ho * a.Wo, Cg = a.C / a.groups, Cog = a.Co / a.groups, Cdg = a.C / a.dg;
  const int t = blockIdx.x % T, cl = (blockIdx.x / T) % Cg, oc = blockIdx.x / (T * Cg);
  const int g = oc / Cog, c = g * Cg + cl, d = c / Cdg, ki = t / a.kw, kj = t - ki * a.kw;
  double s = 0.0;
  for (long long i = threadIdx.x; i < (long long)a.B * P; i += 256) {
    const int b = (int)(i / P), p = (int)(i - (long long)b * P), ho = p / a.Wo, wo = p - ho * a.Wo;
    const double* off = a.offset + ((long long)(b * a.dg + d) * 2 * T) * P;
    const double h_im = (double)(ho * a.sh - a.ph + ki * a.dh) + off[(long long)(2 * t) * P + p];
    const double w_im = (double)(wo * a.sw - a.pw + kj * a.dw) + off[(long long)(2 * t + 1) * P + p];
    double v = 0.0;
    if (h_im > -1 && w_im > -1 && h_im < a.H && w_im < a.W)
      v = bilinear64(a.in + ((long long)b * a.C + c) * a.H * a.W, a.H, a.W, h_im, w_im);
    if (a.mask) v *= a.mask[((long long)(b * a.dg + d) * T + t) * P + p];
    s = fma(a.gout[((long long)b * a.Co + oc) * P + p], v, s);
  }
  s = block_sum64(s, red);
  if (threadIdx.x == 0) a.gw[((long long)oc * Cg + cl) * T + t] += a.scale * s;
}

__global__ __launch_bounds__(256) void dcn64_bwd_bias_kernel(TArgs a) {
  __shared__ double red[4];
  const int P = a.Ho * a.Wo, oc = blockIdx.x;
  double s = 0.0;
  for (long long i = threadIdx.x; i < (long long)a.B * P; i += 256) {
    const int b = (int)(i / P), p = (int)(i - (long long)b * P);
    s += a.gout[((long long)b * a.Co + oc) * P + p];
  }
  s = block_sum64(s, red);
  if (threadIdx.x == 0) a.gbias[oc] += s;
}

struct Dims { int Ho, Wo, T; long long P, n_in, n_off, n_mask, n_w, n_out; };
bool dims_of(int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups,
             int dg, Dims& d) {
  if (B <= 0 || C <= 0 || Co <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 ||
      groups <= 0 || dg <= 0 || C % groups || Co % groups || C % dg)
    return false;
  d.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  d.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (d.Ho <= 0 || d.Wo <= 0) return false;
  d.T = kh * kw; d.P = (long long)d.Ho * d.Wo;
  if (d.P >= (1ll << 31) || (long long)H * W >= (1ll << 31)) return false;
  d.n_in = (long long)B * C * H * W; d.n_off = (long long)B * dg * 2 * d.T * d.P; d.n_mask = (long long)B * dg * d.T * d.P;
  d.n_w = (long long)Co * (C / groups) * d.T; d.n_out = (long long)B * Co * d.P;
  return true;
}
inline long long al256(long long n) { return (n + 255) / 256 * 256; }

}  // namespace

// Workspace (bytes, 256-byte aligned segments) of the typed entry points below.  dtype: CDFO_DTYPE_F32 -> what
// cdfo_dcn_forward wants for its fast path (0 for the backward), _F16 -> room for the fp32 copies of every operand and
// result plus that, _F64 -> 0.
extern "C" long long cdfo_dcn_workspace_bytes_dt(int dtype, int backward, int B, int C, int H, int W, int Co, int kh, int kw,
                                                 int sh, int sw, int ph, int pw, int dh, int dw, int groups,
                                                 int deformable_groups) {
  Dims d;
  if (!dims_of(B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, deformable_groups, d)) return -1;
  if (dtype == CDFO_DTYPE_F64) return 0;
  long long fast = backward ? cdfo_dcn_backward_workspace_bytes(B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, deformable_groups)
                            : cdfo_dcn_workspace_bytes(B, C, H, W, Co, kh, kw, groups, deformable_groups);
  if (fast < 0) return -1;
  if (!backward && fast < d.n_in * 4) fast = d.n_in * 4;             // the general kernel's group-planar copy
  if (dtype == CDFO_DTYPE_F32) return fast;
  if (dtype != CDFO_DTYPE_F16) return -1;
  long long n = al256(d.n_in * 4) + al256(d.n_off * 4) + al256(d.n_mask * 4) + al256(d.n_w * 4) + al256((long long)Co * 4) +
                al256(d.n_out * 4);
  if (backward) n += al256(d.n_in * 4) + al256(d.n_off * 4) + al256(d.n_mask * 4) + al256(d.n_w * 4) + al256((long long)Co * 4);
  return n + al256(fast);
}

// cdfo_dcn_forward for a tagged dtype (tensors as there, elements of `dtype`).
extern "C" int cdfo_dcn_forward_dt(int dtype, const void* in, const void* offset, const void* mask, const void* weight,
                                   const void* bias, void* out, int B, int C, int H, int W, int Co, int kh, int kw, int sh,
                                   int sw, int ph, int pw, int dh, int dw, int groups, int deformable_groups, void* workspace,
                                   long long workspace_bytes, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  Dims d;
  if (!dims_of(B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, deformable_groups, d)) return CDFO_EINVAL;
  if (!in || !offset || !weight || !out) return CDFO_EINVAL;
  if (dtype == CDFO_DTYPE_F32)
    return cdfo_dcn_forward(static_cast<const float*>(in), static_cast<const float*>(offset), static_cast<const float*>(mask),
                            static_cast<const float*>(weight), static_cast<const float*>(bias), static_cast<float*>(out), B, C, H,
                            W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, deformable_groups, workspace, workspace_bytes, stream);
  if (dtype == CDFO_DTYPE_F64) {
    TArgs a{};
    a.in = static_cast<const double*>(in); a.offset = static_cast<const double*>(offset);
    a.mask = static_cast<const double*>(mask); a.weight = static_cast<const double*>(weight);
    a.bias = static_cast<const double*>(bias); a.out = static_cast<double*>(out);
    a.B = B; a.C = C; a.H = H; a.W = W; a.Co = Co; a.Ho = d.Ho; a.Wo = d.Wo; a.kh = kh; a.kw = kw; a.sh = sh; a.sw = sw;
    a.ph = ph; a.pw = pw; a.dh = dh; a.dw = dw; a.groups = groups; a.dg = deformable_groups; a.scale = 1.0;
    const size_t lds = (size_t)TCC * d.T * TP * sizeof(double);
    if (lds > 64 * 1024) return CDFO_EINVAL;
    if (B > 65535 || groups > 65535) return CDFO_EINVAL;
    CdfoProfScope prof(st, KID_DCN, 2.0 * (double)B * d.P * Co * (C / groups) * d.T, 8.0 * (double)(d.n_in + d.n_off + d.n_mask + d.n_out));
    hipLaunchKernelGGL(dcn64_fwd_kernel, dim3((unsigned)((d.P + TP - 1) / TP), groups, B), dim3(256), lds, st, a);
    CDFO_LAUNCH_CHECK();
    return 0;
  }
  if (dtype != CDFO_DTYPE_F16) return CDFO_EINVAL;
  const long long need = cdfo_dcn_workspace_bytes_dt(dtype, 0, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, deformable_groups);
  if (!workspace || workspace_bytes < need || !aligned16(workspace)) return CDFO_EINVAL;
  char* ws = static_cast<char*>(workspace);
  auto take = [&](long long bytes) { char* p = ws; ws += al256(bytes); return reinterpret_cast<float*>(p); };
  float* f_in = take(d.n_in * 4); float* f_off = take(d.n_off * 4); float* f_mask = take(d.n_mask * 4);
  float* f_w = take(d.n_w * 4); float* f_b = take((long long)Co * 4); float* f_out = take(d.n_out * 4);
  typedef _Float16 h;
  cast_launch<h, float, false>(static_cast<const h*>(in), f_in, d.n_in, st);
  cast_launch<h, float, false>(static_cast<const h*>(offset), f_off, d.n_off, st);
  if (mask) cast_launch<h, float, false>(static_cast<const h*>(mask), f_mask, d.n_mask, st);
  cast_launch<h, float, false>(static_cast<const h*>(weight), f_w, d.n_w, st);
  if (bias) cast_launch<h, float, false>(static_cast<const h*>(bias), f_b, Co, st);
  CDFO_LAUNCH_CHECK();
  const int rc = cdfo_dcn_forward(f_in, f_off, mask ? f_mask : nullptr, f_w, bias ? f_b : nullptr, f_out, B, C, H, W, Co, kh, kw,
                                  sh, sw, ph, pw, dh, dw, groups, deformable_groups, ws,
                                  workspace_bytes - (ws - static_cast<char*>(workspace)), stream);
  if (rc) return rc;
  cast_launch<float, h, false>(f_out, static_cast<h*>(out), d.n_out, st);
  CDFO_LAUNCH_CHECK();
  return 0;
}

// cdfo_dcn_backward for a tagged dtype.  grad_in / grad_weight / grad_bias are accumulated into, grad_offset / grad_mask
// assigned, any of them may be NULL.
extern "C" int cdfo_dcn_backward_dt(int dtype, const void* in, const void* offset, const void* mask, const void* weight,
                                    const void* grad_out, void* grad_in, void* grad_offset, void* grad_mask, void* grad_weight,
                                    void* grad_bias, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph,
                                    int pw, int dh, int dw, int groups, int deformable_groups, float scale, void* workspace,
                                    long long workspace_bytes, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  Dims d;
  if (!dims_of(B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, deformable_groups, d)) return CDFO_EINVAL;
  if (!in || !offset || !weight || !grad_out || (grad_mask && !mask)) return CDFO_EINVAL;
  if (dtype == CDFO_DTYPE_F32)
    return cdfo_dcn_backward_ws(static_cast<const float*>(in), static_cast<const float*>(offset), static_cast<const float*>(mask),
                                static_cast<const float*>(weight), static_cast<const float*>(grad_out), static_cast<float*>(grad_in),
                                static_cast<float*>(grad_offset), static_cast<float*>(grad_mask), static_cast<float*>(grad_weight),
                                static_cast<float*>(grad_bias), B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups,
                                deformable_groups, scale, workspace, workspace_bytes, stream);
  if (dtype == CDFO_DTYPE_F64) {
    TArgs a{};
    a.in = static_cast<const double*>(in); a.offset = static_cast<const double*>(offset);
    a.mask = static_cast<const double*>(mask); a.weight = static_cast<const double*>(weight);
    a.gout = static_cast<const double*>(grad_out); a.gin = static_cast<double*>(grad_in);
    a.goff = static_cast<double*>(grad_offset); a.gmask = static_cast<double*>(grad_mask);
    a.gw = static_cast<double*>(grad_weight); a.gbias = static_cast<double*>(grad_bias);
    a.B = B; a.C = C; a.H = H; a.W = W; a.Co = Co; a.Ho = d.Ho; a.Wo = d.Wo; a.kh = kh; a.kw = kw; a.sh = sh; a.sw = sw;
    a.ph = ph; a.pw = pw; a.dh = dh; a.dw = dw; a.groups = groups; a.dg = deformable_groups; a.scale = (double)scale;
    if (B > 65535 || (long long)deformable_groups * d.T > 65535) return CDFO_EINVAL;
    CdfoProfScope prof(st, KID_DCN_BWD, 4.0 * (double)B * d.P * Co * (C / groups) * d.T, 8.0 * (double)(2 * d.n_in + 2 * d.n_off + 2 * d.n_mask + d.n_out));
    if (a.gin || a.goff || a.gmask)
      hipLaunchKernelGGL(dcn64_bwd_data_kernel, dim3((unsigned)((d.P + 255) / 256), deformable_groups * d.T, B), dim3(256), 0, st, a);
    if (a.gw) {
      if (d.n_w >= (1ll << 31)) return CDFO_EINVAL;
      hipLaunchKernelGGL(dcn64_bwd_weight_kernel, dim3((unsigned)d.n_w), dim3(256), 0, st, a);
    }
    if (a.gbias) hipLaunchKernelGGL(dcn64_bwd_bias_kernel, dim3(Co), dim3(256), 0, st, a);
    CDFO_LAUNCH_CHECK();
    return 0;
  }
  if (dtype != CDFO_DTYPE_F16) return CDFO_EINVAL;
  const long long need = cdfo_dcn_workspace_bytes_dt(dtype, 1, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, deformable_groups);
  if (!workspace || workspace_bytes < need || !aligned16(workspace)) return CDFO_EINVAL;
  char* ws = static_cast<char*>(workspace);
  auto take = [&](long long bytes) { char* p = ws; ws += al256(bytes); return reinterpret_cast<float*>(p); };
  float* f_in = take(d.n_in * 4); float* f_off = take(d.n_off * 4); float* f_mask = take(d.n_mask * 4);
  float* f_w = take(d.n_w * 4); float* f_b = take((long long)Co * 4); float* f_go = take(d.n_out * 4);
  float* g_in = take(d.n_in * 4); float* g_off = take(d.n_off * 4); float* g_mask = take(d.n_mask * 4);
  float* g_w = take(d.n_w * 4); float* g_b = take((long long)Co * 4);
  (void)f_b;
  typedef _Float16 h;
  cast_launch<h, float, false>(static_cast<const h*>(in), f_in, d.n_in, st);
  cast_launch<h, float, false>(static_cast<const h*>(offset), f_off, d.n_off, st);
  if (mask) cast_launch<h, float, false>(static_cast<const h*>(mask), f_mask, d.n_mask, st);
  cast_launch<h, float, false>(static_cast<const h*>(weight), f_w, d.n_w, st);
  cast_launch<h, float, false>(static_cast<const h*>(grad_out), f_go, d.n_out, st);
  if (grad_in && hipMemsetAsync(g_in, 0, d.n_in * 4, st) != hipSuccess) return CDFO_EINVAL;
  if (grad_weight && hipMemsetAsync(g_w, 0, d.n_w * 4, st) != hipSuccess) return CDFO_EINVAL;
  if (grad_bias && hipMemsetAsync(g_b, 0, (size_t)Co * 4, st) != hipSuccess) return CDFO_EINVAL;
  CDFO_LAUNCH_CHECK();
  const int rc = cdfo_dcn_backward_ws(f_in, f_off, mask ? f_mask : nullptr, f_w, f_go, grad_in ? g_in : nullptr,
                                      grad_offset ? g_off : nullptr, grad_mask ? g_mask : nullptr, grad_weight ? g_w : nullptr,
                                      grad_bias ? g_b : nullptr, B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups,
                                      deformable_groups, scale, ws, workspace_bytes - (ws - static_cast<char*>(workspace)), stream);
  if (rc) return rc;
  if (grad_in) cast_launch<float, h, true>(g_in, static_cast<h*>(grad_in), d.n_in, st);
  if (grad_offset) cast_launch<float, h, false>(g_off, static_cast<h*>(grad_offset), d.n_off, st);
  if (grad_mask) cast_launch<float, h, false>(g_mask, static_cast<h*>(grad_mask), d.n_mask, st);
  if (grad_weight) cast_launch<float, h, true>(g_w, static_cast<h*>(grad_weight), d.n_w, st);
  if (grad_bias) cast_launch<float, h, true>(g_b, static_cast<h*>(grad_bias), Co, st);
  CDFO_LAUNCH_CHECK();
  return 0;
}
