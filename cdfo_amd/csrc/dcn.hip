// Fused deformable convolution forward (DCNv1 and modulated DCNv2) for gfx950.
//
// Replaces, in ONE kernel and without the `columns` [C*kh*kw, Ho*Wo] HBM round trip, what the reference does with
//   deformable_im2col_gpu_kernel / modulated_deformable_im2col_gpu_kernel   (ops/dcn/src/deform_conv_cuda_kernel.cu:189-242, 569-632)
//   + dmcn_im2col_bilinear (cu:466-496) + per-image, per-group addmm_ and bias add (ops/dcn/src/deform_conv_cuda.cpp:545-563, 151-258).
//
// Same tensor contract as the reference operator: NCHW input [B,C,H,W], offset [B, 2*dg*kh*kw, Ho, Wo] with (h,w)
// interleaved per tap, mask [B, dg*kh*kw, Ho, Wo] (DCNv2 only), OIHW weight [Co, C/groups, kh, kw], NCHW output.
//
// One workgroup = 64 consecutive output pixels of one image x up to 256 output channels of one conv group.
// Per chunk of 4 input channels: 256 threads bilinearly sample the 4*kh*kw modulated column values of the 64 pixels
// straight into LDS (offset/mask reads are coalesced along the pixel axis; the 4-corner gathers hit L1/L2), the
// matching weight rows are staged transposed, and the contraction runs on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: M = output channel, N = pixel), so the output tile is written coalesced in NCHW.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int CC = 4;      // input channels per K chunk
constexpr int PIXT = 64;   // output pixels per workgroup
constexpr int MAXJ = 4;    // up to 4 x 64 output channels per workgroup
constexpr int WSTR = 257;  // LDS row pitch of the staged weights: odd, so that the transposing store (lanes = consecutive k) is conflict-free

struct DcnArgs {
  const float* in; const float* offset; const float* mask; const float* w; const float* bias; float* out;
  const float* gp;   // optional group-planar copy of `in`: [B][dg][H][W][C/dg] (see dcn_to_gp_kernel)
  const unsigned* run_if;   // optional device word: the kernel returns at once while it is 0 (re-run request of the fast path)
  int B, C, H, W, Co, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg;
};

__device__ __forceinline__ float bilinear_zero(const float* __restrict__ plane, int H, int W, float h, float w) {
  // dmcn_im2col_bilinear (cu:466-496): corners outside the image contribute 0
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h - (float)h_low, lw = w - (float)w_low;
  const float hh = 1.f - lh, hw = 1.f - lw;
  float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
  if (h_low >= 0 && w_low >= 0) v1 = plane[h_low * W + w_low];
  if (h_low >= 0 && w_high <= W - 1) v2 = plane[h_low * W + w_high];
  if (h_high <= H - 1 && w_low >= 0) v3 = plane[h_high * W + w_low];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = plane[h_high * W + w_high];
  return (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4;
}

// in NCHW -> group-planar [B][dg][H][W][C/dg]: the C/dg channels that share one sample position become contiguous, so
// a bilinear corner of a 4-channel chunk is ONE 16-byte gather instead of four 4-byte ones
// amax (optional): bits of max |in| over the finite elements, by atomicMax on the bit pattern (the fast path's input scale)
__global__ __launch_bounds__(256) void dcn_to_gp_kernel(const float* __restrict__ in, float* __restrict__ gp, int B, int C,
                                                        int dg, long long HW, unsigned* __restrict__ amax) {
  const int Cdg = C / dg;
  const long long n = (long long)B * dg * HW;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  float m = 0.f;
  if (gid < n) {
    const long long p = gid % HW, bd = gid / HW;                  // bd = b*dg + d
    const float* src = in + bd * Cdg * HW + p;
    float* dst = gp + gid * Cdg;
    for (int c = 0; c < Cdg; c += 4) {
      f32x4 v;
      v[0] = src[(long long)c * HW]; v[1] = src[(long long)(c + 1) * HW];
      v[2] = src[(long long)(c + 2) * HW]; v[3] = src[(long long)(c + 3) * HW];
      *reinterpret_cast<f32x4*>(dst + c) = v;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float av = fabsf(v[e]); m = av < 3.0e38f ? fmaxf(m, av) : m; }
    }
  }
  if (amax) {
    // one atomic per wave at most, and only while it would still raise the maximum (a plain read first: the running
    // maximum settles within the first workgroups, after which nobody touches the contended word any more)
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f && __float_as_uint(m) > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(amax, __float_as_uint(m));
  }
}

template <bool GP>
__global__ __launch_bounds__(256) void dcn_fwd_kernel(DcnArgs a, int KCH /* = CC*kh*kw rounded up to even */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* col = smem;               // [KCH][64]
  float* wt = smem + KCH * PIXT;   // [KCH][WSTR]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, r = lane & 31;
  const int T = a.kh * a.kw;
  const int Cg = a.C / a.groups, Cog = a.Co / a.groups, Cdg = a.C / a.dg;
  const int P = a.Ho * a.Wo;
  const int b = blockIdx.z, g = blockIdx.y / ((Cog + 255) / 256), coblk = blockIdx.y % ((Cog + 255) / 256);
  const int co0 = coblk * 256;                                   // first output channel (within the group)
  const int nco = (Cog - co0) < 256 ? (Cog - co0) : 256;
  const int p0 = blockIdx.x * PIXT;
  const int mt = wave >> 1, nt = wave & 1;                       // this wave's 32-cout / 32-pixel sub-tile
  if (a.run_if && __hip_atomic_load(a.run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;   // workgroup-uniform

  f32x16 acc[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  const float* in_b = a.in + (long long)b * a.C * a.H * a.W;
  const float* off_b = a.offset + (long long)b * a.dg * 2 * T * P;
  const float* msk_b = a.mask ? a.mask + (long long)b * a.dg * T * P : nullptr;

  for (int c0 = 0; c0 < Cg; c0 += CC) {
    __syncthreads();
    if constexpr (GP) {
      // ---- sample: item = (tap t, pixel i); the chunk's 4 channels share the position (same deformable group)
      const int c = g * Cg + c0, d = c / Cdg, cin = c - d * Cdg;
      const float* gp_b = a.gp + ((long long)(b * a.dg + d) * a.H * a.W) * Cdg + cin;
      for (int item = tid; item < T * PIXT; item += 256) {
        const int i = item & (PIXT - 1), t = item >> 6;
        const int p = p0 + i;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (p < P) {
          const int ho = p / a.Wo, wo = p - ho * a.Wo;
          const int ki = t / a.kw, kj = t - ki * a.kw;
          const float oh = off_b[((long long)(d * T + t) * 2) * P + p];
          const float ow = off_b[((long long)(d * T + t) * 2 + 1) * P + p];
          const float h_im = (float)(ho * a.sh - a.ph + ki * a.dh) + oh;
          const float w_im = (float)(wo * a.sw - a.pw + kj * a.dw) + ow;
          if (h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W) {
            const int hl = (int)floorf(h_im), wl = (int)floorf(w_im), hh_ = hl + 1, wh_ = wl + 1;
            const float lh = h_im - (float)hl, lw = w_im - (float)wl, hh = 1.f - lh, hw = 1.f - lw;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 v1 = (hl >= 0 && wl >= 0) ? *reinterpret_cast<const f32x4*>(gp_b + ((long long)hl * a.W + wl) * Cdg) : z;
            const f32x4 v2 = (hl >= 0 && wh_ <= a.W - 1) ? *reinterpret_cast<const f32x4*>(gp_b + ((long long)hl * a.W + wh_) * Cdg) : z;
            const f32x4 v3 = (hh_ <= a.H - 1 && wl >= 0) ? *reinterpret_cast<const f32x4*>(gp_b + ((long long)hh_ * a.W + wl) * Cdg) : z;
            const f32x4 v4 = (hh_ <= a.H - 1 && wh_ <= a.W - 1) ? *reinterpret_cast<const f32x4*>(gp_b + ((long long)hh_ * a.W + wh_) * Cdg) : z;
            const float m = msk_b ? msk_b[(long long)(d * T + t) * P + p] : 1.f;
            const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // same expression order as bilinear_zero(), then the mask
              const float sv = w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e];
              val[e] = msk_b ? sv * m : sv;
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c0 + e < Cg) col[(e * T + t) * PIXT + i] = val[e];
      }
      if (KCH > CC * T)                                           // the odd padding row of the K = 2 MFMA step
        for (int i = tid; i < PIXT; i += 256) col[CC * T * PIXT + i] = 0.f;
    } else {
      // ---- sample: item = (k = cc*T + t, pixel i)
      for (int item = tid; item < KCH * PIXT; item += 256) {
        const int i = item & (PIXT - 1), k = item >> 6;
        const int cc = k / T, t = k - cc * T;
        const int p = p0 + i;
        float val = 0.f;
        if (cc < CC && c0 + cc < Cg && p < P) {
          const int c = g * Cg + c0 + cc;                         // absolute input channel
          const int d = c / Cdg;                                   // deformable group
          const int ho = p / a.Wo, wo = p - ho * a.Wo;
          const int ki = t / a.kw, kj = t - ki * a.kw;
          const float oh = off_b[((long long)(d * T + t) * 2) * P + p];
          const float ow = off_b[((long long)(d * T + t) * 2 + 1) * P + p];
          const float h_im = (float)(ho * a.sh - a.ph + ki * a.dh) + oh;
          const float w_im = (float)(wo * a.sw - a.pw + kj * a.dw) + ow;
          if (h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W) {
            val = bilinear_zero(in_b + (long long)c * a.H * a.W, a.H, a.W, h_im, w_im);
            if (msk_b) val *= msk_b[(long long)(d * T + t) * P + p];
          }
        }
        col[k * PIXT + i] = val;
      }
    }
    // ---- weights of this chunk, transposed: wt[k][o] = W[g*Cog + co0 + o][c0*T + k]
    for (int item = tid; item < KCH * ((nco + 63) & ~63); item += 256) {   // only the 64-channel groups the MFMAs read
      const int k = item % KCH, o = item / KCH;
      float v = 0.f;
      if (o < nco && k < CC * T && c0 * T + k < Cg * T) v = a.w[((long long)(g * Cog + co0 + o) * Cg + c0) * T + k];
      wt[k * WSTR + o] = v;
    }
    __syncthreads();
    for (int k = 0; k < KCH; k += 2) {
      const float bv = col[(k + h) * PIXT + nt * 32 + r];
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        if (j * 64 < nco) {
          const float av = wt[(k + h) * WSTR + j * 64 + mt * 32 + r];
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
        }
      }
    }
  }
  // ---- store: D[row = cout][col = pixel]
  const int p = p0 + nt * 32 + r;
  if (p < P) {
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      if (j * 64 >= nco) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int o = j * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (o < nco) {
          const int oc = g * Cog + co0 + o;
          a.out[((long long)b * a.Co + oc) * P + p] = acc[j][e] + (a.bias ? a.bias[oc] : 0.f);
        }
      }
    }
  }
}

void launch_to_gp(const float* in, float* gp, int B, int C, int dg, long long HW, unsigned* amax, hipStream_t st) {
  const long long n = (long long)B * dg * HW;
  hipLaunchKernelGGL(dcn_to_gp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, gp, B, C, dg, HW, amax);
}

}  // namespace

// dcn_fast.hip
int cdfo_dcn_forward_fast(const float* in, const float* offset, const float* mask, const float* weight, const float* bias,
                          float* out, int B, int C, int H, int W, int Co, int Ho, int Wo, int kh, int kw, int sh, int sw, int ph,
                          int pw, int dh, int dw, int groups, int dg, void* workspace, long long workspace_bytes,
                          hipStream_t st, void (*to_gp)(const float*, float*, int, int, int, long long, unsigned*, hipStream_t),
                          const unsigned** rerun_flag);

// dcn_win.hip
int cdfo_dcn_forward_win(const float* in, const float* offset, const float* mask, const float* weight, const float* bias,
                         float* out, int B, int C, int H, int W, int Co, int Ho, int Wo, int kh, int kw, int sh, int sw, int ph,
                         int pw, int dh, int dw, int groups, int dg, void* workspace, long long workspace_bytes, hipStream_t st,
                         const unsigned** rerun_flag);

extern "C" int cdfo_dcn_forward(const float* in, const float* offset, const float* mask, const float* weight,
                                const float* bias, float* out, int B, int C, int H, int W, int Co, int kh, int kw, int sh,
                                int sw, int ph, int pw, int dh, int dw, int groups, int deformable_groups, void* workspace,
                                long long workspace_bytes, void* stream) {
  if (B <= 0 || C <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || groups <= 0 ||
      deformable_groups <= 0)
    return CDFO_EINVAL;
  if (C % groups || Co % groups || C % deformable_groups) return CDFO_EINVAL;
  const int Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  const int Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (Ho <= 0 || Wo <= 0) return CDFO_EINVAL;
  DcnArgs a{in, offset, mask, weight, bias, out, nullptr, nullptr, B, C, H, W, Co, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, groups,
            deformable_groups};
  // group-planar gathers need: a caller-provided scratch copy of `in`, 4-channel chunks that never straddle a
  // deformable group or a conv group, 16-byte aligned rows
  const bool gp = workspace && workspace_bytes >= (long long)B * C * H * W * 4 && aligned16(workspace) &&
                  (C / deformable_groups) % 4 == 0 && (C / groups) % 4 == 0;
  const int T = kh * kw;
  const int KCH = (CC * T + 1) / 2 * 2;
  const size_t lds = (size_t)KCH * (PIXT + WSTR) * sizeof(float);
  if (lds > 160 * 1024) return CDFO_EINVAL;
  static CdfoAttrGrow grow_a, grow_b;
  if (cdfo_grow_max_lds(grow_a, reinterpret_cast<const void*>(&dcn_fwd_kernel<false>), (int)lds) != hipSuccess ||
      cdfo_grow_max_lds(grow_b, reinterpret_cast<const void*>(&dcn_fwd_kernel<true>), (int)lds) != hipSuccess)
    return CDFO_EINVAL;
  const int Cog = Co / groups;
  dim3 grid(cdiv(Ho * Wo, PIXT), groups * cdiv(Cog, 256), B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double px = (double)B * Ho * Wo;
  CdfoProfScope prof(st, KID_DCN, 2.0 * px * Co * (C / groups) * T,
                     4.0 * (px * (Co + 3.0 * deformable_groups * T) + (double)B * C * H * W + (double)Co * (C / groups) * T));
  const unsigned* rerun = nullptr;
  // (1) window-sampled kernel (dcn_win.hip): 3x3 / stride 1 / dilation 1 shapes of the alignment module's kind
  static const int use_win = []() { const char* e = getenv("CDFO_DCN_WIN"); return e ? atoi(e) : 1; }();     // developer A/B switch
  const int win = use_win ? cdfo_dcn_forward_win(in, offset, mask, weight, bias, out, B, C, H, W, Co, Ho, Wo, kh, kw, sh, sw, ph, pw,
                                                 dh, dw, groups, deformable_groups, workspace, workspace_bytes, st, &rerun) : 0;
  if (win > 1) return win - 2;
  if (win == 1) {      // range-safety re-run as below, from the NCHW input (this path makes no group-planar copy)
    a.run_if = rerun;
    hipLaunchKernelGGL(dcn_fwd_kernel<false>, grid, dim3(256), lds, st, a, KCH);
    CDFO_LAUNCH_CHECK();
    return 0;
  }
  const int fast = cdfo_dcn_forward_fast(in, offset, mask, weight, bias, out, B, C, H, W, Co, Ho, Wo, kh, kw, sh, sw, ph, pw, dh,
                                         dw, groups, deformable_groups, workspace, workspace_bytes, st, &launch_to_gp, &rerun);
  if (fast > 1) return fast - 2;
  if (fast == 1) {
    // The fast kernel holds the sampled values as scaled fp16 hi + lo.  Where a sample left that range (or was not finite)
    // it raised `rerun`: the exact-fp32 kernel then recomputes the whole result; otherwise its workgroups return at once.
    a.gp = static_cast<const float*>(workspace);
    a.run_if = rerun;
    hipLaunchKernelGGL(dcn_fwd_kernel<true>, grid, dim3(256), lds, st, a, KCH);
    CDFO_LAUNCH_CHECK();
    return 0;
  }
  if (gp) {
    a.gp = static_cast<const float*>(workspace);
    launch_to_gp(in, static_cast<float*>(workspace), B, C, deformable_groups, (long long)H * W, nullptr, st);
    hipLaunchKernelGGL(dcn_fwd_kernel<true>, grid, dim3(256), lds, st, a, KCH);
  } else {
    hipLaunchKernelGGL(dcn_fwd_kernel<false>, grid, dim3(256), lds, st, a, KCH);
  }
  CDFO_LAUNCH_CHECK();
  return 0;
}
