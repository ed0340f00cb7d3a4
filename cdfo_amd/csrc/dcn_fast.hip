// Fast path of the fused deformable convolution forward (see dcn.hip for the operator contract and the general kernel).
//
// Applies when the caller hands cdfo_dcn_forward a workspace and: groups == 1, (C / deformable_groups) % 4 == 0,
// Co % 32 == 0, Co <= 128, kh*kw <= 64 -- the alignment module's shape (C = Co = 64, dg = 16, 3x3; arch.py:4242).
//
// Design (HBM roofline: (C + 3*dg*kh*kw + Co) * 4 bytes per output pixel; what actually limits it is the VALU work of
// the C*kh*kw bilinear samples per pixel):
//   * prepass 1: `in` NCHW -> group-planar [B][dg][H][W][C/dg] in the workspace, so the four corners of a sample are
//     16-byte gathers serving four channels at once (dcn.hip: dcn_to_gp_kernel);
//   * prepass 2: weights -> fp16 hi | lo halves (scaled by a power of two so that |w| <= 16 keeps 22 mantissa bits),
//     stored in the exact per-lane order of the v_mfma_f32_32x32x16_f16 A operand, K index = (channel block, tap,
//     channel-in-block): a wave's operand load is one coalesced 1 KiB read;
//   * main kernel: one 512-thread workgroup = 64 consecutive output pixels x all output channels.  Phase 1: all waves
//     sample -- thread = (pixel, slice of the (4-channel block, tap) items); offsets / mask are read coalesced along the
//     pixel axis and every sampled 4-vector is split into fp16 hi + lo and written with two 8-byte LDS stores into the
//     [pixel][K] images of the B operand (row pitch K*2 + 16 bytes: conflict-free ds_read_b128).  Phase 2: the 8 waves
//     = 2 pixel halves x 2 output-channel halves x 2 K halves run hi*hi + lo*hi + hi*lo (3 MFMA passes, fp32
//     accumulation, ~2^-22 relative); the two K halves are summed through LDS and the tile is stored NCHW, coalesced.
//     The whole K = C*kh*kw (<= 576 rows per chunk, more channels loop) lives in LDS: one barrier pair per 64 pixels
//     instead of one per 4 input channels.
// Sampled values are held as fp16 hi + lo.  Range safety: the group-planar prepass also finds max |in|, the sampler
// multiplies by the power of two that puts it in (8, 16] (undone with the weight scale at the end), so small- and
// large-magnitude feature maps keep their 22 bits; a sample whose scaled value would still leave the fp16 range
// (|in * mask| > 3750 max |in|, i.e. a mask far outside [0, 1]) or is not finite raises a device flag, and
// cdfo_dcn_forward then re-runs the exact-fp32 kernel of dcn.hip over the whole result (its workgroups return at once
// while the flag is clear).  Callers who want the exact kernel unconditionally pass workspace = NULL.
#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KMAX = 288;      // K rows (input channels x taps) resident in LDS per chunk: 2 x 37 KiB images -> two workgroups per CU

struct FastArgs {
  const float* gp; const float* offset; const float* mask; const float* bias; float* out;
  const h8* wp_hi; const h8* wp_lo;
  unsigned* wmax;    // [0] = bits of max |w|, [1] = bits of max |in| (-> the power-of-two scales), [2] = re-run request
  int B, C, H, W, Co, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, dg;
  int CCH;       // input channels per K chunk (multiple of 4)
  int nchunks;   // ceil(C / CCH)
  int S;         // 16-row K steps per full chunk = ceil(CCH * kh*kw / 16)
};

// max |w| over the weight tensor -> wmax[0] (as the bit pattern of a non-negative float: integer max == float max)
__global__ __launch_bounds__(256) void dcn_wmax_kernel(const float* __restrict__ w, long long n, unsigned* __restrict__ wmax) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m < 3.0e38f) atomicMax(wmax, __float_as_uint(m));
}

__device__ __forceinline__ float pow2_scale(unsigned max_bits) {
  const float m = __uint_as_float(max_bits);
  // a power of two: m * s in (8, 16]; the exponent is clamped so that neither s nor 1 / s leaves the fp32 range
  return m > 0.f ? exp2f(fminf(fmaxf(4.f - ceilf(log2f(m)), -100.f), 100.f)) : 1.f;
}
__device__ __forceinline__ float weight_scale(const unsigned* wmax) { return pow2_scale(wmax[0]); }

// one thread per fp16 element of the packed A operand: index = (((chunk*S + s)*MT + mtile)*64 + lane)*8 + j
__global__ __launch_bounds__(256) void dcn_wpack_kernel(const float* __restrict__ w, const unsigned* __restrict__ wmax,
                                                        _Float16* __restrict__ hi, _Float16* __restrict__ lo, int C, int Co,
                                                        int T, int CCH, int nchunks, int S) {
  const int MT = Co / 32;
  const long long n = (long long)nchunks * S * MT * 64 * 8;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n) return;
  const int j = (int)(gid & 7), lane = (int)((gid >> 3) & 63);
  const long long r = gid >> 9;
  const int mtile = (int)(r % MT), s = (int)((r / MT) % S), chunk = (int)(r / ((long long)MT * S));
  const int c0 = chunk * CCH, nch = (C - c0) < CCH ? (C - c0) : CCH;
  const int k = s * 16 + 8 * (lane >> 5) + j;                          // K row within the chunk
  float v = 0.f;
  if (k < nch * T) {
    const int q = k >> 2, cc = k & 3, cb = q / T, t = q - cb * T;
    const int m = mtile * 32 + (lane & 31), c = c0 + cb * 4 + cc;
    v = w[((long long)m * C + c) * T + t] * weight_scale(wmax);
  }
  const _Float16 h = (_Float16)v;
  hi[gid] = h;
  lo[gid] = (_Float16)(v - (float)h);
}

template <int MJ>   // output-channel tiles (of 32) per wave: 1 -> Co <= 64, 2 -> Co <= 128
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void dcn_fast_kernel(FastArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float tabh[64], tabw[64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int T = a.kh * a.kw, P = a.Ho * a.Wo, Cdg = a.C / a.dg, MT = a.Co / 32;
  const int ROWB = a.S * 32 + 16;                                   // bytes per pixel row of the B-operand images
  char* col_hi = smem;
  char* col_lo = smem + 64 * ROWB;
  // Workgroups go round-robin to the 8 XCDs (private 4 MB L2s): each XCD walks its own contiguous band of tiles, so that the
  // ~64 tiles it has in flight (8.5 image rows at W = 480) gather from ~20 rows of the group-planar copy (2.4 MB) instead of
  // from the whole image (33 MB): the corner gathers then hit L2 (r1 counters: 0.95 GB per launch re-fetched past it)
  const int band = gridDim.x >> 3;                                   // gridDim.x is a multiple of 8
  const int tile = (blockIdx.x & 7) * band + (blockIdx.x >> 3);
  const int b = blockIdx.y, p0 = tile * 64;
  if (p0 >= P) return;                                               // (whole workgroup)
  if (tid < T) {
    const int ki = tid / a.kw, kj = tid - ki * a.kw;
    tabh[tid] = (float)(ki * a.dh);
    tabw[tid] = (float)(kj * a.dw);
  }
  // sampling role: thread = (pixel, slice)
  const int px = tid & 63, slice = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: item bookkeeping stays scalar
  const int p = p0 + px;
  const bool pvalid = p < P;
  const int ho = pvalid ? p / a.Wo : 0, wo = pvalid ? p - ho * a.Wo : 0;
  const float hb = (float)(ho * a.sh - a.ph), wb = (float)(wo * a.sw - a.pw);
  const float* off_b = a.offset + (long long)b * a.dg * 2 * T * P + p;
  const float* msk_b = a.mask ? a.mask + (long long)b * a.dg * T * P + p : nullptr;
  const float* gp_b = a.gp + (long long)b * a.C * a.H * a.W;
  const float s_in = pow2_scale(a.wmax[1]);
  unsigned ovfbits = 0;
  // matrix role: wave = (pixel half, output-channel half, K half)
  const int nt = wave & 1, mh = (wave >> 1) & 1, kh2 = wave >> 2;
  f32x16 acc[MJ];
#pragma unroll
  for (int j = 0; j < MJ; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  for (int chunk = 0; chunk < a.nchunks; ++chunk) {
    const int c0 = chunk * a.CCH, nch = (a.C - c0) < a.CCH ? (a.C - c0) : a.CCH;
    const int NI = (nch >> 2) * T, KC = NI * 4, S = (KC + 15) >> 4;
    __syncthreads();                                                 // tables written / previous chunk's operands consumed
    if (KC < S * 16) {                                               // zero the K padding of the last 16-row step
      const int padn = S * 16 - KC;
      for (int idx = tid; idx < 64 * padn; idx += 512) {
        const int pp = idx & 63, kk = KC + (idx >> 6);
        *reinterpret_cast<_Float16*>(col_hi + pp * ROWB + kk * 2) = (_Float16)0.f;
        *reinterpret_cast<_Float16*>(col_lo + pp * ROWB + kk * 2) = (_Float16)0.f;
      }
    }
    // ---- phase 1: sample.  item q = (4-channel block cb, tap t) -> K rows q*4 .. q*4+3.  Items are taken NB at a time and
    // every stage (offset / mask loads, the 4 x NB corner gathers, combine + store) is unrolled over the batch, so that a
    // stage's loads are all in flight together: the dependent chain offset -> gather costs two memory latencies per
    // BATCH, not per item; the next batch's offsets are requested right behind the current batch's gathers
    {
      constexpr int NB = 3;
      int cb0 = slice / T, t0 = slice - cb0 * T;
      float noh[NB], now_[NB], nmm[NB];                          // the NEXT batch's offsets / mask, loaded one batch ahead
      int ntt[NB], ncb[NB];
      bool nlive[NB];
      auto load_batch = [&](int q0) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          ntt[i] = t0; ncb[i] = cb0;
          nlive[i] = pvalid && (q0 + 8 * i) < NI;
          noh[i] = now_[i] = 0.f; nmm[i] = 1.f;
          if (nlive[i]) {
            const int d = (c0 + cb0 * 4) / Cdg;
            const int ot = d * T + t0;                      // per-image indices fit 32 bits (checked by the launcher)
            noh[i] = off_b[ot * 2 * P];
            now_[i] = off_b[(ot * 2 + 1) * P];
            if (msk_b) nmm[i] = msk_b[ot * P];
          }
          t0 += 8;
          while (t0 >= T) { t0 -= T; ++cb0; }
        }
      };
      load_batch(slice);
      for (int q0 = slice; q0 < NI; q0 += 8 * NB) {
        float oh[NB], ow[NB], mm[NB];
        int tt[NB], cbb[NB];
        bool live[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          oh[i] = noh[i]; ow[i] = now_[i]; mm[i] = nmm[i]; tt[i] = ntt[i]; cbb[i] = ncb[i]; live[i] = nlive[i];
        }
        f32x4 v[NB][4];
        float wgt[NB][4];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          v[i][0] = v[i][1] = v[i][2] = v[i][3] = z;
          wgt[i][0] = wgt[i][1] = wgt[i][2] = wgt[i][3] = 0.f;
          const float h_im = hb + tabh[tt[i]] + oh[i], w_im = wb + tabw[tt[i]] + ow[i];
          if (live[i] && h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W) {
            const int c = c0 + cbb[i] * 4, d = c / Cdg, cin = c - d * Cdg;
            const int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
            const float lh = h_im - (float)hl, lw = w_im - (float)wl, hh = 1.f - lh, hw = 1.f - lw;
            const float* g0 = gp_b + d * (a.H * a.W * Cdg) + cin;
            // corners outside the image: the load is redirected to a clamped (valid) address and its weight set to 0 --
            // four selects instead of four predicated 16-byte loads with zero-filled destinations
            const bool r0 = hl >= 0, r1 = hl + 1 <= a.H - 1, q0c = wl >= 0, q1c = wl + 1 <= a.W - 1;
            const int ya = r0 ? hl : 0, yb = r1 ? hl + 1 : a.H - 1, xa = q0c ? wl : 0, xb = q1c ? wl + 1 : a.W - 1;
            v[i][0] = *reinterpret_cast<const f32x4*>(g0 + (ya * a.W + xa) * Cdg);
            v[i][1] = *reinterpret_cast<const f32x4*>(g0 + (ya * a.W + xb) * Cdg);
            v[i][2] = *reinterpret_cast<const f32x4*>(g0 + (yb * a.W + xa) * Cdg);
            v[i][3] = *reinterpret_cast<const f32x4*>(g0 + (yb * a.W + xb) * Cdg);
            wgt[i][0] = (r0 && q0c) ? hh * hw : 0.f; wgt[i][1] = (r0 && q1c) ? hh * lw : 0.f;
            wgt[i][2] = (r1 && q0c) ? lh * hw : 0.f; wgt[i][3] = (r1 && q1c) ? lh * lw : 0.f;
          }
        }
        if (q0 + 8 * NB < NI) load_batch(q0 + 8 * NB);             // in flight behind this batch's gathers
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int q = q0 + 8 * i;
          if (q < NI) {
            float val[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              val[e] = (wgt[i][0] * v[i][0][e] + wgt[i][1] * v[i][1][e] + wgt[i][2] * v[i][2][e] + wgt[i][3] * v[i][3][e]) * (mm[i] * s_in);

            // fp16 hi + lo with the packed round-toward-zero conversion (hi truncated, lo = the exact remainder truncated:
            // hi + lo still carries 22 bits)
            typedef __fp16 hp2 __attribute__((ext_vector_type(2)));
            union { hp2 h[2]; h4 v4; } uh, ul;
            uh.h[0] = __builtin_amdgcn_cvt_pkrtz(val[0], val[1]);
            uh.h[1] = __builtin_amdgcn_cvt_pkrtz(val[2], val[3]);
            ul.h[0] = __builtin_amdgcn_cvt_pkrtz(val[0] - (float)uh.h[0][0], val[1] - (float)uh.h[0][1]);
            ul.h[1] = __builtin_amdgcn_cvt_pkrtz(val[2] - (float)uh.h[1][0], val[3] - (float)uh.h[1][1]);
{   // range check on the CONVERTED halves: a hi half at or beyond the largest finite fp16 (0x7bff: what the
              // round-toward-zero conversion clamps to), an infinity or a NaN.  (A floating-point test of `val` placed
              // between its computation and the conversion made hipcc 7.2 produce wrong samples in ~3 % of the tiles
              // at 272x480 -- verified on hardware with both a float and an integer-bits formulation -- this one, on the
              // conversion's output, is bit-exact with the kernel without any check.)
              union { hp2 h; unsigned u; } c0, c1;
              c0.h = uh.h[0]; c1.h = uh.h[1];
              ovfbits |= (((c0.u & 0x7fff7fffu) + 0x04010401u) | ((c1.u & 0x7fff7fffu) + 0x04010401u)) & 0x80008000u;
            }
            *reinterpret_cast<h4*>(col_hi + px * ROWB + q * 8) = uh.v4;
            *reinterpret_cast<h4*>(col_lo + px * ROWB + q * 8) = ul.v4;
          }
        }
      }
    }
    __syncthreads();
    // ---- phase 2: acc += W_chunk x col  (hi*hi + lo*hi + hi*lo); the A operands (L2-resident packed weights) are
    // fetched GB steps at a time so that their latency is paid once per group
    {
      constexpr int GB = 3;
      const int sb = kh2 ? (S >> 1) : 0, se = kh2 ? S : (S >> 1);
      const char* bh = col_hi + (nt * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16;
      const char* bl = col_lo + (nt * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16;
      for (int s0 = sb; s0 < se; s0 += GB) {
        h8 Ah[GB][MJ], Al[GB][MJ];
#pragma unroll
        for (int i = 0; i < GB; ++i)
#pragma unroll
          for (int j = 0; j < MJ; ++j) {
            const int mtile = mh + 2 * j, s = (s0 + i) < se ? (s0 + i) : (se - 1);
            const long long wi = ((long long)(chunk * a.S + s) * MT + (mtile < MT ? mtile : 0)) * 64 + lane;
            Ah[i][j] = a.wp_hi[wi];
            Al[i][j] = a.wp_lo[wi];
          }
#pragma unroll
        for (int i = 0; i < GB; ++i) {
          if (s0 + i < se) {
            const h8 Bh = *reinterpret_cast<const h8*>(bh + (s0 + i) * 32);
            const h8 Bl = *reinterpret_cast<const h8*>(bl + (s0 + i) * 32);
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
              if (mh + 2 * j < MT) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i][j], Bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[i][j], Bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i][j], Bl, acc[j], 0, 0, 0);
              }
            }
          }
        }
      }
    }
  }
  if (ovfbits) atomicOr(a.wmax + 2, 1u);      // out of the fp16 hi + lo range somewhere: the exact kernel re-runs (dcn.hip)
  // ---- sum the two K halves through LDS, then store D[row = cout][col = pixel] (+ bias), NCHW, coalesced along pixels
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);                       // [4 waves][MJ][16][64]
  if (kh2) {
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) red[(((wave - 4) * MJ + j) * 16 + e) * 64 + lane] = acc[j][e];
  }
  __syncthreads();
  if (!kh2) {
    const float inv = (1.f / weight_scale(a.wmax)) * (1.f / s_in);
    const int pp = p0 + nt * 32 + (lane & 31);
    if (pp < P) {
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        const int mtile = mh + 2 * j;
        if (mtile >= MT) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int oc = mtile * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          const float v = (acc[j][e] + red[((wave * MJ + j) * 16 + e) * 64 + lane]) * inv;
          a.out[((long long)b * a.Co + oc) * P + pp] = v + (a.bias ? a.bias[oc] : 0.f);
        }
      }
    }
  }
}

}  // namespace

// Bytes of workspace the fast path needs for these shapes (0 = the fast path does not apply).
extern "C" long long cdfo_dcn_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int groups,
                                              int deformable_groups) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || groups != 1 || deformable_groups <= 0) return 0;
  const int T = kh * kw;
  if (C % deformable_groups || (C / deformable_groups) % 4 || Co % 32 || Co > 128 || T > 64) return 0;
  if ((long long)C * H * W >= (1ll << 30) || (long long)deformable_groups * 2 * T * H * W >= (1ll << 30)) return 0;   // 32-bit per-image indices
  int CCH = (KMAX / T) / 4 * 4;
  if (CCH > C) CCH = C;
  const int nchunks = (C + CCH - 1) / CCH, S = (CCH * T + 15) / 16;
  const long long gp = ((long long)B * C * H * W * 4 + 255) / 256 * 256;
  const long long wp = (long long)nchunks * S * (Co / 32) * 64 * 16;
  return gp + 2 * wp + 256;
}

// Called by cdfo_dcn_forward (dcn.hip) after its argument checks.  Returns 1 when it launched, 0 when the fast path does
// not apply (the caller then runs the general kernel), 2 + hipError_t when a launch failed.
int cdfo_dcn_forward_fast(const float* in, const float* offset, const float* mask, const float* weight, const float* bias,
                          float* out, int B, int C, int H, int W, int Co, int Ho, int Wo, int kh, int kw, int sh, int sw, int ph,
                          int pw, int dh, int dw, int groups, int dg, void* workspace, long long workspace_bytes,
                          hipStream_t st, void (*to_gp)(const float*, float*, int, int, int, long long, unsigned*, hipStream_t),
                          const unsigned** rerun_flag) {
  const long long need = cdfo_dcn_workspace_bytes(B, C, H, W, Co, kh, kw, groups, dg);
  if (!need || !workspace || workspace_bytes < need || !aligned16(workspace)) return 0;
  const int T = kh * kw;
  int CCH = (KMAX / T) / 4 * 4;
  if (CCH > C) CCH = C;
  const int nchunks = (C + CCH - 1) / CCH, S = (CCH * T + 15) / 16, MT = Co / 32;
  size_t lds = (size_t)2 * 64 * (S * 32 + 16);
  const size_t red = (size_t)4 * (MT <= 2 ? 1 : 2) * 16 * 64 * 4;                // scratch of the K-half reduction
  if (lds < red) lds = red;
  if (lds + 1024 > 160 * 1024) return 0;
  char* ws = static_cast<char*>(workspace);
  const long long gpb = ((long long)B * C * H * W * 4 + 255) / 256 * 256;
  const long long wpb = (long long)nchunks * S * MT * 64 * 16;
  float* gp = reinterpret_cast<float*>(ws);
  _Float16* whi = reinterpret_cast<_Float16*>(ws + gpb);
  _Float16* wlo = reinterpret_cast<_Float16*>(ws + gpb + wpb);
  unsigned* scale = reinterpret_cast<unsigned*>(ws + gpb + 2 * wpb);
  if (hipMemsetAsync(scale, 0, 16, st) != hipSuccess) return 2 + (int)hipGetLastError();
  to_gp(in, gp, B, C, dg, (long long)H * W, scale + 1, st);
  *rerun_flag = scale + 2;
  const long long nwt = (long long)Co * C * T;
  hipLaunchKernelGGL(dcn_wmax_kernel, dim3((unsigned)((nwt + 2047) / 2048 < 256 ? (nwt + 2047) / 2048 : 256)), dim3(256), 0, st, weight,
                     nwt, scale);
  const long long nw = wpb / 2;
  hipLaunchKernelGGL(dcn_wpack_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, weight, scale, whi, wlo, C, Co, T,
                     CCH, nchunks, S);
  FastArgs a{gp, offset, mask, bias, out, reinterpret_cast<const h8*>(whi), reinterpret_cast<const h8*>(wlo), scale,
             B, C, H, W, Co, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, dg, CCH, nchunks, S};
  static CdfoAttrGrow grow1, grow2;
  dim3 grid(8 * cdiv(cdiv(Ho * Wo, 64), 8), B);
  if (MT <= 2) {
    if (cdfo_grow_max_lds(grow1, reinterpret_cast<const void*>(&dcn_fast_kernel<1>), (int)lds) != hipSuccess) return 2 + (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(dcn_fast_kernel<1>, grid, dim3(512), lds, st, a);
  } else {
    if (cdfo_grow_max_lds(grow2, reinterpret_cast<const void*>(&dcn_fast_kernel<2>), (int)lds) != hipSuccess) return 2 + (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(dcn_fast_kernel<2>, grid, dim3(512), lds, st, a);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 1 : 2 + (int)e;
}
