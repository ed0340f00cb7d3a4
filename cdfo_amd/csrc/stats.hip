// Whole-image reductions and the tiny per-image "fold" kernels of the channel-attention blocks.
//
// The reference materialises q,k,v = [B,heads,c,H*W], L2-normalises q and k over H*W, forms a c x c attention per head
// and applies it to v, then a 1x1 projection (MDTA arch/SIDECVSR_our.py:1555-1575; DualAttAlignment 3459-3489).
// Because attention and projection are both linear maps on the channel axis, attn@v followed by project_out equals ONE
// per-image 64x64 matrix applied to every pixel of v.  So the HIP path does:
//   (1) gram_partial   : one streaming pass over q,k  -> block-diagonal Gram + squared norms (fixed-order partials)
//   (2) *_fold         : reduce partials, normalise, softmax, multiply with the projection -> per-image packed 1x1
//                        weights in the layout cdfo_conv_igemm reads
//   (3) cdfo_conv_igemm with per-image weights (+ residual) : one streaming pass over v.
// Partials are reduced in a fixed order, so results are run-to-run deterministic.
#include "common.h"

namespace {

// partial[b][chunk][c] = sum over the chunk's pixels of in[b][p][c]          (C == 64)
// 16 lanes (a float4 each) cover one pixel, 16 pixel sub-streams per workgroup, fixed-order tree at the end.
__global__ __launch_bounds__(256) void chan_sum_partial_kernel(const float* __restrict__ in, int ldi, long long P,
                                                               int nchunk, float* __restrict__ partial) {
  __shared__ f32x4 red[16][16];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int c4 = threadIdx.x & 15, sub = threadIdx.x >> 4;
  const long long per = (P + nchunk - 1) / nchunk;
  const long long p0 = chunk * per, p1 = (p0 + per < P) ? p0 + per : P;
  const float* base = in + (long long)b * P * ldi + c4 * 4;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  long long p = p0 + sub;
  for (; p + 16 < p1; p += 32) {
    s0 += *reinterpret_cast<const f32x4*>(base + p * ldi);
    s1 += *reinterpret_cast<const f32x4*>(base + (p + 16) * ldi);
  }
  if (p < p1) s0 += *reinterpret_cast<const f32x4*>(base + p * ldi);
  red[sub][c4] = s0 + s1;
  __syncthreads();
  if (sub == 0) {
    f32x4 t = red[0][c4];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][c4];
    *reinterpret_cast<f32x4*>(partial + ((long long)b * nchunk + chunk) * 64 + c4 * 4) = t;
  }
}

// partial[b][chunk][c*(CH+2) + j]: j<CH: sum_p q[p][c]*k[p][head(c)*CH+j];  j==CH: sum q^2;  j==CH+1: sum k[.][c]^2
template <int CH>
__global__ __launch_bounds__(256) void gram_partial_kernel(const float* __restrict__ q, int ldq,
                                                           const float* __restrict__ k, int ldk, long long P,
                                                           int nchunk, float* __restrict__ partial) {
  __shared__ float red[4][64 * (CH + 2)];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int c = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int hbase = (c / CH) * CH;
  const long long per = (P + nchunk - 1) / nchunk;
  const long long p0 = chunk * per, p1 = (p0 + per < P) ? p0 + per : P;
  const float* qb = q + (long long)b * P * ldq;
  const float* kb = k + (long long)b * P * ldk;
  float g[CH], sq = 0.f, sk = 0.f;
#pragma unroll
  for (int j = 0; j < CH; ++j) g[j] = 0.f;
  for (long long p = p0 + sub; p < p1; p += 4) {
    const float qv = qb[p * ldq + c];
    const float kc = kb[p * ldk + c];
    sq += qv * qv;
    sk += kc * kc;
#pragma unroll
    for (int j4 = 0; j4 < CH / 4; ++j4) {
      const f32x4 kv = *reinterpret_cast<const f32x4*>(kb + p * ldk + hbase + j4 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[j4 * 4 + e] += qv * kv[e];
    }
  }
#pragma unroll
  for (int j = 0; j < CH; ++j) red[sub][c * (CH + 2) + j] = g[j];
  red[sub][c * (CH + 2) + CH] = sq;
  red[sub][c * (CH + 2) + CH + 1] = sk;
  __syncthreads();
  float* out = partial + ((long long)b * nchunk + chunk) * 64 * (CH + 2);
  for (int i = threadIdx.x; i < 64 * (CH + 2); i += 256) out[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// reduce partials -> LDS stats[64*(CH+2)], then attn[c][j] = softmax_j(G/(|q_c||k_j|) * temperature[head])
template <int CH>
__device__ void reduce_and_softmax(const float* __restrict__ partial, int nchunk, const float* __restrict__ temperature,
                                   float* stats, float* attn) {
  constexpr int N = 64 * (CH + 2);
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    float s = 0.f;
#pragma unroll 8
    for (int ch = 0; ch < nchunk; ++ch) s += partial[(long long)ch * N + i];
    stats[i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = threadIdx.x, head = c / CH;
    const float nq = fmaxf(sqrtf(stats[c * (CH + 2) + CH]), 1e-12f);
    const float t = temperature[head];
    float l[CH], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float nk = fmaxf(sqrtf(stats[(head * CH + j) * (CH + 2) + CH + 1]), 1e-12f);
      l[j] = stats[c * (CH + 2) + j] / (nq * nk) * t;
      mx = fmaxf(mx, l[j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) { l[j] = expf(l[j] - mx); sum += l[j]; }
#pragma unroll
    for (int j = 0; j < CH; ++j) attn[c * CH + j] = l[j] / sum;
  }
  __syncthreads();
}

// MDTA: Wout[b] = project_out (64x64) x blockdiag(attn)   packed [cin/4][64][4]
__global__ __launch_bounds__(256) void mdta_fold_kernel(const float* __restrict__ partial, int nchunk,
                                                        const float* __restrict__ temperature,
                                                        const float* __restrict__ proj, float* __restrict__ wout) {
  constexpr int CH = 8;
  __shared__ float stats[64 * (CH + 2)];
  __shared__ float attn[64 * CH];
  const int b = blockIdx.x;
  reduce_and_softmax<CH>(partial + (long long)b * nchunk * 64 * (CH + 2), nchunk, temperature, stats, attn);
  float* wo = wout + (long long)b * 4096;
  for (int idx = threadIdx.x; idx < 4096; idx += blockDim.x) {
    const int o = idx >> 6, i = idx & 63, hb = (i / CH) * CH, j = i - hb;
    float s = 0.f;
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) s += proj[o * 64 + hb + cc] * attn[(hb + cc) * CH + j];
    wo[((i >> 2) * 64 + o) * 4 + (i & 3)] = s;
  }
}

// DualAttAlignment (arch.py:3455-3493): both MSA blocks share q and k, so with A = blockdiag(attn), P = project_out,
// Wf = fusion_out = [Wa | Wb], g1/g2 = conv_du(avgpool(warped/pred)):
//   relu(Wf . cat[P A (g1*warped) + P A (g2*pred), x]) = relu([Wa P A diag(g1) | Wa P A diag(g2) | Wb] . cat[warped,pred,x])
// -> per-image packed 1x1 weights for Cin = 192.
__global__ __launch_bounds__(256) void align_fold_kernel(const float* __restrict__ gram_partial, int nchunk_g,
                                                         const float* __restrict__ sum_warp,
                                                         const float* __restrict__ sum_pred, int nchunk_s, float inv_P,
                                                         const float* __restrict__ temperature,
                                                         const float* __restrict__ du0_w, const float* __restrict__ du0_b,
                                                         const float* __restrict__ du2_w, const float* __restrict__ du2_b,
                                                         const float* __restrict__ proj, const float* __restrict__ wf,
                                                         float* __restrict__ wout) {
  constexpr int CH = 16;
  __shared__ float stats[64 * (CH + 2)];
  __shared__ float attn[64 * CH];
  __shared__ float Q[64 * 64];
  __shared__ float mean[2][64];
  __shared__ float hid[2][4];
  __shared__ float gate[2][64];
  const int b = blockIdx.x;
  reduce_and_softmax<CH>(gram_partial + (long long)b * nchunk_g * 64 * (CH + 2), nchunk_g, temperature, stats, attn);
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
    const float* sp = (which ? sum_pred : sum_warp) + (long long)b * nchunk_s * 64;
    float s = 0.f;
#pragma unroll 8
    for (int ch = 0; ch < nchunk_s; ++ch) s += sp[ch * 64 + c];
    mean[which][c] = s * inv_P;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int which = threadIdx.x >> 2, m = threadIdx.x & 3;
    float s = du0_b[m];
    for (int c = 0; c < 64; ++c) s += du0_w[m * 64 + c] * mean[which][c];
    hid[which][m] = fmaxf(s, 0.f);
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
    float s = du2_b[c];
    for (int m = 0; m < 4; ++m) s += du2_w[c * 4 + m] * hid[which][m];
    gate[which][c] = 1.f / (1.f + expf(-s));
  }
  // Q = Wa . P
  for (int idx = threadIdx.x; idx < 4096; idx += blockDim.x) {
    const int o = idx >> 6, c = idx & 63;
    float s = 0.f;
    for (int m = 0; m < 64; ++m) s += wf[o * 128 + m] * proj[m * 64 + c];
    Q[idx] = s;
  }
  __syncthreads();
  float* wo = wout + (long long)b * (192 * 64);
  for (int idx = threadIdx.x; idx < 4096; idx += blockDim.x) {
    const int o = idx >> 6, i = idx & 63, hb = (i / CH) * CH, j = i - hb;
    float r = 0.f;
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) r += Q[o * 64 + hb + cc] * attn[(hb + cc) * CH + j];
    wo[((i >> 2) * 64 + o) * 4 + (i & 3)] = r * gate[0][i];
    wo[(((64 + i) >> 2) * 64 + o) * 4 + (i & 3)] = r * gate[1][i];
    wo[(((128 + i) >> 2) * 64 + o) * 4 + (i & 3)] = wf[o * 128 + 64 + i];
  }
}

// out[b][o] = act2(W2 . act1(W1 . mean[b] + b1) + b2)     (second layer optional)      all widths <= 64
__global__ __launch_bounds__(64) void vec_mlp_kernel(const float* __restrict__ sum_partial, int nchunk, float inv_P,
                                                     const float* __restrict__ w1, const float* __restrict__ b1, int c1,
                                                     int act1, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, int c2, int act2,
                                                     float* __restrict__ out) {
  __shared__ float mean[64], hid[64];
  const int b = blockIdx.x, t = threadIdx.x;
  float s = 0.f;
#pragma unroll 8
  for (int ch = 0; ch < nchunk; ++ch) s += sum_partial[((long long)b * nchunk + ch) * 64 + t];
  mean[t] = s * inv_P;
  __syncthreads();
  if (t < c1) {
    float a = b1 ? b1[t] : 0.f;
    for (int c = 0; c < 64; ++c) a += w1[t * 64 + c] * mean[c];
    hid[t] = act_apply(a, act1);
  }
  __syncthreads();
  if (!w2) {
    if (t < c1) out[(long long)b * c1 + t] = hid[t];
    return;
  }
  if (t < c2) {
    float a = b2 ? b2[t] : 0.f;
    for (int c = 0; c < c1; ++c) a += w2[t * c1 + c] * hid[c];
    out[(long long)b * c2 + t] = act_apply(a, act2);
  }
}

}  // namespace

extern "C" int cdfo_chan_sum_partial(const float* in, int ldi, int B, long long P, int nchunk, float* partial,
                                     void* stream) {
  if (B <= 0 || P <= 0 || nchunk <= 0 || ldi % 4) return CDFO_EINVAL;
  if (!aligned16(in) || !aligned16(partial)) return CDFO_EALIGN;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_CHAN_SUM, 0, 4.0*64*(double)B*P);
  hipLaunchKernelGGL(chan_sum_partial_kernel, dim3(nchunk, B), dim3(256), 0, static_cast<hipStream_t>(stream), in, ldi,
                     P, nchunk, partial);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_gram_partial(const float* q, int ldq, const float* k, int ldk, int B, long long P, int ch_per_head,
                                 int nchunk, float* partial, void* stream) {
  if (B <= 0 || P <= 0 || nchunk <= 0 || ldk % 4) return CDFO_EINVAL;
  if (!aligned16(k)) return CDFO_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_GRAM, 2.0*64*(ch_per_head+2)*(double)B*P, 4.0*128*(double)B*P);
  if (ch_per_head == 8)
    hipLaunchKernelGGL(gram_partial_kernel<8>, dim3(nchunk, B), dim3(256), 0, st, q, ldq, k, ldk, P, nchunk, partial);
  else if (ch_per_head == 16)
    hipLaunchKernelGGL(gram_partial_kernel<16>, dim3(nchunk, B), dim3(256), 0, st, q, ldq, k, ldk, P, nchunk, partial);
  else
    return CDFO_EINVAL;
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_mdta_fold(const float* partial, int nchunk, const float* temperature, const float* proj_w, int B,
                              float* wout, void* stream) {
  if (B <= 0 || nchunk <= 0) return CDFO_EINVAL;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_FOLD, 0, 0);
  hipLaunchKernelGGL(mdta_fold_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), partial, nchunk,
                     temperature, proj_w, wout);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_align_fold(const float* gram_partial, int nchunk_g, const float* sum_warp, const float* sum_pred,
                               int nchunk_s, long long P, const float* temperature, const float* du0_w,
                               const float* du0_b, const float* du2_w, const float* du2_b, const float* proj_w,
                               const float* fusion_w, int B, float* wout, void* stream) {
  if (B <= 0 || nchunk_g <= 0 || nchunk_s <= 0 || P <= 0) return CDFO_EINVAL;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_FOLD, 0, 0);
  hipLaunchKernelGGL(align_fold_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), gram_partial, nchunk_g,
                     sum_warp, sum_pred, nchunk_s, 1.0f / (float)P, temperature, du0_w, du0_b, du2_w, du2_b, proj_w,
                     fusion_w, wout);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_vec_mlp(const float* sum_partial, int nchunk, long long P, const float* w1, const float* b1, int c1,
                            int act1, const float* w2, const float* b2, int c2, int act2, int B, float* out,
                            void* stream) {
  if (B <= 0 || nchunk <= 0 || c1 <= 0 || c1 > 64 || (w2 && (c2 <= 0 || c2 > 64))) return CDFO_EINVAL;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_FOLD, 0, 0);
  hipLaunchKernelGGL(vec_mlp_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), sum_partial, nchunk,
                     1.0f / (float)P, w1, b1, c1, act1, w2, b2, c2, act2, out);
  CDFO_LAUNCH_CHECK();
  return 0;
}
