// Backward kernels of the CVSR_V8 training path (`train_LD_37.py:376-381` calls the module under
// autograd).  cdfo_amd/autograd.py composes the training-mode forward from torch.autograd.Function objects whose forward
// runs the exact-fp32 forward kernels of this library and whose backward runs:
//   * the forward kernels again where the adjoint IS a forward op: a stride-1 convolution's input gradient is a convolution
//     with the flipped / transposed weights (cdfo_conv_igemm), the strided 16-channel convolutions and their transposes
//     swap roles (cdfo_small_conv16), the depthwise and 9-tap convolutions take flipped taps, the channel attentions'
//     value / query / key gradients are per-image 1x1 convolutions;
//   * the kernels of this file for everything else: weight gradients (split-K exact-fp32 MFMA, partial slabs summed in a
//     fixed order: bit-reproducible), column sums / dot products (bias, LayerNorm and gate gradients), activation and
//     LayerNorm backward, the prior U-net's spatial gate, the Gumbel hard mask as a tensor, the row / column / window
//     attention backward (flash style: probabilities recomputed from the forward's statistics, nothing of size L x L is
//     stored), flow_warp's scatter, the bilinear resamples' adjoints.
// All tensors fp32 pixel-major [B][H][W][C] with a pixel pitch ld >= C (floats), like the rest of the library.
// Correctness-first: plain VALU kernels except the weight gradient; training runs on 64x64 crops (train_LD_37.py:316-325).
#include "common.h"

namespace {

inline int tr_grid(long long threads, int cap = 8192) {
  long long b = (threads + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------------------------------------------------------------ weight gradient
// out[a][b][ky][kx] = sum_{n,y,x} S[n,y,x,a] * L[n, y*s + ky - pad, x*s + kx - pad, b]
//   convolution:            S = grad_out [N,Hs,Ws,A=Cout], L = input  [N,Hl,Wl,B=Cin]  -> dW[Cout][Cin][k][k]   (OIHW)
//   transposed convolution: S = input    [N,Hs,Ws,A=Cin],  L = grad_out [N,Hl,Wl,B=Cout] -> dW[Cin][Cout][k][k]  (IOHW)
// One workgroup = (64 x 64 block of (a, b), one tap, one slice of the N*Hs rows); each wave walks its rows two pixels per
// v_mfma_f32_32x32x2_f32 step (A operand = S values of 32 a's, B operand = L values of 32 b's, both 128-byte coalesced
// reads straight from global memory); the four waves' accumulators are summed through LDS and written to the slice's slab.
struct wg_args {
  const float* S; int lds_; int A;
  const float* L; int ldl; int Bc;
  int N, Hs, Ws, Hl, Wl, ks, stride, pad;
  float* slab;        // [nsplit][T][A64*64][B64*64] partial sums
  int nsplit, a_blocks, b_blocks;
};

__global__ __launch_bounds__(256) void conv_wgrad_kernel(wg_args g) {
  __shared__ float red[3][64 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, kk = lane >> 5;
  const int ab = blockIdx.x % g.a_blocks, bb = blockIdx.x / g.a_blocks;
  const int tap = blockIdx.y, ky = tap / g.ks, kx = tap - ky * g.ks, sp = blockIdx.z;
  const long long rows = (long long)g.N * g.Hs;
  const long long r0 = rows * sp / g.nsplit, r1 = rows * (sp + 1) / g.nsplit;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int a0 = ab * 64 + r, b0 = bb * 64 + r;
  const bool am0 = a0 < g.A, am1 = a0 + 32 < g.A, bm0 = b0 < g.Bc, bm1 = b0 + 32 < g.Bc;
  // valid x range of this tap: 0 <= x*s + kx - pad < Wl
  int x_lo = 0, x_hi = g.Ws;
  {
    const int off = kx - g.pad;
    if (off < 0) x_lo = (-off + g.stride - 1) / g.stride;
    const int lim = g.Wl - 1 - off;                       // x*s <= lim
    x_hi = lim < 0 ? 0 : min(g.Ws, lim / g.stride + 1);
  }
  for (long long row = r0 + wave; row < r1; row += 4) {
    const int n = (int)(row / g.Hs), y = (int)(row - (long long)n * g.Hs);
    const int yl = y * g.stride + ky - g.pad;
    if (yl < 0 || yl >= g.Hl) continue;
    const float* Sr = g.S + ((long long)n * g.Hs + y) * g.Ws * g.lds_;
    const float* Lr = g.L + (((long long)n * g.Hl + yl) * g.Wl + (kx - g.pad)) * g.ldl;
    // (unconditional loads from clamped addresses, masked afterwards -- a predicated load is a branch of its own; four steps'
    // requests leave together)
#pragma unroll 4
    for (int x = x_lo; x < x_hi; x += 2) {
      const int xx = x + kk;
      const bool v = xx < x_hi;
      const float* sp_ = Sr + (long long)(v ? xx : x_lo) * g.lds_;
      const float* lp = Lr + (long long)(v ? xx : x_lo) * g.stride * g.ldl;
      const float t0 = sp_[am0 ? a0 : 0], t1 = sp_[am1 ? a0 + 32 : 0], u0 = lp[bm0 ? b0 : 0], u1 = lp[bm1 ? b0 + 32 : 0];
      const float sa0 = (v && am0) ? t0 : 0.f, sa1 = (v && am1) ? t1 : 0.f;
      const float lb0 = (v && bm0) ? u0 : 0.f, lb1 = (v && bm1) ? u1 : 0.f;
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa0, lb0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa0, lb1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa1, lb0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa1, lb1, acc[1][1], 0, 0, 0);
    }
  }
  // D[row = a][col = b]: lane holds col r, rows (e & 3) + 8 (e >> 2) + 4 kk.  Waves 1-3 -> LDS, wave 0 adds in order.
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          red[wave - 1][(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk) * 64 + j * 32 + r] = acc[i][j][e];
  }
  __syncthreads();
  if (wave == 0) {
    const int T = g.ks * g.ks;
    float* out = g.slab + (((long long)sp * T + tap) * (g.a_blocks * 64)) * (g.b_blocks * 64);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ra = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk, cb = j * 32 + r;
          const float v = ((acc[i][j][e] + red[0][ra * 64 + cb]) + red[1][ra * 64 + cb]) + red[2][ra * 64 + cb];
          out[(long long)(ab * 64 + ra) * (g.b_blocks * 64) + bb * 64 + cb] = v;
        }
  }
}

// The same block decomposition with split-bf16 operands (round 3): 16 pixels per step instead of 2 -- a lane loads 8 consecutive
// pixels of its a / b channel (the K half its fragment holds), splits them into bf16 hi + lo in registers and the step runs
// hi*lo + lo*hi + hi*hi on v_mfma_f32_32x32x16_bf16: 12 MFMAs of 32 cycles per 16 pixels against 32 of 64 cycles above.
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void conv_wgrad_bf16x3_kernel(wg_args g) {
  __shared__ float red[3][64 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, kk = lane >> 5;
  const int ab = blockIdx.x % g.a_blocks, bb = blockIdx.x / g.a_blocks;
  const int tap = blockIdx.y, ky = tap / g.ks, kx = tap - ky * g.ks, sp = blockIdx.z;
  const long long rows = (long long)g.N * g.Hs;
  const long long r0 = rows * sp / g.nsplit, r1 = rows * (sp + 1) / g.nsplit;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int a0 = ab * 64 + r, b0 = bb * 64 + r;
  const bool am[2] = {a0 < g.A, a0 + 32 < g.A}, bm[2] = {b0 < g.Bc, b0 + 32 < g.Bc};
  int x_lo = 0, x_hi = g.Ws;
  {
    const int off = kx - g.pad;
    if (off < 0) x_lo = (-off + g.stride - 1) / g.stride;
    const int lim = g.Wl - 1 - off;
    x_hi = lim < 0 ? 0 : min(g.Ws, lim / g.stride + 1);
  }
  // The (row, 16-pixel step) sequence of this wave, flattened, with the NEXT step's 32 operand values requested before the current
  // step's splits and MFMAs (round 5: every step used to wait a full memory latency for its own loads -- 47 of the 160 ms of a
  // backward at the training script's crop size went here at ~15 % of the matrix rate).
  long long row = r0 + wave;
  int x = x_lo;
  const float *Sr = nullptr, *Lr = nullptr;
  auto seek = [&]() {          // first valid (row, x) at or after the current one; false when the slice is exhausted
    for (; row < r1; row += 4, x = x_lo) {
      if (x >= x_hi) continue;
      const int n = (int)(row / g.Hs), y = (int)(row - (long long)n * g.Hs);
      const int yl = y * g.stride + ky - g.pad;
      if (yl < 0 || yl >= g.Hl) continue;
      Sr = g.S + ((long long)n * g.Hs + y) * g.Ws * g.lds_;
      Lr = g.L + (((long long)n * g.Hl + yl) * g.Wl + (kx - g.pad)) * g.ldl;
      return true;
    }
    return false;
  };
  auto fetch = [&](float (&sv)[2][8], float (&lv)[2][8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int xx = x + 8 * kk + j;
      const bool v = xx < x_hi;
      const float* sp_ = Sr + (long long)(v ? xx : x_lo) * g.lds_;
      const float* lp = Lr + (long long)(v ? xx : x_lo) * g.stride * g.ldl;
      // unconditional loads from clamped (valid) addresses, masked afterwards: a predicated load is a branch of its own and the 32
      // loads of a step then leave one at a time
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float ts = sp_[am[i] ? a0 + 32 * i : 0], tl = lp[bm[i] ? b0 + 32 * i : 0];
        sv[i][j] = (v && am[i]) ? ts : 0.f;
        lv[i][j] = (v && bm[i]) ? tl : 0.f;
      }
    }
  };
  float sv[2][8], lv[2][8];
  bool have = x_lo < x_hi && seek();
  if (have) fetch(sv, lv);
  while (have) {
    float sn[2][8], ln[2][8];
    x += 16;
    const bool more = seek();
    if (more) fetch(sn, ln);
    __builtin_amdgcn_sched_barrier(0);      // the requests stay ahead of this step's arithmetic (the scheduler would sink them to their uses)
    {
      wg_bf16x8 sh[2], sl[2], lh[2], ll[2];
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          sh[i][j] = (__bf16)sv[i][j]; sl[i][j] = (__bf16)(sv[i][j] - (float)sh[i][j]);
          lh[i][j] = (__bf16)lv[i][j]; ll[i][j] = (__bf16)(lv[i][j] - (float)lh[i][j]);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sh[i], ll[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl[i], lh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sh[i], lh[j], acc[i][j], 0, 0, 0);
        }
    }
    have = more;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { sv[i][j] = sn[i][j]; lv[i][j] = ln[i][j]; }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          red[wave - 1][(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk) * 64 + j * 32 + r] = acc[i][j][e];
  }
  __syncthreads();
  if (wave == 0) {
    const int T = g.ks * g.ks;
    float* out = g.slab + (((long long)sp * T + tap) * (g.a_blocks * 64)) * (g.b_blocks * 64);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ra = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk, cb = j * 32 + r;
          const float v = ((acc[i][j][e] + red[0][ra * 64 + cb]) + red[1][ra * 64 + cb]) + red[2][ra * 64 + cb];
          out[(long long)(ab * 64 + ra) * (g.b_blocks * 64) + bb * 64 + cb] = v;
        }
  }
}

// (Round 5 also built a form with one workgroup per (tile, kernel ROW, split) that serves the three kx taps from one ten-pixel window --
// a third of the operand loads and splits per MFMA.  Measured on the training step (backward, 20 x 64 x 64): 326 ms with a 64 x 64
// tile (192 accumulators: one wave per SIMD, every step's load latency exposed) and 223 ms with a 64 x 32 tile (two waves per SIMD)
// against 153 ms for the per-tap kernel above at three waves per SIMD; removed again.  What the per-tap kernel lacks is not fewer
// loads but a software pipeline that hipcc does not collapse: see DESIGN.md section 5.000 #6.)
// dw[(a * Btot + b_off + b) * T + tap] = scale * sum_sp slab[sp][tap][a][b]   (fixed order)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, int T, int A, int Bc,
                                                                int Apad, int Bpad, int Btot, int b_off, float* __restrict__ dw) {
  const long long n = (long long)A * Bc * T;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int tap = (int)(i % T);
    const long long ab = i / T;
    const int b = (int)(ab % Bc), a = (int)(ab / Bc);
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += slab[(((long long)sp * T + tap) * Apad + a) * Bpad + b];
    dw[((long long)a * Btot + b_off + b) * T + tap] = s;
  }
}

// ------------------------------------------------------------------------------------------------ column dots
// part[img][chunk][c] = sum over the chunk's pixels of a[p][c] * (b ? b[p][c] : 1);  finish: out[img][c] = sum_chunk
__global__ __launch_bounds__(256) void coldot_partial_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                             int ldb, long long P, int C, int nchunk, float* __restrict__ part) {
  const int img = blockIdx.z, chunk = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  __shared__ float red[4][64];
  const long long p0 = P * chunk / nchunk, p1 = P * (chunk + 1) / nchunk;
  float s = 0.f;
  if (c < C)
    for (long long p = p0 + rl; p < p1; p += 4) {
      const long long gp = (long long)img * P + p;
      s += a[gp * lda + c] * (b ? b[gp * ldb + c] : 1.f);
    }
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C) part[((long long)img * nchunk + chunk) * C + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
// one WAVE per output: lane l sums chunks l, l + 64, ... (fixed order), then a fixed shuffle tree -- a serial sum over up to 512
// chunks by a handful of threads took 100 us per bias gradient
__global__ __launch_bounds__(256) void coldot_finish_kernel(const float* __restrict__ part, int nchunk, int C, int nimg, float scale,
                                                            float* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= nimg * C) return;
  const int img = i / C, c = i - img * C;
  float s = 0.f;
  for (int k = lane; k < nchunk; k += 64) s += part[((long long)img * nchunk + k) * C + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[i] = s * scale;
}

// ------------------------------------------------------------------------------------------------ element-wise
// mode 0: out = a * b;  1: out = a * (1 - b);  2: out = a + b;  3 (act backward): out = a * act'(y = b), act in `aux`
// (1 LeakyReLU 0.1, 2 ReLU, 3 sigmoid);  4: out[p][c] = a[img][c] * scale (row broadcast, a = [nimg][C])
__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, long long rows,
                                                 int C, int mode, int aux, float scale, long long P, float* __restrict__ out, int ldo) {
  const int cq = C >> 2;
  const long long n = rows * cq;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long p = i / cq;
    const int c = (int)(i - p * cq) * 4;
    f32x4 o;
    if (mode == 4) {
      o = *reinterpret_cast<const f32x4*>(a + (p / P) * lda + c) * scale;
    } else {
      const f32x4 va = *reinterpret_cast<const f32x4*>(a + p * lda + c), vb = *reinterpret_cast<const f32x4*>(b + p * ldb + c);
      if (mode == 0) o = va * vb;
      else if (mode == 1) o = va * (1.f - vb);
      else if (mode == 2) o = va + vb;
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float y = vb[e];
          const float d = aux == 1 ? (y > 0.f ? 1.f : 0.1f) : (aux == 2 ? (y > 0.f ? 1.f : 0.f) : y * (1.f - y));
          o[e] = va[e] * d;
        }
      }
    }
    *reinterpret_cast<f32x4*>(out + p * ldo + c) = o;
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// per pixel over 64 channels (arch.py:1169-1185, biased variance, eps 1e-5): 16 lanes x float4 per pixel.
// dx = rstd * (dy - mean(dy) - xhat * mean(dy * xhat)), dy = g * gamma;  part[block][0][c] += g * xhat, [1][c] += g
__global__ __launch_bounds__(256) void layernorm64_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g, int ldg,
                                                              const float* __restrict__ gamma, long long npix, float* __restrict__ dx,
                                                              int ldo, float* __restrict__ part) {
  __shared__ float acc[2][16][64];
  const int q = threadIdx.x & 15, pl = threadIdx.x >> 4;           // 16 pixels per pass
  const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + q * 4);
  f32x4 sgx = {0.f, 0.f, 0.f, 0.f}, sg = {0.f, 0.f, 0.f, 0.f};
  for (long long p = (long long)blockIdx.x * 16 + pl; p < npix; p += (long long)gridDim.x * 16) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + p * ldx + q * 4), gv = *reinterpret_cast<const f32x4*>(g + p * ldg + q * 4);
    float sm = (xv[0] + xv[1]) + (xv[2] + xv[3]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    const f32x4 d = xv - sm * (1.f / 64.f);
    float sq = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = 1.f / sqrtf(sq * (1.f / 64.f) + 1e-5f);
    const f32x4 xh = d * rstd, dy = gv * gm;
    float m1 = (dy[0] + dy[1]) + (dy[2] + dy[3]);
    float m2 = (dy[0] * xh[0] + dy[1] * xh[1]) + (dy[2] * xh[2] + dy[3] * xh[3]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o, 64); m2 += __shfl_xor(m2, o, 64); }
    *reinterpret_cast<f32x4*>(dx + p * ldo + q * 4) = (dy - m1 * (1.f / 64.f) - xh * (m2 * (1.f / 64.f))) * rstd;
    sgx += gv * xh;
    sg += gv;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { acc[0][pl][q * 4 + e] = sgx[e]; acc[1][pl][q * 4 + e] = sg[e]; }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int w = threadIdx.x >> 6, c = threadIdx.x & 63;
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += acc[w][k][c];
    part[((long long)blockIdx.x * 2 + w) * 64 + c] = s;
  }
}

// ------------------------------------------------------------------------------------------------ depthwise 3x3 weight gradient
// part[block][tap][c] = sum over the block's pixels of g[p][c] * x[p + tap][c]   (zero padding)
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g, int ldg,
                                                              int B, int H, int W, int C, float* __restrict__ part) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long long npix = (long long)B * H * W;
  const long long p0 = npix * blockIdx.x / gridDim.x, p1 = npix * (blockIdx.x + 1) / gridDim.x;
  for (long long p = p0; p < p1; ++p) {
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    const float gv = g[p * ldg + c];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      if (yy + dy >= 0 && yy + dy < H && xx + dx >= 0 && xx + dx < W) s[t] += gv * x[(p + (long long)dy * W + dx) * ldx + c];
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) part[((long long)blockIdx.x * 9 + t) * C + c] = s[t];
}

// ------------------------------------------------------------------------------------------------ spatial gate (16 channels)
// forward (arch.py:2719-2730): out = t * sigmoid(conv7x7([max_c t, mean_c t]) + b)
// pass 1: pooled[p] = {max, mean, argmax};  pass 2: gate, ds = (sum_c g t) gate (1 - gate), dt = g * gate;
// pass 3: dt += conv7x7^T(ds) routed to the arg-max channel / spread over the mean;  dw by cdfo_sg16_wgrad
__global__ __launch_bounds__(256) void sg16_pool_kernel(const float* __restrict__ t, int ldt, long long npix, float* __restrict__ pooled) {
  for (long long p = blockIdx.x * 256ll + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
    float mx = -INFINITY, sm = 0.f;
    int am = 0;
    for (int c = 0; c < 16; ++c) {
      const float v = t[p * ldt + c];
      if (v > mx) { mx = v; am = c; }
      sm += v;
    }
    pooled[p * 4 + 0] = mx; pooled[p * 4 + 1] = sm * (1.f / 16.f); pooled[p * 4 + 2] = (float)am; pooled[p * 4 + 3] = 0.f;
  }
}
__global__ __launch_bounds__(256) void sg16_ds_kernel(const float* __restrict__ t, int ldt, const float* __restrict__ g, int ldg,
                                                      const float* __restrict__ pooled, const float* __restrict__ w,
                                                      const float* __restrict__ bias, int B, int H, int W, float* __restrict__ ds,
                                                      float* __restrict__ dt, int ldo) {
  const long long npix = (long long)B * H * W;
  for (long long p = blockIdx.x * 256ll + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
    const int x = (int)(p % W), y = (int)((p / W) % H);
    float s = bias[0];
    for (int dy = 0; dy < 7; ++dy) {
      const int yy = y + dy - 3;
      if (yy < 0 || yy >= H) continue;
      for (int dx = 0; dx < 7; ++dx) {
        const int xx = x + dx - 3;
        if (xx < 0 || xx >= W) continue;
        const long long q = p + (long long)(dy - 3) * W + (dx - 3);
        s += pooled[q * 4] * w[dy * 7 + dx] + pooled[q * 4 + 1] * w[49 + dy * 7 + dx];
      }
    }
    const float gate = 1.f / (1.f + expf(-s));
    float dot = 0.f;
    for (int c = 0; c < 16; ++c) {
      const float gv = g[p * ldg + c];
      dot += gv * t[p * ldt + c];
      dt[p * ldo + c] = gv * gate;
    }
    ds[p] = dot * gate * (1.f - gate);
  }
}
__global__ __launch_bounds__(256) void sg16_fin_kernel(const float* __restrict__ pooled, const float* __restrict__ ds,
                                                       const float* __restrict__ w, int B, int H, int W, float* __restrict__ dt, int ldo) {
  const long long npix = (long long)B * H * W;
  for (long long p = blockIdx.x * 256ll + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
    const int x = (int)(p % W), y = (int)((p / W) % H);
    float dmax = 0.f, dmean = 0.f;       // d loss / d pooled[p]: every output q = p - (tap - 3) that saw p through tap
    for (int dy = 0; dy < 7; ++dy) {
      const int yy = y - (dy - 3);
      if (yy < 0 || yy >= H) continue;
      for (int dx = 0; dx < 7; ++dx) {
        const int xx = x - (dx - 3);
        if (xx < 0 || xx >= W) continue;
        const float d = ds[p - (long long)(dy - 3) * W - (dx - 3)];
        dmax += d * w[dy * 7 + dx];
        dmean += d * w[49 + dy * 7 + dx];
      }
    }
    const int am = (int)pooled[p * 4 + 2];
    for (int c = 0; c < 16; ++c) dt[p * ldo + c] += dmean * (1.f / 16.f) + (c == am ? dmax : 0.f);
  }
}
// dw[ch][tap] = sum_p ds[p] * pooled[p + tap - 3][ch], db = sum ds: one workgroup per (ch, tap), slot 98 = bias
__global__ __launch_bounds__(256) void sg16_wgrad_kernel(const float* __restrict__ pooled, const float* __restrict__ ds, int B, int H, int W,
                                                         float* __restrict__ dw) {
  __shared__ float red[4];
  const int k = blockIdx.x, ch = k / 49, tap = k % 49, dy = tap / 7 - 3, dx = tap % 7 - 3;
  const long long npix = (long long)B * H * W;
  float s = 0.f;
  for (long long p = threadIdx.x; p < npix; p += 256) {
    if (k == 98) { s += ds[p]; continue; }
    const int x = (int)(p % W), y = (int)((p / W) % H);
    if (y + dy < 0 || y + dy >= H || x + dx < 0 || x + dx >= W) continue;
    s += ds[p] * pooled[(p + (long long)dy * W + dx) * 4 + ch];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dw[k] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------ Gumbel hard mask as a tensor
// mask[p][c] = softmax_c(vmax[b][c] - log(-log u[b][c][p])) >= 0.5   (arch.py:2168-2195); 4 lanes per pixel.
// RNG: the uniforms are drawn here exactly as cdfo_rdab_prep_rng draws them (attention.hip: Philox4x32-10, counter =
// (pixel, image, channel group, draw), key = seed), so the training and the inference path see the same noise for one seed.
__device__ __forceinline__ void tr_philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                                 unsigned (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
template <bool RNG>
__global__ __launch_bounds__(256) void gumbel_mask_kernel(const float* __restrict__ vmax, const float* __restrict__ noise, long long P,
                                                          long long npix, float* __restrict__ mask, int ldm, unsigned long long seed,
                                                          unsigned draw, float* __restrict__ noise_out) {
  const long long p = blockIdx.x * 64ll + (threadIdx.x >> 2);
  const int part = threadIdx.x & 3;
  if (p >= npix) return;                                   // whole 4-lane groups leave together
  const long long b = p / P, pin = p - b * P;
  float z[16], m = -INFINITY;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    unsigned o[4] = {0u, 0u, 0u, 0u};
    if (RNG) tr_philox4x32_10((unsigned)pin, (unsigned)b, (unsigned)(part * 4 + jj), draw, (unsigned)seed, (unsigned)(seed >> 32), o);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = jj * 4 + k, ch = part * 16 + c;
      float u;
      if (RNG) {
        u = ((float)(o[k] >> 8) + 0.5f) * (1.f / 16777216.f);
        if (noise_out) noise_out[(b * 64 + ch) * P + pin] = u;
      } else {
        u = noise[(b * 64 + ch) * P + pin];
      }
      z[c] = vmax[b * 64 + ch] + (-logf(-logf(u)));
      m = fmaxf(m, z[c]);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 1, 64));
  m = fmaxf(m, __shfl_xor(m, 2, 64));
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) { z[c] = expf(z[c] - m); s += z[c]; }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
#pragma unroll
  for (int c = 0; c < 16; ++c) mask[p * ldm + part * 16 + c] = (z[c] / s >= 0.5f) ? 1.f : 0.f;
}

// ------------------------------------------------------------------------------------------------ 9-tap convolutions
// along the channel axis (directW1_conv, arch.py:2161,2216-2219): out[p][c] = bias + sum_t w[t] in[p][c + t - 4], zero padded.
// flip = 1 uses w[8 - t] (the adjoint: input gradient).
__global__ __launch_bounds__(256) void chanconv9_kernel(const float* __restrict__ in, int ldi, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int flip, long long npix, float* __restrict__ out,
                                                        int ldo) {
  float w9[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w9[t] = w[flip ? 8 - t : t];
  const float bb = bias ? bias[0] : 0.f;
  const long long n = npix * 64;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long p = i >> 6;
    const int c = (int)(i & 63);
    float s = bb;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int cc = c + t - 4;
      if (cc >= 0 && cc < 64) s += w9[t] * in[p * ldi + cc];
    }
    out[p * ldo + c] = s;
  }
}
// weight gradient of a 9-tap convolution: axis 0 = channels, 1 = image rows (directH1_conv, arch.py:2162,2225).
// part[block][t] = sum g[p][c] * in[shifted by t - 4 along the axis][c], part[block][9] = sum g
__global__ __launch_bounds__(256) void corr9_kernel(const float* __restrict__ in, int ldi, const float* __restrict__ g, int ldg, int axis,
                                                    int B, int H, int W, float* __restrict__ part) {
  __shared__ float red[4][10];
  float s[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long long n = (long long)B * H * W * 64;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long p = i >> 6;
    const int c = (int)(i & 63);
    const float gv = g[p * ldg + c];
    s[9] += gv;
    if (axis == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int cc = c + t - 4;
        if (cc >= 0 && cc < 64) s[t] += gv * in[p * ldi + cc];
      }
    } else {
      const int y = (int)((p / W) % H);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t - 4;
        if (yy >= 0 && yy < H) s[t] += gv * in[(p + (long long)(t - 4) * W) * ldi + c];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    const float v = wave_sum(s[t]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 10) part[(long long)blockIdx.x * 10 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------ sequence attention backward
// forward (attention.hip): o_i = sum_j softmax_j(q_i . q_j) v_j over one image row (mode 0), column (1) or 8x8 window (2).
// One workgroup per sequence, four lanes (16 channels each) per element.  Phase A: softmax statistics m_i, l_i of every row
// and D_i = g_i . o_i.  Phase B, element j against every i (s_ij = s_ji):
//   dv_j += p_ij g_i,   dq_j += dS_ij q_i + dS_ji q_i,   p_ij = exp(s - m_i) / l_i,  dS_ij = p_ij (g_i . v_j - D_i).
struct sa_args {
  const float* q; int ldq; const float* v; int ldv; const float* o; int ldo; const float* g; int ldg;
  float* dq; int lddq; float* dv; int lddv;
  int B, H, W, mode;
};
__device__ __forceinline__ long long sa_pixel(const sa_args& a, long long seq, int j) {
  if (a.mode == 0) return seq * a.W + j;                                                   // seq = (b, y)
  if (a.mode == 1) { const long long b = seq / a.W; const int x = (int)(seq - b * a.W); return (b * a.H + j) * a.W + x; }
  const int nwx = a.W >> 3, nwy = a.H >> 3;
  const long long b = seq / (nwx * nwy);
  const int wi = (int)(seq - b * nwx * nwy), wy = wi / nwx, wx = wi - wy * nwx;
  return (b * a.H + wy * 8 + (j >> 3)) * a.W + wx * 8 + (j & 7);
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}
__global__ __launch_bounds__(1024) void seq_attn_bwd_kernel(sa_args a) {
  extern __shared__ __attribute__((aligned(16))) float st[];          // [L][3]: m, l, D
  const int L = a.mode == 0 ? a.W : (a.mode == 1 ? a.H : 64);
  const long long seq = blockIdx.x;
  const int grp = threadIdx.x >> 2, part = threadIdx.x & 3, ngrp = blockDim.x >> 2;
  // ---- phase A
  for (int j0 = 0; j0 < L; j0 += ngrp) {
    const int j = j0 + grp;
    if (j < L) {
      const long long pj = sa_pixel(a, seq, j);
      float qj[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) qj[c] = a.q[pj * a.ldq + part * 16 + c];
      float m = -INFINITY, l = 0.f;
      for (int i = 0; i < L; ++i) {
        const float* qi = a.q + sa_pixel(a, seq, i) * a.ldq + part * 16;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) s += qj[c] * qi[c];
        s = quad_sum(s);
        const float mn = fmaxf(m, s);
        l = l * __expf(m - mn) + __expf(s - mn);
        m = mn;
      }
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) d += a.g[pj * a.ldg + part * 16 + c] * a.o[pj * a.ldo + part * 16 + c];
      d = quad_sum(d);
      if (part == 0) { st[j * 3] = m; st[j * 3 + 1] = l; st[j * 3 + 2] = d; }
    }
  }
  __syncthreads();
  // ---- phase B
  for (int j0 = 0; j0 < L; j0 += ngrp) {
    const int j = j0 + grp;
    if (j >= L) continue;                         // (whole quads leave together; no barrier below)
    const long long pj = sa_pixel(a, seq, j);
    float qj[16], vj[16], gj[16], dq[16], dv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      qj[c] = a.q[pj * a.ldq + part * 16 + c]; vj[c] = a.v[pj * a.ldv + part * 16 + c]; gj[c] = a.g[pj * a.ldg + part * 16 + c];
      dq[c] = 0.f; dv[c] = 0.f;
    }
    const float mj = st[j * 3], lj = st[j * 3 + 1], Dj = st[j * 3 + 2];
    for (int i = 0; i < L; ++i) {
      const long long pi = sa_pixel(a, seq, i);
      const float* qi = a.q + pi * a.ldq + part * 16;
      const float* vi = a.v + pi * a.ldv + part * 16;
      const float* gi = a.g + pi * a.ldg + part * 16;
      float s = 0.f, dPij = 0.f, dPji = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) { s += qj[c] * qi[c]; dPij += gi[c] * vj[c]; dPji += gj[c] * vi[c]; }
      s = quad_sum(s); dPij = quad_sum(dPij); dPji = quad_sum(dPji);
      const float pij = __expf(s - st[i * 3]) / st[i * 3 + 1];          // row i, column j
      const float pji = __expf(s - mj) / lj;                             // row j, column i
      const float dS = pij * (dPij - st[i * 3 + 2]) + pji * (dPji - Dj);
#pragma unroll
      for (int c = 0; c < 16; ++c) { dv[c] += pij * gi[c]; dq[c] += dS * qi[c]; }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) { a.dq[pj * a.lddq + part * 16 + c] = dq[c]; a.dv[pj * a.lddv + part * 16 + c] = dv[c]; }
  }
}

// ------------------------------------------------------------------------------------------------ flow_warp backward
// adjoint of cdfo_flow_warp w.r.t. its input (the motion field gets no gradient: it is an input of the network):
// dx[corner] += w_corner * g[p]   (fp32 atomics; dx zero-filled by the caller)
__global__ __launch_bounds__(256) void flow_warp_bwd_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ mv,
                                                            long long mv_bstride, int B, int H, int W, int C, float* __restrict__ dx, int ldo) {
  const int cgs = C >> 2;
  const long long total = (long long)B * H * W * cgs;
  const float wm = (float)(W - 1 > 1 ? W - 1 : 1), hm = (float)(H - 1 > 1 ? H - 1 : 1);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cg = (int)(i % cgs);
    const long long p = i / cgs;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const long long b = p / ((long long)W * H);
    const float* m = mv + b * mv_bstride;
    const float fx = m[(long long)y * W + x], fy = m[(long long)(H + y) * W + x];
    const float nx = 2.0f * ((float)x + fx) / wm - 1.0f, ny = 2.0f * ((float)y + fy) / hm - 1.0f;
    const float sx = ((nx + 1.f) / 2.f) * (float)(W - 1), sy = ((ny + 1.f) / 2.f) * (float)(H - 1);
    if (!(sx > -1.f && sx < (float)W && sy > -1.f && sy < (float)H)) continue;
    const float x0f = floorf(sx), y0f = floorf(sy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float tx = sx - x0f, ty = sy - y0f;
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + p * ldg + cg * 4);
    float* base = dx + b * H * W * ldo + cg * 4;
    const bool xa = x0 >= 0 && x0 < W, xb = x0 + 1 >= 0 && x0 + 1 < W;
    auto add = [&](int yy, int xx, float w) {
      float* d = base + ((long long)yy * W + xx) * ldo;
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(d + e, gv[e] * w);
    };
    if (y0 >= 0 && y0 < H) {
      if (xa) add(y0, x0, (1.f - tx) * (1.f - ty));
      if (xb) add(y0, x0 + 1, tx * (1.f - ty));
    }
    if (y0 + 1 >= 0 && y0 + 1 < H) {
      if (xa) add(y0 + 1, x0, (1.f - tx) * ty);
      if (xb) add(y0 + 1, x0 + 1, tx * ty);
    }
  }
}

// ------------------------------------------------------------------------------------------------ resample adjoints
// adjoint of bilinear x2, align_corners = False (F.interpolate, arch.py:324-333): out[Y] = 0.25 in[clamp(y - 1)] + 0.75 in[y]
// for Y = 2y, 0.75 in[y] + 0.25 in[clamp(y + 1)] for Y = 2y + 1.  din[y] gathers Y = 2y-1 .. 2y+2 (weights .25 .75 .75 .25;
// the clamped taps of the border rows fold back onto the border).  g: [B][2H][2W][C] -> din: [B][H][W][C]
__device__ __forceinline__ void up2_adj_taps(int y, int H, float (&w)[4]) {
  w[0] = y >= 1 ? 0.25f : 0.f;                 // Y = 2y - 1
  w[1] = 0.75f + (y == 0 ? 0.25f : 0.f);       // Y = 2y
  w[2] = 0.75f + (y == H - 1 ? 0.25f : 0.f);   // Y = 2y + 1
  w[3] = y <= H - 2 ? 0.25f : 0.f;             // Y = 2y + 2
}
__global__ __launch_bounds__(256) void up2_bwd_kernel(const float* __restrict__ g, int ldg, int B, int H, int W, int C, float* __restrict__ din,
                                                      int ldo) {
  const int cq = C >> 2;
  const long long n = (long long)B * H * W * cq;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cq) * 4;
    const long long p = i / cq;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const long long b = p / ((long long)W * H);
    float wy[4], wx[4];
    up2_adj_taps(y, H, wy);
    up2_adj_taps(x, W, wx);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (wy[a] == 0.f) continue;
      const int Y = 2 * y - 1 + a;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (wx[e] == 0.f) continue;
        const int X = 2 * x - 1 + e;
        s += *reinterpret_cast<const f32x4*>(g + ((b * 2 * H + Y) * (2ll * W) + X) * ldg + c) * (wy[a] * wx[e]);
      }
    }
    *reinterpret_cast<f32x4*>(din + p * ldo + c) = s;
  }
}
// adjoint of the 2x2 mean (bilinear x0.5): din[2y + a][2x + b] = g[y][x] / 4.   g: [B][H/2][W/2][C] -> din: [B][H][W][C]
__global__ __launch_bounds__(256) void down2_bwd_kernel(const float* __restrict__ g, int ldg, int B, int H, int W, int C,
                                                        float* __restrict__ din, int ldo) {
  const int cq = C >> 2;
  const long long n = (long long)B * H * W * cq;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cq) * 4;
    const long long p = i / cq;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const long long b = p / ((long long)W * H);
    *reinterpret_cast<f32x4*>(din + p * ldo + c) =
        *reinterpret_cast<const f32x4*>(g + ((b * (H >> 1) + (y >> 1)) * (long long)(W >> 1) + (x >> 1)) * ldg + c) * 0.25f;
  }
}

}  // namespace

// =================================================================================================== C-ABI
extern "C" long long cdfo_conv_wgrad_slab_floats(int A, int Bc, int ks, int nsplit) {
  if (A <= 0 || Bc <= 0 || ks <= 0 || nsplit <= 0) return -1;
  return (long long)nsplit * ks * ks * ((A + 63) / 64 * 64) * ((Bc + 63) / 64 * 64);
}

extern "C" int cdfo_conv_wgrad_prec(const float* S, int lds_, int A, const float* L, int ldl, int Bc, int N, int Hs, int Ws, int Hl,
                                    int Wl, int ks, int stride, int pad, int nsplit, float* slab, float* dw, int Btot, int b_off,
                                    int prec, void* stream);
extern "C" int cdfo_conv_wgrad(const float* S, int lds_, int A, const float* L, int ldl, int Bc, int N, int Hs, int Ws, int Hl,
                               int Wl, int ks, int stride, int pad, int nsplit, float* slab, float* dw, int Btot, int b_off,
                               void* stream) {
  return cdfo_conv_wgrad_prec(S, lds_, A, L, ldl, Bc, N, Hs, Ws, Hl, Wl, ks, stride, pad, nsplit, slab, dw, Btot, b_off, CDFO_PREC_F32, stream);
}

extern "C" int cdfo_conv_wgrad_prec(const float* S, int lds_, int A, const float* L, int ldl, int Bc, int N, int Hs, int Ws, int Hl,
                                    int Wl, int ks, int stride, int pad, int nsplit, float* slab, float* dw, int Btot, int b_off,
                                    int prec, void* stream) {
  if (!S || !L || !slab || !dw || A <= 0 || Bc <= 0 || N <= 0 || Hs <= 0 || Ws <= 0 || Hl <= 0 || Wl <= 0 || ks <= 0 || ks > 9 ||
      stride <= 0 || nsplit <= 0 || nsplit > 65535 || lds_ < A || ldl < Bc || Btot < b_off + Bc)
    return CDFO_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  wg_args g{S, lds_, A, L, ldl, Bc, N, Hs, Ws, Hl, Wl, ks, stride, pad, slab, nsplit, (A + 63) / 64, (Bc + 63) / 64};
  CdfoProfScope prof(st, KID_CONV3_NARROW, 2.0 * N * Hs * Ws * (double)A * Bc * ks * ks, 4.0 * N * ((double)Hs * Ws * A + (double)Hl * Wl * Bc));
  if (prec != CDFO_PREC_F32 && prec != CDFO_PREC_BF16X3) return CDFO_EINVAL;
  if (prec == CDFO_PREC_BF16X3) hipLaunchKernelGGL(conv_wgrad_bf16x3_kernel, dim3(g.a_blocks * g.b_blocks, ks * ks, nsplit), dim3(256), 0, st, g);
  else hipLaunchKernelGGL(conv_wgrad_kernel, dim3(g.a_blocks * g.b_blocks, ks * ks, nsplit), dim3(256), 0, st, g);
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(tr_grid((long long)A * Bc * ks * ks)), dim3(256), 0, st, slab, nsplit, ks * ks, A, Bc,
                     g.a_blocks * 64, g.b_blocks * 64, Btot, b_off, dw);
  CDFO_LAUNCH_CHECK();
  return 0;
}

// out[img][c] = scale * sum_p a[img][p][c] * (b ? b[img][p][c] : 1);  part: nimg * nchunk * C floats of scratch
extern "C" int cdfo_coldot(const float* a, int lda, const float* b, int ldb, int nimg, long long P, int C, int nchunk, float scale,
                           float* part, float* out, void* stream) {
  if (!a || !part || !out || nimg <= 0 || P <= 0 || C <= 0 || nchunk <= 0 || nchunk > 65535 || nimg > 65535 || lda < C || (b && ldb < C))
    return CDFO_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(coldot_partial_kernel, dim3(nchunk, (C + 63) / 64, nimg), dim3(256), 0, st, a, lda, b, ldb, P, C, nchunk, part);
  hipLaunchKernelGGL(coldot_finish_kernel, dim3((nimg * C + 3) / 4), dim3(256), 0, st, part, nchunk, C, nimg, scale, out);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_ew(const float* a, int lda, const float* b, int ldb, long long rows, int C, int mode, int aux, float scale,
                       long long P, float* out, int ldo, void* stream) {
  if (!a || !out || rows <= 0 || C <= 0 || C % 4 || mode < 0 || mode > 4 || (mode != 4 && (!b || lda % 4 || ldb % 4 || lda < C || ldb < C)) ||
      ldo % 4 || ldo < C || (mode == 4 && (P <= 0 || lda % 4)))
    return CDFO_EINVAL;
  if (!aligned16(a) || (b && !aligned16(b)) || !aligned16(out)) return CDFO_EALIGN;
  hipLaunchKernelGGL(ew_kernel, dim3(tr_grid(rows * (C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), a, lda, b, ldb, rows, C, mode, aux,
                     scale, P > 0 ? P : 1, out, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}

// part: nblk * 128 floats; after the call sum part[:, 0, :] -> dgamma, part[:, 1, :] -> dbeta (fixed order)
extern "C" int cdfo_layernorm64_bwd(const float* x, int ldx, const float* g, int ldg, const float* gamma, long long npix, float* dx,
                                    int ldo, float* part, int nblk, void* stream) {
  if (!x || !g || !gamma || !dx || !part || npix <= 0 || nblk <= 0 || ldx % 4 || ldg % 4 || ldo % 4 || ldx < 64 || ldg < 64 || ldo < 64)
    return CDFO_EINVAL;
  if (!aligned16(x) || !aligned16(g) || !aligned16(gamma) || !aligned16(dx)) return CDFO_EALIGN;
  hipLaunchKernelGGL(layernorm64_bwd_kernel, dim3(nblk), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, g, ldg, gamma, npix, dx, ldo, part);
  CDFO_LAUNCH_CHECK();
  return 0;
}

// part: nblk * 9 * C floats; dw[c][tap] = sum_blk part[blk][tap][c]
extern "C" int cdfo_dwconv3x3_wgrad(const float* x, int ldx, const float* g, int ldg, int B, int H, int W, int C, float* part, int nblk,
                                    void* stream) {
  if (!x || !g || !part || B <= 0 || H <= 0 || W <= 0 || C <= 0 || nblk <= 0 || ldx < C || ldg < C) return CDFO_EINVAL;
  hipLaunchKernelGGL(dwconv3x3_wgrad_kernel, dim3(nblk, (C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, g, ldg, B, H, W, C, part);
  CDFO_LAUNCH_CHECK();
  return 0;
}

// spatial gate backward: scratch = B*H*W*5 floats (pooled [.,4] + ds); dt [.,16] assigned; dw99 = 98 weights + bias
extern "C" int cdfo_spatial_gate16_bwd(const float* t, int ldt, const float* g, int ldg, const float* w, const float* bias, int B, int H,
                                       int W, float* scratch, float* dt, int ldo, float* dw99, void* stream) {
  if (!t || !g || !w || !bias || !scratch || !dt || !dw99 || B <= 0 || H <= 0 || W <= 0 || ldt < 16 || ldg < 16 || ldo < 16) return CDFO_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long npix = (long long)B * H * W;
  float* pooled = scratch;
  float* ds = scratch + npix * 4;
  hipLaunchKernelGGL(sg16_pool_kernel, dim3(tr_grid(npix)), dim3(256), 0, st, t, ldt, npix, pooled);
  hipLaunchKernelGGL(sg16_ds_kernel, dim3(tr_grid(npix)), dim3(256), 0, st, t, ldt, g, ldg, pooled, w, bias, B, H, W, ds, dt, ldo);
  hipLaunchKernelGGL(sg16_fin_kernel, dim3(tr_grid(npix)), dim3(256), 0, st, pooled, ds, w, B, H, W, dt, ldo);
  hipLaunchKernelGGL(sg16_wgrad_kernel, dim3(99), dim3(256), 0, st, pooled, ds, B, H, W, dw99);
  CDFO_LAUNCH_CHECK();
  return 0;
}

// noise != NULL: the uniforms [B][64][P] are given; noise == NULL: drawn in the kernel (seed, draw), optionally copied to noise_out
extern "C" int cdfo_gumbel_mask(const float* vmax, const float* noise, long long seed, int draw, float* noise_out, int B, long long P,
                                float* mask, int ldm, void* stream) {
  if (!vmax || !mask || B <= 0 || P <= 0 || P >= (1ll << 32) || ldm < 64) return CDFO_EINVAL;
  const long long npix = (long long)B * P;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (noise)
    hipLaunchKernelGGL(gumbel_mask_kernel<false>, dim3((unsigned)((npix + 63) / 64)), dim3(256), 0, st, vmax, noise, P, npix, mask, ldm, 0ull, 0u, nullptr);
  else
    hipLaunchKernelGGL(gumbel_mask_kernel<true>, dim3((unsigned)((npix + 63) / 64)), dim3(256), 0, st, vmax, nullptr, P, npix, mask, ldm,
                       (unsigned long long)seed, (unsigned)draw, noise_out);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_chanconv9(const float* in, int ldi, const float* w9, const float* bias, int flip, long long npix, float* out, int ldo,
                              void* stream) {
  if (!in || !w9 || !out || npix <= 0 || ldi < 64 || ldo < 64) return CDFO_EINVAL;
  hipLaunchKernelGGL(chanconv9_kernel, dim3(tr_grid(npix * 64)), dim3(256), 0, static_cast<hipStream_t>(stream), in, ldi, w9, bias, flip, npix, out, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}

// part: nblk * 10 floats (9 taps + the bias gradient), summed by the caller
extern "C" int cdfo_corr9(const float* in, int ldi, const float* g, int ldg, int axis, int B, int H, int W, float* part, int nblk,
                          void* stream) {
  if (!in || !g || !part || B <= 0 || H <= 0 || W <= 0 || nblk <= 0 || (axis != 0 && axis != 1) || ldi < 64 || ldg < 64) return CDFO_EINVAL;
  hipLaunchKernelGGL(corr9_kernel, dim3(nblk), dim3(256), 0, static_cast<hipStream_t>(stream), in, ldi, g, ldg, axis, B, H, W, part);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_seq_attn_bwd(const float* q, int ldq, const float* v, int ldv, const float* o, int ldo, const float* g, int ldg,
                                 float* dq, int lddq, float* dv, int lddv, int B, int H, int W, int mode, void* stream) {
  if (!q || !v || !o || !g || !dq || !dv || B <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 2) return CDFO_EINVAL;
  if (ldq < 64 || ldv < 64 || ldo < 64 || ldg < 64 || lddq < 64 || lddv < 64) return CDFO_EINVAL;
  if (mode == 2 && ((H | W) & 7)) return CDFO_EINVAL;
  const int L = mode == 0 ? W : (mode == 1 ? H : 64);
  if (L > 4096) return CDFO_EINVAL;
  const long long nseq = mode == 0 ? (long long)B * H : (mode == 1 ? (long long)B * W : (long long)B * (H >> 3) * (W >> 3));
  if (nseq >= (1ll << 31)) return CDFO_EINVAL;
  int threads = ((L * 4 + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  sa_args a{q, ldq, v, ldv, o, ldo, g, ldg, dq, lddq, dv, lddv, B, H, W, mode};
  CdfoProfScope prof(static_cast<hipStream_t>(stream), mode == 0 ? KID_ATTN_ROW : (mode == 1 ? KID_ATTN_COL : KID_ATTN_WIN),
                     8.0 * 64 * (double)nseq * L * L, 4.0 * 6 * 64 * (double)B * H * W);
  hipLaunchKernelGGL(seq_attn_bwd_kernel, dim3((unsigned)nseq), dim3(threads), (size_t)L * 3 * sizeof(float), static_cast<hipStream_t>(stream), a);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_flow_warp_bwd(const float* g, int ldg, const float* mv, long long mv_bstride, int B, int H, int W, int C, float* dx,
                                  int ldo, void* stream) {
  if (!g || !mv || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || ldg % 4 || ldo % 4 || ldg < C || ldo < C) return CDFO_EINVAL;
  if (!aligned16(g) || !aligned16(dx)) return CDFO_EALIGN;
  hipLaunchKernelGGL(flow_warp_bwd_kernel, dim3(tr_grid((long long)B * H * W * (C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), g, ldg, mv,
                     mv_bstride, B, H, W, C, dx, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}

// up = 1: g [B][2H][2W][C] -> din [B][H][W][C] (adjoint of bilinear x2);  up = 0: g [B][H/2][W/2][C] -> din [B][H][W][C]
extern "C" int cdfo_resample2_bwd(const float* g, int ldg, int B, int H, int W, int C, int up, float* din, int ldo, void* stream) {
  if (!g || !din || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || ldg % 4 || ldo % 4 || ldg < C || ldo < C || (!up && ((H | W) & 1)))
    return CDFO_EINVAL;
  if (!aligned16(g) || !aligned16(din)) return CDFO_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (up) hipLaunchKernelGGL(up2_bwd_kernel, dim3(tr_grid((long long)B * H * W * (C / 4))), dim3(256), 0, st, g, ldg, B, H, W, C, din, ldo);
  else hipLaunchKernelGGL(down2_bwd_kernel, dim3(tr_grid((long long)B * H * W * (C / 4))), dim3(256), 0, st, g, ldg, B, H, W, C, din, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}
