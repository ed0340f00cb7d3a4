// Prior-fusion attention `LLongRangAttention` (arch/SIDECVSR_our.py:2179-2249): residual-driven hard mask, row /
// column long-range attention and 8x8 window attention -- without ever writing a WxW / HxH / 64x64 score map to HBM.
//
//   rdab_prep   : Gumbel-softmax hard mask (arch.py:2168-2195) from the injected uniform noise; q,v split of the
//                 128-channel input_conv output; sparse_q = conv1x9_over_channels(mask*q), v' = conv1x9(v)
//                 (directW1_conv, arch.py:2216-2219); window query (1-mask)*q (arch.py:2235-2238).
//   colconv9    : directH1_conv, the 9-tap conv along H on sparse_q (arch.py:2225).
//   seq_attn    : softmax(Q Q^T) V over a row, a column or an 8x8 window, one query per lane, keys streamed through
//                 wave-uniform (scalar) loads, online softmax in registers.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int ZS = 65;  // LDS pitch for the [pixel][channel] logits (conflict-free both ways)
constexpr int QS = 72;  // 4 zero floats on both sides of the 64 channels for the 9-tap channel conv

// Philox4x32-10 (Salmon et al., SC'11): counter-based generator, 4 x 32 random bits per (counter, key).
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
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

// RNG: the uniform draws of gumbel_softmax (arch.py:2169, torch.rand_like + "redraw while any == 0") are generated here
// instead of being read: u = (24 random bits + 0.5) * 2^-24 lies strictly inside (0, 1), element (image b, channel c,
// pixel p) takes word c & 3 of Philox(counter = (p, b, c >> 2, draw), key = seed).  noise_out (optional): the drawn
// values, [B][64][P], for callers that replay the forward elsewhere (parity tests).
template <bool RNG>
__global__ __launch_bounds__(256) void rdab_prep_kernel(const float* __restrict__ xq, int ldx,  // [.,128]: q | v
                                                        const float* __restrict__ vmax,         // [B][64]
                                                        const float* __restrict__ noise,        // [B][64][P] (NCHW)
                                                        const float* __restrict__ wW, const float* __restrict__ bW,
                                                        long long P, float* __restrict__ sq, int lds_,
                                                        float* __restrict__ vrow, int ldv, float* __restrict__ qwin,
                                                        int ldw, unsigned long long seed, unsigned draw,
                                                        float* __restrict__ noise_out,
                                                        const unsigned long long* __restrict__ seed_dev) {
  __shared__ float z[64 * ZS];
  __shared__ float mq[64 * QS];
  __shared__ float vv[64 * QS];
  __shared__ float ev[64];
  const int tid = threadIdx.x;
  const long long p0 = (long long)blockIdx.x * 64;  // P % 64 == 0, so the 64 pixels share one image
  const long long b = p0 / P, pin = p0 - b * P;
  // softmax_c(v_c + g_c) with g = -log(-log u)  ==  (E_c / L_c) / sum_j (E_j / L_j),  E_c = exp(v_c - max v), L_c = -log u_c > 0:
  // ONE logarithm per element instead of two logarithms and an exponential (the kernel is bound by exactly this arithmetic and by
  // the Philox rounds), E is 64 values per image.  Same quantity as arch.py:2168-2177 up to fp32 rounding.  (The v_log_f32 /
  // v_rcp_f32 forms of the logarithm and the quotient were measured too: no change, 0.3835 ms per launch either way -- the kernel
  // moves 1.34 GB in that time, 3.5 TB/s stand-alone.)
  if (tid < 64) {
    float vm = vmax[b * 64];
    for (int c = 1; c < 64; ++c) vm = fmaxf(vm, vmax[b * 64 + c]);
    ev[tid] = expf(vmax[b * 64 + tid] - vm);
  }
  __syncthreads();
  // phase 1: weights w[i][c] = E_c / (-log u)   (coalesced along pixels)
  if (RNG) {
    if (seed_dev) seed = *seed_dev;                // key in device memory (graph replays: rewritten between replays)
    const int i = tid & 63;
    for (int j = tid >> 6; j < 16; j += 4) {       // channel group j = channels 4j .. 4j+3 of pixel i
      unsigned o[4];
      philox4x32_10((unsigned)(pin + i), (unsigned)b, (unsigned)j, draw, (unsigned)seed, (unsigned)(seed >> 32), o);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = 4 * j + k;
        const float u = ((float)(o[k] >> 8) + 0.5f) * (1.f / 16777216.f);
        if (noise_out) noise_out[(b * 64 + c) * P + pin + i] = u;
        z[i * ZS + c] = ev[c] / (-logf(u));
      }
    }
  } else {
    const int i = tid & 63;
    for (int c = tid >> 6; c < 64; c += 4) {
      const float u = noise[(b * 64 + c) * P + pin + i];
      z[i * ZS + c] = ev[c] / (-logf(u));
    }
  }
  for (int i = tid; i < 64 * QS; i += 256) { mq[i] = 0.f; vv[i] = 0.f; }
  __syncthreads();
  // phase 2: hard mask = softmax_c >= 0.5  <=>  w_c >= 0.5 * sum_j w_j ; 4 lanes per pixel, 16 channels each
  const int i = tid >> 2, part = tid & 3;
  {
    float e[16], s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) { e[c] = z[i * ZS + part * 16 + c]; s += e[c]; }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const float* px = xq + (p0 + i) * ldx + part * 16;
    float* pw = qwin + (p0 + i) * ldw + part * 16;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const f32x4 q4 = *reinterpret_cast<const f32x4*>(px + c4 * 4);
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(px + 64 + c4 * 4);
      f32x4 w4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float mk = (e[c4 * 4 + k] >= 0.5f * s) ? 1.f : 0.f;
        mq[i * QS + 4 + part * 16 + c4 * 4 + k] = mk * q4[k];
        vv[i * QS + 4 + part * 16 + c4 * 4 + k] = v4[k];
        w4[k] = (1.f - mk) * q4[k];
      }
      *reinterpret_cast<f32x4*>(pw + c4 * 4) = w4;
    }
  }
  __syncthreads();
  // phase 3: 9-tap cross-correlation along the channel axis (zero padded), + bias
  {
    float w9[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w9[t] = wW[t];
    const float bb = bW[0];
    float* ps = sq + (p0 + i) * lds_ + part * 16;
    float* pv = vrow + (p0 + i) * ldv + part * 16;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      f32x4 a, bq;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = part * 16 + c4 * 4 + k;
        float s1 = bb, s2 = bb;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          s1 += w9[t] * mq[i * QS + c + t];
          s2 += w9[t] * vv[i * QS + c + t];
        }
        a[k] = s1;
        bq[k] = s2;
      }
      *reinterpret_cast<f32x4*>(ps + c4 * 4) = a;
      *reinterpret_cast<f32x4*>(pv + c4 * 4) = bq;
    }
  }
}

// directH1_conv (arch.py:2162, 2225): 9 taps along the image rows.  A thread owns one (image, row segment, column, 4-channel
// group) and slides down its segment with the nine input rows of the current output in registers: every input element is
// read once per segment (the rows of a segment's 4 + 4 halo twice), coalesced across columns / channel groups.
constexpr int CC9_SEG = 34;     // output rows per thread
__global__ __launch_bounds__(256) void colconv9_kernel(const float* __restrict__ in, int ldi,
                                                       const float* __restrict__ wH, const float* __restrict__ bH,
                                                       int B, int H, int W, int nseg, float* __restrict__ out, int ldo) {
  float w9[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w9[t] = wH[t];
  const float bb = bH[0];
  const long long total = (long long)B * nseg * W * 16;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cg = idx & 15;
    long long r = idx >> 4;
    const int x = (int)(r % W); r /= W;
    const int seg = (int)(r % nseg);
    const long long b = r / nseg;
    const int y0 = seg * CC9_SEG, y1 = min(H, y0 + CC9_SEG);
    const float* col = in + (b * H * W + x) * ldi + cg * 4;
    float* ocol = out + (b * H * W + x) * ldo + cg * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 win[9];                                        // win[t] = in[y + t - 4]
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int yy = y0 + t - 4;
      win[t + 1] = (yy >= 0 && yy < H) ? *reinterpret_cast<const f32x4*>(col + (long long)yy * W * ldi) : z;
    }
    for (int y = y0; y < y1; ++y) {
#pragma unroll
      for (int t = 0; t < 8; ++t) win[t] = win[t + 1];
      const int yy = y + 4;
      win[8] = yy < H ? *reinterpret_cast<const f32x4*>(col + (long long)yy * W * ldi) : z;
      f32x4 acc = {bb, bb, bb, bb};
#pragma unroll
      for (int t = 0; t < 9; ++t) acc += w9[t] * win[t];
      *reinterpret_cast<f32x4*>(ocol + (long long)y * W * ldo) = acc;
    }
  }
}

// MODE 0: sequence = image row (keys along W); 1: image column (keys along H); 2: 8x8 window.
template <int MODE>
__global__ __launch_bounds__(64) void seq_attn_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ v,
                                                      int ldv, float* __restrict__ out, int ldo, int B, int H, int W) {
  const int lane = threadIdx.x;
  long long kbase;      // pixel index of key 0
  long long kstep;      // pixel stride between consecutive keys (MODE 0/1)
  int L;                // number of keys
  long long qpix;       // this lane's query pixel
  bool active = true;
  if (MODE == 0) {
    const int nb = (W + 63) / 64;
    const int blk = blockIdx.x % nb;
    const long long row = blockIdx.x / nb;  // b*H + h
    kbase = row * W; kstep = 1; L = W;
    int x = blk * 64 + lane;
    if (x >= W) { x = W - 1; active = false; }
    qpix = kbase + x;
  } else if (MODE == 1) {
    const int nb = (H + 63) / 64;
    const int blk = blockIdx.x % nb;
    const long long col = blockIdx.x / nb;  // b*W + w
    const long long b = col / W, w = col - b * W;
    kbase = b * H * W + w; kstep = W; L = H;
    int y = blk * 64 + lane;
    if (y >= H) { y = H - 1; active = false; }
    qpix = kbase + (long long)y * W;
  } else {
    const int nwx = W >> 3, nwy = H >> 3;
    const int wx = blockIdx.x % nwx, wy = (blockIdx.x / nwx) % nwy;
    const long long b = blockIdx.x / (nwx * nwy);
    kbase = (b * H + wy * 8) * W + wx * 8; kstep = 0; L = 64;
    qpix = kbase + (long long)(lane >> 3) * W + (lane & 7);
  }
  float qr[64], o[64];
#pragma unroll
  for (int c4 = 0; c4 < 16; ++c4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(q + qpix * ldq + c4 * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { qr[c4 * 4 + e] = t[e]; o[c4 * 4 + e] = 0.f; }
  }
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < L; ++j) {
    const long long kp = (MODE == 2) ? kbase + (long long)(j >> 3) * W + (j & 7) : kbase + (long long)j * kstep;
    const float* __restrict__ kr = q + kp * ldq;  // wave-uniform address -> scalar loads
    const float* __restrict__ vr = v + kp * ldv;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int c = 0; c < 64; c += 4) {
      s0 = fmaf(qr[c], kr[c], s0);
      s1 = fmaf(qr[c + 1], kr[c + 1], s1);
      s2 = fmaf(qr[c + 2], kr[c + 2], s2);
      s3 = fmaf(qr[c + 3], kr[c + 3], s3);
    }
    const float s = (s0 + s1) + (s2 + s3);
    if (s > m) {  // new running max: rescale what has been accumulated so far
      const float alpha = expf(m - s);
      l *= alpha;
#pragma unroll
      for (int c = 0; c < 64; ++c) o[c] *= alpha;
      m = s;
    }
    const float pj = expf(s - m);
    l += pj;
#pragma unroll
    for (int c = 0; c < 64; ++c) o[c] = fmaf(pj, vr[c], o[c]);
  }
  if (active) {
    const float inv = 1.f / l;
    float* po = out + qpix * ldo;
#pragma unroll
    for (int c4 = 0; c4 < 16; ++c4) {
      f32x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = o[c4 * 4 + e] * inv;
      *reinterpret_cast<f32x4*>(po + c4 * 4) = t;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Row / column attention on the matrix cores, flash style.
// One wave owns 32 queries; 4 waves (128 queries of ONE sequence) share the key/value tiles staged in LDS.
//   S^T[key][query] = K Q^T : A = K (rows = keys), B = Q^T, on the fp16 matrix cores with BOTH operands split into
//                     fp16 hi + fp16 lo (22 significant bits; hi*hi + lo*hi + hi*lo, the 2^-22 lo*lo term dropped):
//                     12 v_mfma_f32_32x32x16_f16 (384 cycles) per 32x32 tile instead of 32 v_mfma_f32_32x32x2_f32
//                     (2048 cycles), scores exact to ~1e-6 relative.  Lane (r, h) keeps Q[r][16s + 8h .. +7] (s = 0..3)
//                     in registers and reads the same channels of K[r] from LDS (hi | lo halves of a 272-byte row).
//   softmax          : the accumulator holds, per lane, 16 keys of ITS query (column) -> the running max needs one
//                     cross-half shuffle, the exponentials are lane-local.
//   O^T[ch][query]  += V^T P^T, also fp16 hi/lo x 3 passes: the probabilities are already the B operand -- register
//                     8u + j of half h is key 16u + 4h + 8(j>>2) + (j&3) of query r, and the key order inside a dot
//                     product is free, so V is staged TRANSPOSED ([channel][key slot], slot = 16u + 8h + j) in that
//                     same order: a staging thread loads 4 keys x 4 channels, transposes 4x4 in registers and writes
//                     8-byte runs of 4 keys; no data movement between the two products.
typedef _Float16 attn_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 attn_f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned attn_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned attn_u32x4 __attribute__((ext_vector_type(4)));

// Two fp32 values -> packed fp16 hi (round to nearest) and packed fp16 lo = fp16(v - hi), four instructions per pair:
// v_cvt_pk_f16_f32, two v_fma_mix_f32 (f32 * 1.0 - f16 -> f32: the exact remainder, the fp16 operand read straight from its
// half of the packed register), v_cvt_pk_f16_f32.  hipcc emits cvt_f32_f16 + sub per element for the plain C expression (six
// per pair) and folds a source-level fma back into it; the kernel is bound by its vector instructions, so
// the form is spelled out.  Full-register results only: the three-instruction form through v_fma_mixlo_f16 / v_fma_mixhi_f16
// writes half registers, and gfx950 needs a wait state between such a write and the next vector read of the register, which
// hipcc cannot insert around inline assembly (measured: wrong window-attention results where the consumer followed directly).
__device__ __forceinline__ void split_pair_f16(float a, float b, unsigned& hi, unsigned& lo) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 hv = {(_Float16)a, (_Float16)b};
  hi = __builtin_bit_cast(unsigned, hv);
  float ra, rb;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "v"(hi));
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "v"(hi));
  const h2 lv = {(_Float16)ra, (_Float16)rb};
  lo = __builtin_bit_cast(unsigned, lv);
}

// NW waves per workgroup = 32*NW queries of one sequence at most; a stage is 8*NW keys (NW/4 sub-tiles of 32), staged by
// all NW*64 threads with one 4-key x 4-channel unit each -- the larger the workgroup, the less staging work (loads,
// fp32 -> fp16 hi/lo splits, LDS writes) per query.  The query tiles of a sequence are spread evenly over its workgroups
// (tpb tiles each; waves >= tpb only stage).  The kernel is VALU-bound (softmax + splits), not MFMA-bound, hence:
// scores in the log2 domain (Q is scaled by log2 e once, p = v_exp_f32(s - m) -- one instruction instead of expf's ten),
// key masking only in a sequence's last sub-tile, and the accumulator rescale skipped while no lane's running maximum moves.
// PV1 (the default fp16x2 arithmetic of the forward): the SECOND product with single-fp16 operands -- probabilities p in [0, 1]
// (sum 1) and values rounded once to fp16: |error of an output| <= 2^-11 * sum_j p_j |v_j| <= 4.9e-4 * max|v| in the worst case,
// ~1e-5 * |v| measured (random signs average); the scores, whose error the exponential amplifies, keep all three passes.  8 of the
// 24 MFMAs per 32 x 32 tile, the lo halves of V's staging and of P's conversion, and half of V's fragment reads go away.
template <int MODE, int NW, bool PV1 = false>   // MODE 0: sequence = image row, 1: image column, 2: 8x8 window (row-major inside the window)
__global__ __launch_bounds__(NW * 64, NW == 4 ? 3 : 2) void seq_attn_mfma_kernel(const float* __restrict__ q, int ldq,
                                                                const float* __restrict__ v, int ldv,
                                                                float* __restrict__ out, int ldo, int B, int H, int W,
                                                                int nb, int tpb, int xcd_map) {
  constexpr int KROW = 272;                                 // bytes per staged key row: 128 B hi | 128 B lo | 16 B pad
  constexpr int NT = NW * 64, SUB = NW / 4, KT = 32 * SUB, KQ = KT / 4;
  constexpr int K_BYTES = KT * KROW, V_BYTES = SUB * 64 * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
  unsigned char* const sK = attn_smem;                      // [2][KT][KROW]
  // V^T per sub-tile: row = channel (128 B: four 16-byte hi slots 2u+h, four lo slots 4+2u+h), slot index XORed with
  // (ch>>1)&7 so that the 16 channels of a ds_read_b128 lane group hit 16 distinct 16-byte slots
  unsigned char* const sV = attn_smem + 2 * K_BYTES;        // [2][SUB][64][128]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, r = lane & 31;
  const int L = MODE == 0 ? W : (MODE == 1 ? H : 64);
  // The nb workgroups of a sequence each stage ALL of its keys / values: give them consecutive slots of ONE XCD (workgroup i
  // runs on XCD i % 8), so that they run at about the same time behind the same L2 and the sequence leaves HBM once
  unsigned bid = blockIdx.x;
  if (xcd_map) {
    const unsigned q8 = gridDim.x >> 3, r8 = gridDim.x & 7, xcd = bid & 7, idx = bid >> 3;
    bid = xcd * q8 + (xcd < r8 ? xcd : r8) + idx;
  }
  const int blk = bid % nb;
  const long long seq = bid / nb;
  long long kbase;
  if (MODE == 0) kbase = seq * W;                           // seq = b*H + y
  else if (MODE == 1) { const long long b = seq / W, x = seq - b * W; kbase = b * H * W + x; }
  else {                                                    // seq = (b*(H/8) + wy)*(W/8) + wx
    const int wpr = W >> 3;
    const long long t = seq / wpr;
    const int wx = (int)(seq - t * wpr);
    kbase = (t * 8) * W + wx * 8;                           // t*8 = b*H + wy*8
  }
  auto pix_of = [&](int i) -> long long {                   // pixel of element i of the sequence
    return MODE == 0 ? kbase + i : (MODE == 1 ? kbase + (long long)i * W : kbase + (long long)(i >> 3) * W + (i & 7));
  };
  const int q0 = (blk * tpb + wave) * 32;                   // this wave's first query
  const bool wave_active = wave < tpb && q0 < L;
  int qi = q0 + r;
  const bool q_ok = qi < L;
  if (qi >= L) qi = L - 1;
  const long long qpix = pix_of(qi);

  attn_f16x8 qh[4], ql[4];                                  // log2(e) * Q[r][16s + 8h + j], fp16 hi / lo
  constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(q + qpix * ldq + 16 * s + 8 * h) * LOG2E;
    const f32x4 t1 = *reinterpret_cast<const f32x4*>(q + qpix * ldq + 16 * s + 8 * h + 4) * LOG2E;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qh[s][e] = (_Float16)t0[e];      ql[s][e] = (_Float16)(t0[e] - (float)qh[s][e]);
      qh[s][4 + e] = (_Float16)t1[e];  ql[s][4 + e] = (_Float16)(t1[e] - (float)qh[s][4 + e]);
    }
  }
  f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
  float m = -INFINITY, l = 0.f;                             // running maximum (log2 domain) and sum of this half-wave's keys

  // staging (4 float4 loads per thread and stage): the first NT/2 threads take V -- keys 4sm..4sm+3 x channels 4c..4c+3
  // each --, the others K -- keys sm + KQ*s (s = 0..3) x channels 4c..4c+3
  const bool st_v = tid < NT / 2;
  const int sj = st_v ? tid : tid - NT / 2, sm = sj >> 4, sc4 = sj & 15;
  const float* const sbase = (st_v ? v : q) + sc4 * 4;
  const int sld = st_v ? ldv : ldq;
  f32x4 rs[4];
  auto load_tile = [&](int t0) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      int key = t0 + (st_v ? 4 * sm + s : sm + KQ * s);
      key = key < L ? key : L - 1;                          // clamped (masked below), always loaded
      rs[s] = *reinterpret_cast<const f32x4*>(sbase + pix_of(key) * sld);
    }
  };
  auto write_tile = [&](int buf) {
    if (st_v) {
      // keys 4m..4m+3 of sub-tile sm>>3 = slots 16u + 8hh + 4g + (0..3) with m = sm&7, u = m>>2, hh = m&1, g = (m&3)>>1
      const int mm = sm & 7, c = 2 * (mm >> 2) + (mm & 1), g = (mm & 3) >> 1;
      unsigned char* const vb = sV + buf * V_BYTES + (sm >> 3) * 8192 + g * 8;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ch = 4 * sc4 + e, swz = (ch >> 1) & 7;
        if (PV1) {                                          // keys 4m .. 4m+3 of channel ch, rounded once
          const attn_f16x4 vh = {(_Float16)rs[0][e], (_Float16)rs[1][e], (_Float16)rs[2][e], (_Float16)rs[3][e]};
          *reinterpret_cast<attn_f16x4*>(vb + ch * 128 + ((c ^ swz) * 16)) = vh;
        } else {
          unsigned h0, l0, h1, l1;
          split_pair_f16(rs[0][e], rs[1][e], h0, l0);
          split_pair_f16(rs[2][e], rs[3][e], h1, l1);
          const attn_u32x2 vh = {h0, h1}, vl = {l0, l1};
          *reinterpret_cast<attn_u32x2*>(vb + ch * 128 + ((c ^ swz) * 16)) = vh;
          *reinterpret_cast<attn_u32x2*>(vb + ch * 128 + (((4 + c) ^ swz) * 16)) = vl;
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        unsigned h0, l0, h1, l1;
        split_pair_f16(rs[s][0], rs[s][1], h0, l0);
        split_pair_f16(rs[s][2], rs[s][3], h1, l1);
        const attn_u32x2 kh = {h0, h1}, kl = {l0, l1};
        unsigned char* row = sK + buf * K_BYTES + (sm + KQ * s) * KROW + sc4 * 8;
        *reinterpret_cast<attn_u32x2*>(row) = kh;
        *reinterpret_cast<attn_u32x2*>(row + 128) = kl;
      }
    }
  };

  const int nstages = (L + KT - 1) / KT;
  load_tile(0);
  write_tile(0);
  __syncthreads();
  for (int t = 0; t < nstages; ++t) {
    const int buf = t & 1;
    if (t + 1 < nstages) load_tile((t + 1) * KT);
    if (wave_active) {
      // The stage's SUB sub-tiles of 32 keys share ONE softmax update: all their score products first (independent
      // accumulators, back to back on the matrix pipe), one running-maximum step for the 64 keys, then all the V^T P^T products
      // -- half the rescale tests and longer MFMA runs for the vector work of the SIMD's other wave to hide behind
      const int kv_stage = L - t * KT;                      // keys of this stage that exist (> 0)
      const int nsub = kv_stage >= KT ? SUB : (kv_stage + 31) >> 5;
      f32x16 sacc[SUB];
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        if (sub >= nsub) break;
        // ---- S^T = K Q^T (log2 domain)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[sub][e] = 0.f;
        const unsigned char* const kb = sK + buf * K_BYTES + (sub * 32 + r) * KROW + 16 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const attn_f16x8 kh = *reinterpret_cast<const attn_f16x8*>(kb + 32 * s);
          const attn_f16x8 kl = *reinterpret_cast<const attn_f16x8*>(kb + 32 * s + 128);
          sacc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], sacc[sub], 0, 0, 0);
          sacc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], sacc[sub], 0, 0, 0);
          sacc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], sacc[sub], 0, 0, 0);
        }
      }
      // ---- online softmax over the stage's keys (per sub-tile 16 here, 16 in the partner half-wave)
      float tmax = -INFINITY;
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        if (sub >= nsub) break;
        const int kv_left = kv_stage - sub * 32;
        if (kv_left < 32) {                                 // only a sequence's last sub-tile has keys to mask
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if ((e & 3) + 8 * (e >> 2) + 4 * h >= kv_left) sacc[sub][e] = -INFINITY;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, sacc[sub][e]);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float mn = fmaxf(m, tmax);
      if (__any(mn > m)) {                                  // some lane's running maximum moved: rescale
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        l *= alpha;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
        m = mn;
      }
      float psum = 0.f;
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        if (sub >= nsub) break;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sacc[sub][e] = __builtin_amdgcn_exp2f(sacc[sub][e] - m); psum += sacc[sub][e]; }
      }
      l += psum;
      // ---- O^T += V^T P^T
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        if (sub >= nsub) break;
        const unsigned char* const vb = sV + buf * V_BYTES + sub * 8192;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          attn_f16x8 ph, pl;
          if (PV1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ph[j] = (_Float16)sacc[sub][8 * u + j];
          } else {
            unsigned hh[4], ll[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split_pair_f16(sacc[sub][8 * u + 2 * j], sacc[sub][8 * u + 2 * j + 1], hh[j], ll[j]);
            const attn_u32x4 phu = {hh[0], hh[1], hh[2], hh[3]}, plu = {ll[0], ll[1], ll[2], ll[3]};
            ph = __builtin_bit_cast(attn_f16x8, phu);
            pl = __builtin_bit_cast(attn_f16x8, plu);
          }
#pragma unroll
          for (int half = 0; half < 2; ++half) {            // channels 0-31 -> o0, 32-63 -> o1
            const int ch = 32 * half + r, swz = (ch >> 1) & 7, c = 2 * u + h;
            const attn_f16x8 vh = *reinterpret_cast<const attn_f16x8*>(vb + ch * 128 + ((c ^ swz) * 16));
            f32x16& o = half ? o1 : o0;
            if (!PV1) {
              const attn_f16x8 vl = *reinterpret_cast<const attn_f16x8*>(vb + ch * 128 + (((4 + c) ^ swz) * 16));
              o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o, 0, 0, 0);
              o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o, 0, 0, 0);
            }
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o, 0, 0, 0);
          }
        }
      }
    }
    if (t + 1 < nstages) write_tile(buf ^ 1);
    __syncthreads();
  }
  if (wave_active) {
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    if (q_ok) {
      float* po = out + qpix * ldo;
#pragma unroll
      for (int g = 0; g < 4; ++g) {                          // rows 8g+4h .. +3 of each 32-channel tile
        f32x4 a0, a1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a0[e] = o0[4 * g + e] * inv; a1[e] = o1[4 * g + e] * inv; }
        *reinterpret_cast<f32x4*>(po + 8 * g + 4 * h) = a0;
        *reinterpret_cast<f32x4*>(po + 32 + 8 * g + 4 * h) = a1;
      }
    }
  }
}

template <int MODE, int NW, bool PV1 = false>
int seq_attn_launch(const float* q, int ldq, const float* v, int ldv, float* out, int ldo, int B, int H, int W, hipStream_t st) {
  static CdfoAttrOnce once;
  constexpr int SUB = NW / 4, LDSB = 2 * (32 * SUB * 272 + SUB * 64 * 128);
  const hipError_t e = cdfo_set_max_lds(once, reinterpret_cast<const void*>(&seq_attn_mfma_kernel<MODE, NW, PV1>), LDSB);
  if (e != hipSuccess) return (int)e;
  const int L = MODE == 0 ? W : (MODE == 1 ? H : 64);
  const int ntq = cdiv(L, 32), nb = cdiv(ntq, NW), tpb = cdiv(ntq, nb);
  const long long nseq = MODE == 0 ? (long long)B * H : (MODE == 1 ? (long long)B * W : (long long)B * (H / 8) * (W / 8));
  if (nseq * nb >= (1ll << 31)) return CDFO_EINVAL;
  static const int xcd_map = []() { const char* e = getenv("CDFO_ATTN_XCD"); return e ? atoi(e) : 1; }();   // developer A/B switch
  hipLaunchKernelGGL((seq_attn_mfma_kernel<MODE, NW, PV1>), dim3((unsigned)(nseq * nb)), dim3(NW * 64), LDSB, st, q, ldq, v, ldv,
                     out, ldo, B, H, W, nb, tpb, (nb > 1 && xcd_map) ? 1 : 0);
  return 0;
}

}  // namespace

extern "C" int cdfo_rdab_prep(const float* xq, int ldx, const float* vmax, const float* noise, const float* wW,
                              const float* bW, int B, long long P, float* sq, int lds_, float* vrow, int ldv, float* qwin,
                              int ldw, void* stream) {
  if (B <= 0 || P <= 0 || P % 64 || ldx % 4 || lds_ % 4 || ldv % 4 || ldw % 4) return CDFO_EINVAL;
  if (!aligned16(xq) || !aligned16(sq) || !aligned16(vrow) || !aligned16(qwin)) return CDFO_EALIGN;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_RDAB_PREP, 0, 4.0*(128+64+192)*(double)B*P);
  hipLaunchKernelGGL(rdab_prep_kernel<false>, dim3((unsigned)(B * P / 64)), dim3(256), 0, static_cast<hipStream_t>(stream), xq,
                     ldx, vmax, noise, wW, bW, P, sq, lds_, vrow, ldv, qwin, ldw, 0ull, 0u, nullptr, nullptr);
  CDFO_LAUNCH_CHECK();
  return 0;
}

// The same with the uniform noise drawn inside the kernel (Philox4x32-10 keyed by `seed`; `draw` numbers the call, so the
// draws of one forward are independent).  noise_out: optional [B][64][P] copy of the drawn values.
extern "C" int cdfo_rdab_prep_rng(const float* xq, int ldx, const float* vmax, long long seed, int draw, float* noise_out,
                                  const float* wW, const float* bW, int B, long long P, float* sq, int lds_, float* vrow,
                                  int ldv, float* qwin, int ldw, void* stream) {
  if (B <= 0 || P <= 0 || P % 64 || P >= (1ll << 32) || ldx % 4 || lds_ % 4 || ldv % 4 || ldw % 4) return CDFO_EINVAL;
  if (!aligned16(xq) || !aligned16(sq) || !aligned16(vrow) || !aligned16(qwin)) return CDFO_EALIGN;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_RDAB_PREP, 0, 4.0*(128+192)*(double)B*P);
  hipLaunchKernelGGL(rdab_prep_kernel<true>, dim3((unsigned)(B * P / 64)), dim3(256), 0, static_cast<hipStream_t>(stream), xq,
                     ldx, vmax, nullptr, wW, bW, P, sq, lds_, vrow, ldv, qwin, ldw, (unsigned long long)seed, (unsigned)draw,
                     noise_out, nullptr);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_rdab_prep_rng_dev(const float* xq, int ldx, const float* vmax, const void* seed_dev, int draw, float* noise_out,
                                      const float* wW, const float* bW, int B, long long P, float* sq, int lds_, float* vrow,
                                      int ldv, float* qwin, int ldw, void* stream) {
  if (B <= 0 || P <= 0 || P % 64 || P >= (1ll << 32) || ldx % 4 || lds_ % 4 || ldv % 4 || ldw % 4 || !seed_dev) return CDFO_EINVAL;
  if (!aligned16(xq) || !aligned16(sq) || !aligned16(vrow) || !aligned16(qwin) || (reinterpret_cast<uintptr_t>(seed_dev) & 7u)) return CDFO_EALIGN;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_RDAB_PREP, 0, 4.0*(128+192)*(double)B*P);
  hipLaunchKernelGGL(rdab_prep_kernel<true>, dim3((unsigned)(B * P / 64)), dim3(256), 0, static_cast<hipStream_t>(stream), xq,
                     ldx, vmax, nullptr, wW, bW, P, sq, lds_, vrow, ldv, qwin, ldw, 0ull, (unsigned)draw, noise_out,
                     static_cast<const unsigned long long*>(seed_dev));
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_colconv9(const float* in, int ldi, const float* wH, const float* bH, int B, int H, int W, float* out,
                             int ldo, void* stream) {
  if (B <= 0 || ldi % 4 || ldo % 4) return CDFO_EINVAL;
  if (!aligned16(in) || !aligned16(out)) return CDFO_EALIGN;
  const int nseg = cdiv(H, CC9_SEG);
  long long blocks = ((long long)B * nseg * W * 16 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_COLCONV9, 0, 4.0*128*(double)B*H*W);
  hipLaunchKernelGGL(colconv9_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), in, ldi, wH,
                     bH, B, H, W, nseg, out, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_seq_attn(const float* q, int ldq, const float* v, int ldv, float* out, int ldo, int B, int H, int W,
                             int mode, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || ldq % 4 || ldv % 4 || ldo % 4) return CDFO_EINVAL;
  if (!aligned16(q) || !aligned16(v) || !aligned16(out)) return CDFO_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CdfoProfScope prof(static_cast<hipStream_t>(stream), (mode%10)==0?KID_ATTN_ROW:((mode%10)==1?KID_ATTN_COL:KID_ATTN_WIN), 4.0*64*(double)B*H*W*((mode%10)==0?W:((mode%10)==1?H:64)), 4.0*192*(double)B*H*W);
  const bool pv1 = mode >= 20 && mode <= 22;           // modes 20 / 21 / 22: modes 0 / 1 / 2 with the single-fp16 second product (PV1)
  if (pv1) mode -= 20;
  if (mode == 0 || mode == 1) {
    // 8-wave workgroups amortise the staging over twice the queries, but only if the sequence's 32-query tiles fill them:
    // 272 keys = 9 tiles = 2 x 8 waves at 56 % or 3 x 4 waves at 75 % (column attention at 24 x 272 x 480: 1.61 vs 1.35 ms;
    // row attention, 480 keys: 1.67 ms with 8 waves, 1.86 with 4).  The strided key rows of the column form are NOT what
    // it waits for: the same products over transposed tensors (contiguous rows of 272 keys) took 1.50 ms.
    const int L = mode == 0 ? W : H;
    const int ntq = cdiv(L, 32);
    static const int force_nw = getenv("CDFO_ATTN_NW") ? atoi(getenv("CDFO_ATTN_NW")) : 0;     // developer switch (4 / 8)
    const bool wide = force_nw ? force_nw == 8
                               : L > 128 && (double)ntq / (cdiv(ntq, 8) * 8) >= (double)ntq / (cdiv(ntq, 4) * 4) - 0.1;
    int rc;
    if (pv1) {
      if (mode == 0) rc = !wide ? seq_attn_launch<0, 4, true>(q, ldq, v, ldv, out, ldo, B, H, W, st)
                                : seq_attn_launch<0, 8, true>(q, ldq, v, ldv, out, ldo, B, H, W, st);
      else rc = !wide ? seq_attn_launch<1, 4, true>(q, ldq, v, ldv, out, ldo, B, H, W, st)
                      : seq_attn_launch<1, 8, true>(q, ldq, v, ldv, out, ldo, B, H, W, st);
    } else if (mode == 0) rc = !wide ? seq_attn_launch<0, 4>(q, ldq, v, ldv, out, ldo, B, H, W, st)
                                     : seq_attn_launch<0, 8>(q, ldq, v, ldv, out, ldo, B, H, W, st);
    else rc = !wide ? seq_attn_launch<1, 4>(q, ldq, v, ldv, out, ldo, B, H, W, st)
                    : seq_attn_launch<1, 8>(q, ldq, v, ldv, out, ldo, B, H, W, st);
    if (rc) return rc;
  } else if (mode == 10) {   // VALU reference forms of modes 0 / 1 (kept for A/B tests)
    hipLaunchKernelGGL(seq_attn_kernel<0>, dim3((unsigned)((long long)B * H * cdiv(W, 64))), dim3(64), 0, st, q, ldq, v,
                       ldv, out, ldo, B, H, W);
  } else if (mode == 11) {
    hipLaunchKernelGGL(seq_attn_kernel<1>, dim3((unsigned)((long long)B * W * cdiv(H, 64))), dim3(64), 0, st, q, ldq, v,
                       ldv, out, ldo, B, H, W);
  } else if (mode == 2) {
    if ((H & 7) || (W & 7)) return CDFO_EINVAL;
    const int rc = pv1 ? seq_attn_launch<2, 4, true>(q, ldq, v, ldv, out, ldo, B, H, W, st)
                       : seq_attn_launch<2, 4>(q, ldq, v, ldv, out, ldo, B, H, W, st);
    if (rc) return rc;
  } else if (mode == 12) {   // VALU reference form of mode 2
    if ((H & 7) || (W & 7)) return CDFO_EINVAL;
    hipLaunchKernelGGL(seq_attn_kernel<2>, dim3((unsigned)((long long)B * (H / 8) * (W / 8))), dim3(64), 0, st, q, ldq, v,
                       ldv, out, ldo, B, H, W);
  } else {
    return CDFO_EINVAL;
  }
  CDFO_LAUNCH_CHECK();
  return 0;
}
