// Window-sampled fused deformable convolution forward (DCNv2 / DCNv1), the alignment module's kind of shape:
// 3x3, stride 1, dilation 1, groups == 1, (C / deformable_groups) % 4 == 0, Co in {32, 64}  (arch.py:3265-3352: C = Co = 64, dg = 16).
// Replaces modulated_deformable_im2col_gpu_kernel + dmcn_im2col_bilinear + addmm_ (ops/dcn/src/deform_conv_cuda_kernel.cu:466-496,
// 569-632; deform_conv_cuda.cpp:534-563) without the `columns` buffer.  See dcn.hip for the operator contract and the general
// kernel, dcn_fast.hip for the round-1/2 form this one supersedes at these shapes.
//
// Why: dcn_fast gathers every bilinear corner from global memory (16 bytes per lane and corner out of a group-planar copy of
// the input made by a prepass): 63 % texture-address-unit busy, 157 vector instructions per (pixel, 4 channels, tap) item,
// 0.21 of the HBM roofline, 1.36x the algorithmic traffic.  Here:
//   * one 512-thread workgroup = an 8 x 32 tile of output pixels x all output channels; wave w = tile row w;
//   * per pair of 4-channel blocks (a "chunk": 2 x 9 taps x 4 channels = 72 K rows, padded to 5 MFMA K steps of 16) the input
//     WINDOW that the tile's samples can reach with offsets up to ~10 pixels -- 31 rows x 56 columns x 4 channels fp32 = 27 KB
//     per block -- is copied NCHW -> LDS [row][column][4 channels] with 16-byte loads along the image row (no group-planar
//     prepass, no second copy of the input in HBM), zero-filled outside the image (= the operator's zero padding), double
//     buffered: the next chunk's window and packed weights are fetched into registers while this chunk is sampled;
//   * a lane samples for ITS pixel exactly the 8 K rows its v_mfma_f32_32x32x16_f16 B operand holds (2 items x 4 channels;
//     the two half-waves take different taps of the same 32 pixels): 4 ds_read_b128 per item from the window, the bilinear
//     combine, mask, fp16 hi + lo split -- and the values ARE the B fragments: no LDS round trip of the sampled columns, no
//     barrier between sampling and contraction; hi*hi + lo*hi + hi*lo against fp16 hi | lo weights (fp32-grade);
//   * a sample whose corners fall outside the window (offsets beyond ~10 pixels) is gathered from global memory by that lane
//     alone (rare; correct for any offset, NaN / infinite offsets sample nothing like the reference's range test, cu:617);
//   * range safety without a pass over the input: every chunk's window maximum (found while staging) sets a power-of-two
//     scale for the fp16 split; the accumulators carry a running unit that only grows (exact rescale by a power of two);
//     a converted half at the fp16 limit (mask far outside [0, 1], non-finite data) raises the re-run flag and
//     cdfo_dcn_forward re-runs the exact-fp32 kernel, as for dcn_fast.
// HBM roofline: (C + 3*dg*9 + Co) * 4 bytes per output pixel; offsets + mask are 77 % of it and are read
// exactly once, coalesced (each half-wave reads 128-byte rows of one tap plane).
#include "common.h"

typedef _Float16 wn_h8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int WN_TH = 8, WN_TW = 32;                 // output tile
constexpr int WN_WH = 31, WN_WW = 56;                // window: rows oy0 - ph - 10 .. + 30, columns floor4(ox0 - pw - 10) .. + 55
constexpr int WN_RY = 10, WN_RX = 10;
constexpr int WN_WIN = WN_WH * WN_WW * 16;           // 27,776 bytes per 4-channel block
constexpr int WN_T = 9, WN_STEPS = 5;                // 18 items (2 blocks x 9 taps) + 2 zero items = 5 K steps of 4 items
constexpr int WN_THREADS = 512;
constexpr int WN_QPR = WN_WW / 4;                    // 14 four-pixel quads per window row
constexpr int WN_TASKS = WN_WH * WN_QPR;             // 434 staging tasks per block
constexpr int WN_NTASK = 2;                          // tasks per thread and chunk: 2 * 512 >= 2 * 434

struct WinArgs {
  const float* in; const float* offset; const float* mask; const float* bias; float* out;
  const wn_h8* wp;        // packed weights: [chunk][step][mj][hi | lo][64 lanes] x 8 fp16 (dcn_win_pack_kernel)
  unsigned* flags;        // [0] = bits of max |w| (-> weight scale), [2] = re-run request
  int B, C, H, W, Co, Ho, Wo, ph, pw, dg;
  int nchunks;            // ceil(C / 8)
  int tiles_x, ntiles;
};

__device__ __forceinline__ float wn_pow2_scale(unsigned max_bits) {      // as dcn_fast.hip: m * s in (8, 16]
  const float m = __uint_as_float(max_bits);
  return m > 0.f ? exp2f(fminf(fmaxf(4.f - ceilf(log2f(m)), -100.f), 100.f)) : 1.f;
}

__global__ __launch_bounds__(256) void dcn_win_wmax_kernel(const float* __restrict__ w, long long n, unsigned* __restrict__ wmax) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m < 3.0e38f) atomicMax(wmax, __float_as_uint(m));
}

// one thread per fp16 element: gid = ((((chunk*5 + s)*MJ + mj)*2 + hl)*64 + lane)*8 + e.  K row 8*(lane>>5) + e of step s is
// item i = 4 s + 2 (lane>>5) + (e>>2), channel e&3: block (i >= 9), tap i % 9 -> input channel 8 chunk + 4 block + (e&3)
__global__ __launch_bounds__(256) void dcn_win_pack_kernel(const float* __restrict__ w, const unsigned* __restrict__ wmax,
                                                           _Float16* __restrict__ wp, int C, int Co, int MJ, long long n) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n) return;
  const int e = (int)(gid & 7), lane = (int)((gid >> 3) & 63), hl = (int)((gid >> 9) & 1);
  const long long r = gid >> 10;
  const int mj = (int)(r % MJ), s = (int)((r / MJ) % WN_STEPS), chunk = (int)(r / ((long long)MJ * WN_STEPS));
  const int i = 4 * s + 2 * (lane >> 5) + (e >> 2), blk = i >= WN_T ? 1 : 0, t = i - WN_T * blk;
  const int c = 8 * chunk + 4 * blk + (e & 3), m = mj * 32 + (lane & 31);
  float v = 0.f;
  if (i < 2 * WN_T && c < C && m < Co) v = w[((long long)m * C + c) * WN_T + t] * wn_pow2_scale(wmax[0]);
  const _Float16 h = (_Float16)v;
  wp[gid] = hl ? (_Float16)(v - (float)h) : h;
}

template <int MJ, bool AL4>     // MJ: 32-channel output tiles (Co = 32 MJ); AL4: W % 4 == 0 and a 16-byte aligned input
__global__ __launch_bounds__(WN_THREADS) void dcn_win_kernel(WinArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WTS = WN_STEPS * MJ * 2 * 1024;                 // packed weights of one chunk
  constexpr int WSLOTS = WTS / 16, WLD = (WSLOTS + WN_THREADS - 1) / WN_THREADS;
  unsigned char* const sWin = smem;                              // [2 buffers][2 blocks][WN_WIN]
  unsigned char* const sWt = smem + 4 * WN_WIN;                  // [2 buffers][WTS]
  float* const sMax = reinterpret_cast<float*>(smem + 4 * WN_WIN + 2 * WTS);      // [2 buffers][8 waves]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, n = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroups go round-robin to the 8 XCDs: each XCD walks its own contiguous band of tiles (its L2 then sees a band of image
  // rows, and neighbouring tiles' overlapping windows hit it)
  const int band = gridDim.x >> 3;
  const int tile = (blockIdx.x & 7) * band + (blockIdx.x >> 3);
  if (tile >= a.ntiles) return;
  const int b = blockIdx.y;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int oy0 = ty * WN_TH, ox0 = tx * WN_TW;
  const int wy0 = oy0 - a.ph - WN_RY, wx0 = (ox0 - a.pw - WN_RX) & ~3;
  const int H = a.H, W = a.W, P = a.Ho * a.Wo, HW = H * W;
  const int oy = oy0 + wave, ox = ox0 + n;
  const bool pvalid = oy < a.Ho && ox < a.Wo;
  const int p = pvalid ? oy * a.Wo + ox : 0;
  const float hb = (float)(oy - a.ph), wb = (float)(ox - a.pw);
  const float* const in_b = a.in + (long long)b * a.C * HW;
  const float* const off_b = a.offset + (long long)b * a.dg * 2 * WN_T * P + p;
  const float* const msk_b = a.mask ? a.mask + (long long)b * a.dg * WN_T * P + p : nullptr;
  const int cdg4 = (a.C / a.dg) >> 2;                            // 4-channel blocks per deformable group

  // ---- this lane's ten items of a chunk: k = 2 s + j -> item i = 4 s + 2 half + j
  int it_t[2 * WN_STEPS], it_blk[2 * WN_STEPS];
  float it_h[2 * WN_STEPS], it_w[2 * WN_STEPS];
#pragma unroll
  for (int k = 0; k < 2 * WN_STEPS; ++k) {
    const int i = 4 * (k >> 1) + 2 * half + (k & 1);
    it_blk[k] = i >= WN_T ? 1 : 0;
    const int t = i - WN_T * it_blk[k];
    it_t[k] = i < 2 * WN_T ? t : -1;
    const int ki = t / 3;
    it_h[k] = (float)ki;
    it_w[k] = (float)(t - 3 * ki);
  }

  // ---- staging tasks of this thread: (block, window row, quad)
  int tk_off[WN_NTASK], tk_lds[WN_NTASK], tk_blk[WN_NTASK];
  bool tk_on[WN_NTASK], tk_in[WN_NTASK];
  bool tk_el[WN_NTASK][4];
#pragma unroll
  for (int q = 0; q < WN_NTASK; ++q) {
    const int task = tid + q * WN_THREADS;
    tk_on[q] = task < 2 * WN_TASKS;
    const int blk = task >= WN_TASKS ? 1 : 0, rem = task - blk * WN_TASKS;
    const int r = rem / WN_QPR, qx = rem - r * WN_QPR;
    const int gy = wy0 + r, gx = wx0 + 4 * qx;
    tk_blk[q] = blk;
    tk_lds[q] = blk * WN_WIN + (r * WN_WW + 4 * qx) * 16;
    const bool row_in = gy >= 0 && gy < H;
    tk_in[q] = tk_on[q] && row_in && gx >= 0 && gx + 3 < W;
#pragma unroll
    for (int e = 0; e < 4; ++e) tk_el[q][e] = tk_on[q] && row_in && gx + e >= 0 && gx + e < W;
    tk_off[q] = (row_in ? gy : 0) * W + gx;
  }
  f32x4 wr[WN_NTASK][4];       // prefetched window values: [task][channel] = 4 consecutive pixels
  wn_h8 wtr[WLD];              // prefetched packed weights
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int q = 0; q < WN_NTASK; ++q) {
      const int c0 = 8 * chunk + 4 * tk_blk[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c0 + e < a.C) {
          const float* src = in_b + (long long)(c0 + e) * HW + tk_off[q];
          if (AL4) {
            if (tk_in[q]) v = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int x = 0; x < 4; ++x)
              if (tk_el[q][x]) v[x] = src[x];
          }
        }
        wr[q][e] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int slot = tid + i * WN_THREADS;
      if (slot < WSLOTS) wtr[i] = a.wp[(long long)chunk * WSLOTS + slot];
    }
  };
  auto commit = [&](int buf) {      // registers -> LDS buffer `buf` (+ this wave's window maximum)
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < WN_NTASK; ++q) {
      if (tk_on[q]) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const f32x4 v = {wr[q][0][x], wr[q][1][x], wr[q][2][x], wr[q][3][x]};
          *reinterpret_cast<f32x4*>(sWin + buf * 2 * WN_WIN + tk_lds[q] + x * 16) = v;
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float av = fabsf(v[e]); m = av < 3.0e38f ? fmaxf(m, av) : m; }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int slot = tid + i * WN_THREADS;
      if (slot < WSLOTS) *reinterpret_cast<wn_h8*>(sWt + buf * WTS + slot * 16) = wtr[i];
    }
    m = wave_max(m);
    if (lane == 0) sMax[buf * 8 + wave] = m;
  };

  // ---- offsets / mask of a chunk's ten items (prefetched one chunk ahead)
  float noh[2 * WN_STEPS], now_[2 * WN_STEPS], nmk[2 * WN_STEPS];
  bool nlive[2 * WN_STEPS];
  auto load_offsets = [&](int chunk) {
    const int d0 = (2 * chunk) / cdg4, d1 = (2 * chunk + 1) / cdg4;
#pragma unroll
    for (int k = 0; k < 2 * WN_STEPS; ++k) {
      const bool live = pvalid && it_t[k] >= 0 && 8 * chunk + 4 * it_blk[k] < a.C;
      nlive[k] = live;
      noh[k] = now_[k] = 0.f; nmk[k] = 1.f;
      if (live) {
        const int ot = (it_blk[k] ? d1 : d0) * WN_T + it_t[k];          // per-image indices fit 32 bits (checked by the launcher)
        noh[k] = off_b[(long long)(ot * 2) * P];
        now_[k] = off_b[(long long)(ot * 2 + 1) * P];
        if (msk_b) nmk[k] = msk_b[(long long)ot * P];
      }
    }
  };

  f32x16 acc[MJ];
#pragma unroll
  for (int j = 0; j < MJ; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  int e_run = -100000;          // binary exponent of the accumulators' unit: sampled values are scaled by 2^(3 - e_run)
  unsigned ovfbits = 0;

  fetch(0);
  load_offsets(0);
  commit(0);
  const int nch = a.nchunks;
  for (int chunk = 0; chunk < nch; ++chunk) {
    const int buf = chunk & 1;
    __syncthreads();            // buffer `buf` is complete; nobody still reads the other one
    // ---- this chunk's power-of-two scale from its window maximum
    float M = sMax[buf * 8];
#pragma unroll
    for (int i = 1; i < 8; ++i) M = fmaxf(M, sMax[buf * 8 + i]);
    int e_c = -100;
    if (M > 0.f) { frexpf(M, &e_c); e_c = e_c < -100 ? -100 : (e_c > 100 ? 100 : e_c); }       // M < 2^e_c
    if (e_c > e_run) {
      const int de = e_run - e_c;
      const float f = ldexpf(1.f, de < -200 ? -200 : de);          // <= 1 (0 for the first chunk: the sums are zero)
#pragma unroll
      for (int j = 0; j < MJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] *= f;
      e_run = e_c;
    }
    const float s_in = ldexpf(1.f, 3 - e_run);
    // ---- current offsets; next chunk's window / weights / offsets go in flight behind them
    float oh[2 * WN_STEPS], ow[2 * WN_STEPS], mk[2 * WN_STEPS];
    bool live[2 * WN_STEPS];
#pragma unroll
    for (int k = 0; k < 2 * WN_STEPS; ++k) { oh[k] = noh[k]; ow[k] = now_[k]; mk[k] = nmk[k]; live[k] = nlive[k]; }
    if (chunk + 1 < nch) {
      fetch(chunk + 1);
      load_offsets(chunk + 1);
    }
    const unsigned char* const win = sWin + buf * 2 * WN_WIN;
    const unsigned char* const wt = sWt + buf * WTS + lane * 16;
#pragma unroll
    for (int s = 0; s < WN_STEPS; ++s) {
      float val[8];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = 2 * s + j;
        const float h_im = hb + it_h[k] + oh[k], w_im = wb + it_w[k] + ow[k];
        const bool valid = live[k] && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;      // cu:617
        const float hs = valid ? h_im : 0.f, ws = valid ? w_im : 0.f;
        const float fh = floorf(hs), fw = floorf(ws);
        const int hl = (int)fh, wl = (int)fw;
        const float lh = hs - fh, lw = ws - fw, hh = 1.f - lh, hw = 1.f - lw;
        const int ly = hl - wy0, lx = wl - wx0;
        const bool inwin = ly >= 0 && ly < WN_WH - 1 && lx >= 0 && lx < WN_WW - 1;
        f32x4 v1, v2, v3, v4;
        if (__builtin_expect(valid && !inwin, 0)) {
          // beyond the window: this lane gathers its four corners from global memory (zero outside the image, cu:481-488)
          const int c0 = 8 * chunk + 4 * it_blk[k];
          const bool r0 = hl >= 0, r1 = hl + 1 <= H - 1, q0 = wl >= 0, q1 = wl + 1 <= W - 1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float* pl = in_b + (long long)(c0 + e) * HW;
            v1[e] = (r0 && q0) ? pl[hl * W + wl] : 0.f;
            v2[e] = (r0 && q1) ? pl[hl * W + wl + 1] : 0.f;
            v3[e] = (r1 && q0) ? pl[(hl + 1) * W + wl] : 0.f;
            v4[e] = (r1 && q1) ? pl[(hl + 1) * W + wl + 1] : 0.f;
          }
        } else {
          const int o = (valid ? (ly * WN_WW + lx) * 16 : 0) + it_blk[k] * WN_WIN;
          v1 = *reinterpret_cast<const f32x4*>(win + o);
          v2 = *reinterpret_cast<const f32x4*>(win + o + 16);
          v3 = *reinterpret_cast<const f32x4*>(win + o + WN_WW * 16);
          v4 = *reinterpret_cast<const f32x4*>(win + o + WN_WW * 16 + 16);
        }
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const float ms = mk[k] * s_in;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sv = (w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e]) * ms;
          val[4 * j + e] = valid ? sv : 0.f;
        }
      }
      // fp16 hi + lo with the packed round-toward-zero conversion (hi truncated, lo = the exact remainder truncated)
      typedef __fp16 hp2 __attribute__((ext_vector_type(2)));
      union { hp2 h[4]; wn_h8 v8; } uh, ul;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uh.h[q] = __builtin_amdgcn_cvt_pkrtz(val[2 * q], val[2 * q + 1]);
        ul.h[q] = __builtin_amdgcn_cvt_pkrtz(val[2 * q] - (float)uh.h[q][0], val[2 * q + 1] - (float)uh.h[q][1]);
        union { hp2 h; unsigned u; } cv;       // range check on the CONVERTED halves (see dcn_fast.hip): >= 0x7bff, inf, NaN
        cv.h = uh.h[q];
        ovfbits |= ((cv.u & 0x7fff7fffu) + 0x04010401u) & 0x80008000u;
      }
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        const wn_h8 Ah = *reinterpret_cast<const wn_h8*>(wt + ((s * MJ + j) * 2 + 0) * 1024);
        const wn_h8 Al = *reinterpret_cast<const wn_h8*>(wt + ((s * MJ + j) * 2 + 1) * 1024);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, uh.v8, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, uh.v8, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, ul.v8, acc[j], 0, 0, 0);
      }
    }
    if (chunk + 1 < nch) commit(buf ^ 1);
  }
  if (ovfbits) atomicOr(a.flags + 2, 1u);      // out of the fp16 hi + lo range somewhere: the exact kernel re-runs (dcn.hip)
  // ---- store D[row = cout][col = pixel] (+ bias), NCHW: 32 lanes = 128 contiguous bytes of one output row
  if (pvalid) {
    const float inv = ldexpf(1.f / wn_pow2_scale(a.flags[0]), e_run > -1000 ? e_run - 3 : 0);
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int oc = j * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (oc < a.Co) a.out[((long long)b * a.Co + oc) * P + p] = acc[j][e] * inv + (a.bias ? a.bias[oc] : 0.f);
      }
  }
}

template <int MJ, bool AL4>
hipError_t wn_launch(const WinArgs& a, dim3 grid, hipStream_t st) {
  constexpr int LDSB = 4 * WN_WIN + 2 * (WN_STEPS * MJ * 2 * 1024) + 64;
  static CdfoAttrOnce once;
  const hipError_t e = cdfo_set_max_lds(once, reinterpret_cast<const void*>(&dcn_win_kernel<MJ, AL4>), LDSB);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((dcn_win_kernel<MJ, AL4>), grid, dim3(WN_THREADS), LDSB, st, a);
  return hipGetLastError();
}

}  // namespace

// Workspace bytes of this path (0 = it does not apply to these shapes).
long long cdfo_dcn_win_workspace_bytes(int C, int Co, int kh, int kw, int sh, int sw, int dh, int dw, int groups, int dg) {
  if (kh != 3 || kw != 3 || sh != 1 || sw != 1 || dh != 1 || dw != 1 || groups != 1 || dg <= 0 || C % dg || (C / dg) % 4) return 0;
  if (Co != 32 && Co != 64) return 0;
  const long long nch = (C + 7) / 8;
  return nch * WN_STEPS * (Co / 32) * 2 * 1024 + 256;
}

// Called by cdfo_dcn_forward (dcn.hip) after its argument checks.  Returns 1 when it launched, 0 when this path does not
// apply (the caller tries the next one), 2 + hipError_t when a launch failed.
int cdfo_dcn_forward_win(const float* in, const float* offset, const float* mask, const float* weight, const float* bias,
                         float* out, int B, int C, int H, int W, int Co, int Ho, int Wo, int kh, int kw, int sh, int sw, int ph,
                         int pw, int dh, int dw, int groups, int dg, void* workspace, long long workspace_bytes, hipStream_t st,
                         const unsigned** rerun_flag) {
  const long long need = cdfo_dcn_win_workspace_bytes(C, Co, kh, kw, sh, sw, dh, dw, groups, dg);
  if (!need || !workspace || workspace_bytes < need || !aligned16(workspace)) return 0;
  if ((long long)C * H * W >= (1ll << 30) || (long long)dg * 2 * WN_T * Ho * Wo >= (1ll << 30)) return 0;      // 32-bit per-image indices
  if (ph < 0 || pw < 0 || ph > 8 || pw > 8) return 0;                                                          // window margins assume a small pad
  const int MJ = Co / 32, nch = (C + 7) / 8;
  char* ws = static_cast<char*>(workspace);
  const long long wpb = need - 256;
  _Float16* wp = reinterpret_cast<_Float16*>(ws);
  unsigned* flags = reinterpret_cast<unsigned*>(ws + wpb);
  if (hipMemsetAsync(flags, 0, 16, st) != hipSuccess) return 2 + (int)hipGetLastError();
  const long long nwt = (long long)Co * C * WN_T;
  hipLaunchKernelGGL(dcn_win_wmax_kernel, dim3((unsigned)((nwt + 2047) / 2048 < 256 ? (nwt + 2047) / 2048 : 256)), dim3(256), 0, st,
                     weight, nwt, flags);
  const long long nw = wpb / 2;
  hipLaunchKernelGGL(dcn_win_pack_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, weight, flags, wp, C, Co, MJ, nw);
  *rerun_flag = flags + 2;
  WinArgs a{in, offset, mask, bias, out, reinterpret_cast<const wn_h8*>(wp), flags, B, C, H, W, Co, Ho, Wo, ph, pw, dg, nch,
            cdiv(Wo, WN_TW), cdiv(Wo, WN_TW) * cdiv(Ho, WN_TH)};
  dim3 grid(8 * cdiv(a.ntiles, 8), B);
  const bool al4 = W % 4 == 0 && aligned16(in);
  hipError_t e;
  if (MJ == 1) e = al4 ? wn_launch<1, true>(a, grid, st) : wn_launch<1, false>(a, grid, st);
  else e = al4 ? wn_launch<2, true>(a, grid, st) : wn_launch<2, false>(a, grid, st);
  return e == hipSuccess ? 1 : 2 + (int)e;
}
