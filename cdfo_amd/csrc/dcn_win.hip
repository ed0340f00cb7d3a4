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

// Geometry.  WN_NW waves per workgroup = tile rows: 8 (two per SIMD, <= 256 VGPRs).  The kernel is latency-, not throughput-
// bound (VALU 26 %, MFMA 15 %, LDS ~35 % busy, 37 % of the wave-cycles parked on waits: profiles/r03_pmc_counters_dcn_fwd.txt),
// so more waves are what it wants -- but the 16-wave form (a 16 x 32 tile fits the LDS with a +-8 pixel window) needs the
// per-wave state in 128 VGPRs and hipcc spills 130 of them (a spill is ruinous here, see the kernel); -DWN_NW=16 builds it.
#ifndef WN_NW
#define WN_NW 8
#endif
constexpr int WN_TH = WN_NW, WN_TW = 32;             // output tile
// window: rows oy0 - ph - RY .. + WH - 1, columns floor4(ox0 - pw - RX) .. + WW - 1  (reach: ~10 pixels at 8 waves, ~8 at 16)
constexpr int WN_WH = WN_NW == 8 ? 31 : 35, WN_WW = WN_NW == 8 ? 56 : 52;
constexpr int WN_RY = WN_NW == 8 ? 10 : 8, WN_RX = WN_NW == 8 ? 10 : 7;
constexpr int WN_WIN = WN_WH * WN_WW * 16;           // bytes per 4-channel block (27,776 / 29,120)
constexpr int WN_T = 9, WN_STEPS = 5;                // 2 blocks x (9 taps + 1 zero pad) = 5 K steps of 4 items
constexpr int WN_THREADS = 64 * WN_NW;
constexpr int WN_QPR = WN_WW / 4;                    // four-pixel quads per window row
constexpr int WN_TASKS = WN_WH * WN_QPR;             // staging tasks per block
constexpr int WN_NTASK = (2 * WN_TASKS + WN_THREADS - 1) / WN_THREADS;     // tasks per thread and chunk (2 at 8 waves, 1 at 16)
static_assert(WN_NTASK == 1 || WN_NTASK == 2, "staging schedule");

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
// block (lane>>5) of the chunk, tap 2 s + (e>>2) (tap 9 = the zero pad), channel e&3 -> input channel 8 chunk + 4 block + (e&3)
__global__ __launch_bounds__(256) void dcn_win_pack_kernel(const float* __restrict__ w, const unsigned* __restrict__ wmax,
                                                           _Float16* __restrict__ wp, int C, int Co, int MJ, long long n) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n) return;
  const int e = (int)(gid & 7), lane = (int)((gid >> 3) & 63), hl = (int)((gid >> 9) & 1);
  const long long r = gid >> 10;
  const int mj = (int)(r % MJ), s = (int)((r / MJ) % WN_STEPS), chunk = (int)(r / ((long long)MJ * WN_STEPS));
  const int blk = lane >> 5, t = 2 * s + (e >> 2);
  const int c = 8 * chunk + 4 * blk + (e & 3), m = mj * 32 + (lane & 31);
  float v = 0.f;
  if (t < WN_T && c < C && m < Co) v = w[((long long)m * C + c) * WN_T + t] * wn_pow2_scale(wmax[0]);
  const _Float16 h = (_Float16)v;
  wp[gid] = hl ? (_Float16)(v - (float)h) : h;
}

__device__ __forceinline__ float wn_bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
// (offsets and masks are read exactly once per launch, 1.8 GB beside a 267 MB input whose windows neighbouring tiles re-read from L2)
__device__ __forceinline__ float wn_bload_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 2));
}
__device__ __forceinline__ f32x4 wn_bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// MJ: 32-channel output tiles (Co = 32 MJ); AL4: W % 4 == 0 and a 16-byte aligned input; MASK: modulated (DCNv2)
// DBG (developer ablations through CDFO_DCN_DBG in -DCDFO_DEV_ABLATIONS builds, wrong results, tools/bench_dcn.py only): 1 = no MFMAs, 2 = no LDS reads of the
// samples' corners, 4 = no window staging after the first chunk, 8 = no offset / mask loads after the first chunk, 16 = no
// workgroup barriers
template <int MJ, bool AL4, bool MASK, int DBG = 0>
__global__ __launch_bounds__(WN_THREADS) void dcn_win_kernel(WinArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WTS = WN_STEPS * MJ * 2 * 1024;                 // packed weights of one chunk
  constexpr int WSLOTS = WTS / 16, WLD = (WSLOTS + WN_THREADS - 1) / WN_THREADS;
  constexpr unsigned OOR = 0x80000000u;                          // buffer offset past every descriptor: the load returns zeros
  unsigned char* const sWin = smem;                              // [2 buffers][2 blocks][WN_WIN]
  unsigned char* const sWt = smem + 4 * WN_WIN;                  // [2 buffers][WTS]
  unsigned* const sMax = reinterpret_cast<unsigned*>(smem + 4 * WN_WIN + 2 * WTS);   // [2 buffers][waves]: max |bits| of the windows
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
  const int cdg4 = (a.C / a.dg) >> 2;                            // 4-channel blocks per deformable group
  // per-image buffer descriptors (wave-uniform): 32-bit lane offsets + scalar offsets, hardware zero fill out of range
  const float* const in_b = a.in + (long long)b * a.C * HW;
  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b), 0, a.C * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_off = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.offset + (long long)b * a.dg * 2 * WN_T * P), 0, a.dg * 2 * WN_T * P * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_msk = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(MASK ? a.mask + (long long)b * a.dg * WN_T * P : a.offset), 0, a.dg * WN_T * P * 4, 0x00020000);

  // ---- the chunk's two 4-channel blocks belong to the two half-waves: lanes 0-31 sample block 0, lanes 32-63 block 1, both at
  // taps (2 s, 2 s + 1) in K step s (tap 9 of step 4 is a zero pad: skipped at compile time).  Taps are compile-time constants
  // and the block only enters through per-lane byte offsets that change once per chunk.

  // ---- staging tasks of this thread: (block, window row, quad)
  unsigned tk_vo[WN_NTASK][AL4 ? 1 : 4];      // byte offset inside a channel plane, OOR where the image ends
  int tk_lds[WN_NTASK], tk_blk[WN_NTASK];
  bool tk_on[WN_NTASK];
#pragma unroll
  for (int q = 0; q < WN_NTASK; ++q) {
    const int task = tid + q * WN_THREADS;
    tk_on[q] = task < 2 * WN_TASKS;
    const int blk = task >= WN_TASKS ? 1 : 0, rem = task - blk * WN_TASKS;
    const int r = rem / WN_QPR, qx = rem - r * WN_QPR;
    const int gy = wy0 + r, gx = wx0 + 4 * qx;
    tk_blk[q] = blk;
    tk_lds[q] = blk * WN_WIN + (r * WN_WW + 4 * qx) * 16;
    const bool row_in = tk_on[q] && gy >= 0 && gy < H;
    // (the task's block enters through the LANE offset: the scalar offset of a buffer load must be wave-uniform, and a wave's
    // tasks straddle the two blocks -- a per-lane scalar would make the compiler emit a waterfall loop)
    const unsigned blk_off = (unsigned)(blk * 4 * HW) * 4u;
    if (AL4) {
      tk_vo[q][0] = (row_in && gx >= 0 && gx + 3 < W) ? (unsigned)(gy * W + gx) * 4u + blk_off : OOR;
    } else {
#pragma unroll
      for (int x = 0; x < 4; ++x)
        tk_vo[q][AL4 ? 0 : x] = (row_in && gx + x >= 0 && gx + x < W) ? (unsigned)(gy * W + gx + x) * 4u + blk_off : OOR;
    }
  }
  // Register budget: with one 512-thread workgroup per CU a wave may use 256 VGPRs, and a single spill is ruinous here --
  // scratch reloads count in vmcnt, so waiting for one drains every prefetch in flight.  Hence the ROLLING prefetch below: a
  // window task / a tap's offsets are re-requested right after their last use, into the registers that use just freed; and
  // nothing but the write to LDS (half a chunk later) consumes a fetched register.
  wn_h8 wtr[WLD];              // prefetched packed weights of the next chunk
  auto fetch_task = [&](int chunk, int q, f32x4 (&wr)[4]) {
    const bool blk_ok = 8 * chunk + 4 * tk_blk[q] < a.C;      // per lane (C % 8 == 4: the last chunk has one block)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned so = (unsigned)((8 * chunk + e) * HW) * 4u;     // wave-uniform
      if (AL4) {
        wr[e] = wn_bload4(r_in, blk_ok ? tk_vo[q][0] : OOR, so);
      } else {
#pragma unroll
        for (int x = 0; x < 4; ++x) wr[e][x] = wn_bload(r_in, blk_ok ? tk_vo[q][AL4 ? 0 : x] : OOR, so);
      }
    }
  };
  auto commit_task = [&](int buf, int q, const f32x4 (&wr)[4], unsigned& m) {   // registers -> window buffer `buf`; m = running max |bits|
    if (tk_on[q]) {
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const f32x4 v = {wr[0][x], wr[1][x], wr[2][x], wr[3][x]};
        *reinterpret_cast<f32x4*>(sWin + buf * 2 * WN_WIN + tk_lds[q] + x * 16) = v;
        // running maximum of |bits|: as integers, so that a NaN (pattern above infinity's) is SEEN, not skipped like fmax does
#pragma unroll
        for (int e = 0; e < 4; ++e) { const unsigned ub = __float_as_uint(v[e]) & 0x7fffffffu; m = ub > m ? ub : m; }
      }
    }
  };
  auto fetch_weights = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int slot = tid + i * WN_THREADS;
      wtr[i] = a.wp[(unsigned)(chunk * WSLOTS + (slot < WSLOTS ? slot : 0))];
    }
  };
  auto commit_weights = [&](int buf, unsigned m) {
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int slot = tid + i * WN_THREADS;
      if (slot < WSLOTS) *reinterpret_cast<wn_h8*>(sWt + buf * WTS + slot * 16) = wtr[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
    if (lane == 0) sMax[buf * WN_NW + wave] = m;
  };

  // ---- offsets / mask of the lane's nine taps: lane byte offset = (its block's deformable group, its pixel), scalar = the tap
  float oh[WN_T], ow[WN_T], mk[WN_T];
  auto lane_offsets = [&](int chunk, unsigned& vo_off, unsigned& vo_msk, bool& alive) {
    int d0 = (2 * chunk) / cdg4, d1 = (2 * chunk + 1) / cdg4;
    d0 = d0 < a.dg ? d0 : a.dg - 1;
    d1 = d1 < a.dg ? d1 : a.dg - 1;
    const int d = half ? d1 : d0;
    alive = pvalid && 8 * chunk + 4 * half < a.C;
    vo_off = (unsigned)(d * 2 * WN_T * P + p) * 4u;          // dead lanes read a valid address and are ignored
    vo_msk = (unsigned)(d * WN_T * P + p) * 4u;
  };
  auto load_tap = [&](unsigned vo_off, unsigned vo_msk, int t, float& o_h, float& o_w, float& m_k) {
    o_h = wn_bload_nt(r_off, vo_off, (unsigned)(2 * t * P) * 4u);
    o_w = wn_bload_nt(r_off, vo_off, (unsigned)((2 * t + 1) * P) * 4u);
    m_k = MASK ? wn_bload_nt(r_msk, vo_msk, (unsigned)(t * P) * 4u) : 1.f;
  };

  f32x16 acc[MJ];
#pragma unroll
  for (int j = 0; j < MJ; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  int e_run = -100000;          // binary exponent of the accumulators' unit: sampled values are scaled by 2^(3 - e_run)
  unsigned ovfbits = 0;
  float vmax = 0.f;            // largest |scaled sample| of this lane's cold-path samples (not bounded by a window maximum)
  float mmax = 0.f;            // largest |mask| of this lane's window samples: |sample * mask * scale| < 8 |mask| (see s_in below)

  {   // prologue: chunk 0 into buffer 0
    unsigned m = 0u;
    f32x4 w0[4];
#pragma unroll
    for (int q = 0; q < WN_NTASK; ++q) {
      fetch_task(0, q, w0);
      commit_task(0, q, w0, m);
    }
    fetch_weights(0);
    unsigned vo, vm;
    bool al;
    lane_offsets(0, vo, vm, al);
#pragma unroll
    for (int t = 0; t < WN_T; ++t) load_tap(vo, vm, t, oh[t], ow[t], mk[t]);
    commit_weights(0, m);
  }
  const int nch = a.nchunks;
  const float fH = (float)H, fW = (float)W;
  const float fwy0 = (float)wy0, fwy1 = (float)(wy0 + WN_WH - 2), fwx0 = (float)wx0, fwx1 = (float)(wx0 + WN_WW - 2);
  const float hbt[3] = {hb, hb + 1.f, hb + 2.f}, wbt[3] = {wb, wb + 1.f, wb + 2.f};       // base position per tap row / column
  const unsigned char* const win_lane = sWin + half * WN_WIN;
  for (int chunk = 0; chunk < nch; ++chunk) {
    const int buf = chunk & 1;
    const bool more = chunk + 1 < nch;
    unsigned vo_c, vm_c, vo_n, vm_n;      // lane offsets of this chunk (cold path) and of the next one (rolling prefetch)
    bool alive, alive_n;
    lane_offsets(chunk, vo_c, vm_c, alive);
    lane_offsets(more ? chunk + 1 : chunk, vo_n, vm_n, alive_n);
    if (!(DBG & 16)) __syncthreads();            // buffer `buf` is complete; nobody still reads the other one
    // ---- this chunk's power-of-two scale from its window maximum
    unsigned Mb = sMax[buf * WN_NW];
#pragma unroll
    for (int i = 1; i < WN_NW; ++i) Mb = sMax[buf * WN_NW + i] > Mb ? sMax[buf * WN_NW + i] : Mb;
    int e_c = -100;
    if (Mb >= 0x7f800000u) {        // an infinity or a NaN in the window: this path is not exact for it -> the exact kernel re-runs
      ovfbits = 1u;
      e_c = 100;
    } else if (Mb > 0u) {
      frexpf(__uint_as_float(Mb), &e_c);                             // max < 2^e_c
      e_c = e_c < -100 ? -100 : (e_c > 100 ? 100 : e_c);
    }
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
    const unsigned char* const win = win_lane + buf * 2 * WN_WIN;
    const unsigned char* const wt = sWt + buf * WTS + lane * 16;
    unsigned fbmask = 0;                 // taps of this lane whose corners leave the window (redone below from global memory)
    typedef __fp16 hp2 __attribute__((ext_vector_type(2)));
    auto split_mma = [&](const float (&val)[8], const wn_h8 (&Ah)[MJ], const wn_h8 (&Al)[MJ], bool track) {
      // fp16 hi + lo with the packed round-toward-zero conversion (hi truncated, lo = the exact remainder truncated)
      union { hp2 h[4]; wn_h8 v8; } uh, ul;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uh.h[q] = __builtin_amdgcn_cvt_pkrtz(val[2 * q], val[2 * q + 1]);
        // remainders val - hi in ONE instruction each: v_fma_mix_f32 reads the fp16 half straight out of the packed register
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(uh.h[q]), "v"(val[2 * q]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(uh.h[q]), "v"(val[2 * q + 1]));
        ul.h[q] = __builtin_amdgcn_cvt_pkrtz(r0, r1);
        // range (the round-toward-zero conversion would clamp silently at 65504): a window sample is below 8 |mask| by
        // construction -- window values times s_in are below 2^3, the bilinear weights are in [0, 1] and sum to at most 1 -- so
        // the hot path only tracks the largest |mask| (one instruction per tap instead of four per K step); the cold path's
        // samples come from outside the window and are tracked value by value.  A NaN can only come from a NaN mask (the
        // reference's result is NaN there too) -- non-finite window data is caught by the window maximum
        if (track) vmax = fmaxf(vmax, fmaxf(fabsf(val[2 * q]), fabsf(val[2 * q + 1])));
      }
      if (DBG & 1) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[j][0] += (float)Ah[j][0] * (float)uh.v8[0] + (float)Al[j][1] * (float)ul.v8[1];
        return;
      }
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[j], uh.v8, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[j], uh.v8, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[j], ul.v8, acc[j], 0, 0, 0);
    };
    // next chunk's window: task 0 is requested now and written after step 2, task 1 then and written after step 4 (16 staging
    // registers at a time); its packed weights ride along
    unsigned wmax_n = 0u;
    f32x4 wr[4];
    if (more) {
      if (!(DBG & 4)) fetch_task(chunk + 1, 0, wr);
      fetch_weights(chunk + 1);
    }
#pragma unroll
    for (int s = 0; s < WN_STEPS; ++s) {
      wn_h8 Ah[MJ], Al[MJ];
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        Ah[j] = *reinterpret_cast<const wn_h8*>(wt + ((s * MJ + j) * 2 + 0) * 1024);
        Al[j] = *reinterpret_cast<const wn_h8*>(wt + ((s * MJ + j) * 2 + 1) * 1024);
      }
      float val[8];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = 2 * s + j;                     // compile-time tap
        if (t >= WN_T) {
#pragma unroll
          for (int e = 0; e < 4; ++e) val[4 * j + e] = 0.f;
          continue;
        }
        const float h_im = hbt[t / 3] + oh[t], w_im = wbt[t % 3] + ow[t];
        const float fh = floorf(h_im), fw = floorf(w_im);
        // inside the staged window?  (float compares: a NaN / infinite offset fails them and goes to the cold path, which applies
        // the reference's range test cu:617.  No such test is needed HERE: the window is zero-filled outside the image, so a
        // position the reference rejects -- all four corners outside -- samples zeros.)
        const bool use = alive && fh >= fwy0 && fh <= fwy1 && fw >= fwx0 && fw <= fwx1;
        fbmask |= (alive && !use) ? (1u << t) : 0u;
        // branch-free: fractions clamped (max / min drop a NaN from inf - inf), window coordinates clamped (an unused tap reads
        // somewhere harmless inside the window and gets weight 0)
        const float lh = fminf(fmaxf(h_im - fh, 0.f), 1.f), lw = fminf(fmaxf(w_im - fw, 0.f), 1.f);
        const int ly = min(max((int)fh - wy0, 0), WN_WH - 2), lx = min(max((int)fw - wx0, 0), WN_WW - 2);
        const int o = (ly * WN_WW + lx) * 16;
        f32x4 v1, v2, v3, v4;
        if (DBG & 2) {
          v1 = f32x4{lh, lw, lh, lw}; v2 = v1 * 0.5f; v3 = v1 * 0.25f; v4 = v1 + (float)o;
        } else {
          v1 = *reinterpret_cast<const f32x4*>(win + o);
          v2 = *reinterpret_cast<const f32x4*>(win + o + 16);
          v3 = *reinterpret_cast<const f32x4*>(win + o + WN_WW * 16);
          v4 = *reinterpret_cast<const f32x4*>(win + o + WN_WW * 16 + 16);
        }
        if (MASK) mmax = fmaxf(mmax, fabsf(mk[t]));
        const float ms = use ? mk[t] * s_in : 0.f;                 // an unused tap has weight 0 (finite: lh, lw were sanitised)
        const float mlh = lh * ms, mhh = ms - mlh;                 // (1 - lh) * ms
        const float w4 = mlh * lw, w3 = mlh - w4, w2 = mhh * lw, w1 = mhh - w2;
        {   // the four channels as two packed pairs: v_pk_mul_f32 / v_pk_fma_f32 (8 instead of 16 vector instructions per tap)
          typedef float wn_f2 __attribute__((ext_vector_type(2)));
          const wn_f2 W1 = {w1, w1}, W2 = {w2, w2}, W3 = {w3, w3}, W4 = {w4, w4};
          const wn_f2 lo = W4 * __builtin_shufflevector(v4, v4, 0, 1) + (W3 * __builtin_shufflevector(v3, v3, 0, 1) +
                           (W2 * __builtin_shufflevector(v2, v2, 0, 1) + W1 * __builtin_shufflevector(v1, v1, 0, 1)));
          const wn_f2 hi = W4 * __builtin_shufflevector(v4, v4, 2, 3) + (W3 * __builtin_shufflevector(v3, v3, 2, 3) +
                           (W2 * __builtin_shufflevector(v2, v2, 2, 3) + W1 * __builtin_shufflevector(v1, v1, 2, 3)));
          val[4 * j] = lo[0]; val[4 * j + 1] = lo[1]; val[4 * j + 2] = hi[0]; val[4 * j + 3] = hi[1];
        }
        if (!(DBG & 8)) load_tap(vo_n, vm_n, t, oh[t], ow[t], mk[t]);      // this tap's next offsets, into the registers just freed (the last
                                                           // chunk re-reads its own: harmless, and no branch in the stream)
      }
      split_mma(val, Ah, Al, false);
      if (WN_NTASK == 2 && more && s == 2 && !(DBG & 4)) {
        commit_task(buf ^ 1, 0, wr, wmax_n);
        fetch_task(chunk + 1, 1, wr);
      }
    }
    // ---- cold path, wave-uniform: some lane sampled beyond its window (offsets past ~10 pixels).  Those taps were zero above;
    // gather them from global memory now (zero outside the image, cu:481-488) and add their product -- the contraction is
    // linear.  (Their offsets are re-read: the registers already hold the next chunk's.)
    if (__builtin_expect(__ballot(fbmask != 0) != 0ull, 0)) {
#pragma unroll
      for (int s = 0; s < WN_STEPS; ++s) {
        if (__ballot(((fbmask >> (2 * s)) & 3u) != 0) == 0ull) continue;
        wn_h8 Ah[MJ], Al[MJ];
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          Ah[j] = *reinterpret_cast<const wn_h8*>(wt + ((s * MJ + j) * 2 + 0) * 1024);
          Al[j] = *reinterpret_cast<const wn_h8*>(wt + ((s * MJ + j) * 2 + 1) * 1024);
        }
        float val[8];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int t = 2 * s + j;
#pragma unroll
          for (int e = 0; e < 4; ++e) val[4 * j + e] = 0.f;
          if (t < WN_T && ((fbmask >> t) & 1u)) {
            float o_h, o_w, m_k;
            load_tap(vo_c, vm_c, t, o_h, o_w, m_k);
            float h_im = hb + (float)(t / 3) + o_h, w_im = wb + (float)(t % 3) + o_w;
            const bool valid = h_im > -1.f && w_im > -1.f && h_im < fH && w_im < fW;      // cu:617 (false for NaN)
            if (!valid) { h_im = 0.f; w_im = 0.f; m_k = 0.f; }
            const float fh = floorf(h_im), fw = floorf(w_im);
            const int hl = (int)fh, wl = (int)fw;
            const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
            const bool r0 = hl >= 0, r1 = hl + 1 <= H - 1, q0 = wl >= 0, q1 = wl + 1 <= W - 1;
            const int c0 = 8 * chunk + 4 * half;
            const float ms = m_k * s_in;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float* pl = in_b + (long long)(c0 + e) * HW;
              const float g1 = (r0 && q0) ? pl[hl * W + wl] : 0.f, g2 = (r0 && q1) ? pl[hl * W + wl + 1] : 0.f;
              const float g3 = (r1 && q0) ? pl[(hl + 1) * W + wl] : 0.f, g4 = (r1 && q1) ? pl[(hl + 1) * W + wl + 1] : 0.f;
              val[4 * j + e] = ((hh * hw) * g1 + (hh * lw) * g2 + (lh * hw) * g3 + (lh * lw) * g4) * ms;
            }
          }
        }
        split_mma(val, Ah, Al, true);
      }
    }
    if (more) {
      if (!(DBG & 4)) commit_task(buf ^ 1, WN_NTASK - 1, wr, wmax_n);
      commit_weights(buf ^ 1, wmax_n);
    }
  }
  if (!(vmax < 65504.f) || !(8.f * mmax < 65504.f)) ovfbits = 1u;
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

template <int MJ, bool AL4, bool MASK, int DBG = 0>
hipError_t wn_launch(const WinArgs& a, dim3 grid, hipStream_t st) {
  constexpr int LDSB = 4 * WN_WIN + 2 * (WN_STEPS * MJ * 2 * 1024) + 2 * WN_NW * 4;
  static_assert(LDSB <= 160 * 1024, "LDS budget");
  static CdfoAttrOnce once;
  const hipError_t e = cdfo_set_max_lds(once, reinterpret_cast<const void*>(&dcn_win_kernel<MJ, AL4, MASK, DBG>), LDSB);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((dcn_win_kernel<MJ, AL4, MASK, DBG>), grid, dim3(WN_THREADS), LDSB, st, a);
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
  if ((long long)C * H * W >= (1ll << 29) || (long long)dg * 2 * WN_T * Ho * Wo >= (1ll << 29)) return 0;      // 31-bit per-image byte offsets (0x80000000 = out of range)
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
  const int variant = (MJ == 2 ? 4 : 0) | (al4 ? 2 : 0) | (mask ? 1 : 0);
  switch (variant) {
    case 0: e = wn_launch<1, false, false>(a, grid, st); break;
    case 1: e = wn_launch<1, false, true>(a, grid, st); break;
    case 2: e = wn_launch<1, true, false>(a, grid, st); break;
    case 3: e = wn_launch<1, true, true>(a, grid, st); break;
    case 4: e = wn_launch<2, false, false>(a, grid, st); break;
    case 5: e = wn_launch<2, false, true>(a, grid, st); break;
    case 6: e = wn_launch<2, true, false>(a, grid, st); break;
    default: {
#ifdef CDFO_DEV_ABLATIONS
      // developer ablations of the alignment module's variant (tools/bench_dcn.py; results are wrong by construction): compiled only
      // into developer builds (-DCDFO_DEV_ABLATIONS) -- the shipped library reads no environment variable on this path
      static const int dbg_sel = [] { const char* v = getenv("CDFO_DCN_DBG"); return v ? atoi(v) : 0; }();      // read once per process
      switch (dbg_sel) {
        case 1: e = wn_launch<2, true, true, 1>(a, grid, st); break;
        case 2: e = wn_launch<2, true, true, 2>(a, grid, st); break;
        case 4: e = wn_launch<2, true, true, 4>(a, grid, st); break;
        case 8: e = wn_launch<2, true, true, 8>(a, grid, st); break;
        case 16: e = wn_launch<2, true, true, 16>(a, grid, st); break;
        case 3: e = wn_launch<2, true, true, 3>(a, grid, st); break;
        case 12: e = wn_launch<2, true, true, 12>(a, grid, st); break;
        case 15: e = wn_launch<2, true, true, 15>(a, grid, st); break;
        case 31: e = wn_launch<2, true, true, 31>(a, grid, st); break;
        default: e = wn_launch<2, true, true>(a, grid, st); break;
      }
#else
      e = wn_launch<2, true, true>(a, grid, st);
#endif
      break;
    }
  }
  return e == hipSuccess ? 1 : 2 + (int)e;
}
