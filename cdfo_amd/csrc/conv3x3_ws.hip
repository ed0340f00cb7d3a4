// 3x3 stride-1 convolution with 64 input channels, weights-stationary and barrier-free (fp16 MFMA, fp32 accumulate).
//
// Block_.body[0] (arch/SIDECVSR_our.py:383-387: Conv2d(64, 256, 3, 1, 1) + LeakyReLU) runs three times per block (own,
// half and double resolution) and is half of the forward's FLOPs.  With 64 input channels the fp16 weights of one
// 64-output-channel block are 73,728 bytes: they fit the CU's 160 KB LDS next to the activations, so this kernel
//   * is persistent: one 512-thread workgroup per CU loads its weight block ONCE and then walks over pixel tiles
//     (cdfo_conv3x3_bf16 re-stages 18 KB of weights per 16-channel chunk of every 256-pixel tile: ~1.1 KB of
//     L2->LDS traffic per output pixel, more than the activations);
//   * has NO workgroup barrier after the weight load: every wave owns a 2-row x 32-pixel x 64-channel output tile and
//     stages its own 4 x 34 pixel halo, 16 channels at a time, straight from global memory into a wave-private LDS
//     ring by LDS-DMA (buffer_load_dwordx4 ... lds: no staging VGPRs, no ds_write pass; out-of-image lanes are given an
//     out-of-range buffer offset, which the hardware zero-fills = the convolution's zero padding).  The eight waves
//     drift apart freely, so one wave's DMA wait or epilogue is covered by its SIMD partner's MFMAs; tiles are handed
//     out through an LDS counter (the two waves of a SIMD do not progress at the same rate);
//   * reads a "chunk-planar" fp16 source [B][Cin/16][H][W][16]: a staged image row is 34 x 32 contiguous bytes, every
//     DMA piece moves whole cache lines (the pixel-major layout would touch 32 bytes of each 128-byte line);
//   * computes the transposed product (M = output channels, N = pixels) with the weight rows of each 32-channel block
//     permuted so that a lane's accumulator registers hold 8 CONSECUTIVE channels of one pixel per 16-channel chunk:
//     the epilogue is activation + fp16 pack + one 16-byte store per lane and chunk straight from registers into the
//     chunk-planar result [B][Cout/16][H][W][16] (a wave-instruction writes 1 KiB contiguous) -- no LDS transpose.
//     The result is the source layout of cdfo_conv3x3_ring (Block_.body[2]); CDFO_STORE_S2D writes the
//     space-to-depth form [B][4*Cout/16][H/2][W/2][16] (chunk = phase*Cout/16 + channel/16) for the composed
//     stride-2 convolution of the double-resolution branch.
// The LDS image is dense (32 bytes per pixel) because LDS-DMA writes lane-linear; bank conflicts of the ds_read_b128
// fragment reads are removed by swapping the two 16-byte halves of every second group of 8 pixels, applied on the DMA
// source side and on the read side.
//
// DMA completion is tracked by hand (hipcc does not count inline-asm memory operations) with counted s_waitcnt; the
// rules are written next to each wait.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int WS_W_BYTES = 4 * 9 * 2 * 64 * 16;            // 73,728: [chunk][tap][k-half][64 cout][8 x fp16]
constexpr int WS_BIAS_OFF = WS_W_BYTES;                    // 64 floats
constexpr int WS_STG_OFF = WS_W_BYTES + 256;
constexpr int WS_BUF = 5 * 1024;                           // one 16-channel chunk of a wave's halo: 136 px x 32 B in 5 DMA pieces
// NW waves per workgroup: 8 (two per SIMD, each with a 2-deep private staging ring) or 12 (three per SIMD, ONE staging
// buffer each: 12 x 2 x 5 KB would not fit next to the weights)
template <int NW> struct WsLds {
  static constexpr int NBUF = NW == 8 ? 2 : 1;
  static constexpr int CNT_OFF = WS_STG_OFF + NW * NBUF * WS_BUF;    // work counter of the workgroup
  static constexpr int TOTAL = CNT_OFF + 16;                         // 155,920 bytes (8 waves) / 135,440 (12 waves)
};
constexpr int WS_IW = 34, WS_NPIX = 4 * 34;

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ws_args {
  const void* src; unsigned src_bytes;
  int B, H, W;
  const unsigned short* w; int CoutP;       // cdfo_pack_conv3x3_f16 packing, CoutP = its padded channel count
  const float* bias;
  int Cout, act;
  _Float16* out; int s2d;        // fp16 chunk-planar result (optional when out32 is given)
  float* out32; int ldo32;       // RES form: fp32 pixel-major result = act(conv + bias) + res1 (+ res2)
  const float* res1; int ldr1;
  const float* res2; int ldr2;
  unsigned long long* clk;      // developer probe (dbg 128): per wave {shader-clock cycles, start, end in 100 MHz real-time ticks}
};

// five 1 KiB LDS-DMA pieces: lane l of piece k writes LDS bytes lds + 1024 k + 16 l from (buffer base + voff[k] + soff)
__device__ __forceinline__ void ws_dma5(const unsigned (&voff)[5], i32x4 rsrc, unsigned soff, unsigned lds) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %7\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %1, %6, %8 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %2, %6, %8 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %3, %6, %8 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %4, %6, %8 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %5, %6, %8 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "s"(rsrc), "s"(lds), "s"(soff)
      : "memory", "scc");
}

// DBG (developer ablations): 1 = skip the MFMAs, 2 = skip the DMA, 8 = skip the epilogue, 64 = no start-up stagger,
// 128 = clock probe
// RES: residual form -- fp32 pixel-major output with one or two fp32 residual inputs (ResidualBlock_noBN's second
// convolution, arch.py:261-262), optionally also the fp16 chunk-planar copy.  The residual values of a tile are fetched
// by inline-asm loads while its last chunk computes (hipcc would wait for a visible load with vmcnt(0) and so drain the
// DMA that is in flight behind it) and consumed behind a counted wait.
template <int DBG, bool RES = false, int NW = 8>
__global__ __launch_bounds__(NW * 64) void conv3x3_c64_ws_kernel(ws_args a) {
  constexpr int WS_THREADS = NW * 64, NBUF = WsLds<NW>::NBUF, WS_CNT_OFF = WsLds<NW>::CNT_OFF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long clk0 = 0, rt0 = 0;
  if (DBG & 128) { clk0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
  // workgroups are dealt round-robin over the 8 XCDs: the nco output-channel blocks of one pixel partition share an L2
  const int nco = a.Cout >> 6;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nb = slot % nco, q = slot / nco;
  const int nparts = ((int)(gridDim.x >> 3) / nco) * 8, part = q * 8 + xcd;
  const int n0 = nb * 64;
  const int H = a.H, W = a.W;

  // ---- one-time: this workgroup's weight block and bias
  // MFMA row m = 8j + 4h + k of a 32-channel block holds channel (j>>1)*16 + h*8 + (j&1)*4 + k, so that a lane's
  // accumulators (fixed h; j, k = register index) are 8 consecutive channels of each 16-channel chunk
  auto chan_of_row = [](int n) { const int m = n & 31; return (n & 32) + ((m >> 4) & 1) * 16 + ((m >> 2) & 1) * 8 + ((m >> 3) & 1) * 4 + (m & 3); };
  for (int i = tid; i < WS_W_BYTES / 16; i += WS_THREADS) {
    const int row = i >> 6, n = i & 63;          // row = (chunk*9 + tap)*2 + k-half
    *reinterpret_cast<u32x4*>(smem + i * 16) =
        *reinterpret_cast<const u32x4*>(a.w + ((long long)row * a.CoutP + n0 + chan_of_row(n)) * 8);
  }
  if (tid < 64) reinterpret_cast<float*>(smem + WS_BIAS_OFF)[tid] = a.bias ? a.bias[n0 + chan_of_row(tid)] : 0.f;
  unsigned* s_next = reinterpret_cast<unsigned*>(smem + WS_CNT_OFF);
  if (tid == 0) *s_next = NW;                    // tiles 0..NW-1 of the workgroup go to its waves, the rest first come first served
  __syncthreads();

  // ---- per-lane constants
  // DMA slot s = 64 k + lane holds pixel p = s >> 1 of the 4 x 34 halo, k-half (s & 1) ^ ((p >> 3) & 1)
  int d_iy[5], d_ix[5], d_rel[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int s = lane + 64 * k, p = s >> 1, half = (s & 1) ^ ((p >> 3) & 1);
    const int iy = p / WS_IW, ix = p - iy * WS_IW;
    d_iy[k] = p < WS_NPIX ? iy : 1 << 20;          // pad slots: never inside the image
    d_ix[k] = ix;
    d_rel[k] = (iy * W + ix) * 32 + half * 16;
  }
  // fragment read offsets: halo row rr (0..3), column offset dx (0..2), this lane's pixel r
  int p_off[12];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int p = rr * WS_IW + dx + r;
      p_off[rr * 3 + dx] = (2 * p + (h ^ ((p >> 3) & 1))) * 16;
    }
  const unsigned char* sWl = smem + (h * 64 + r) * 16;
  unsigned char* stg = smem + WS_STG_OFF + wave * (NBUF * WS_BUF);
  const unsigned stg_lds = (unsigned)(unsigned long long)(smem) + WS_STG_OFF + wave * (NBUF * WS_BUF);

  i32x4 rsrc;
  {
    const unsigned long long p = reinterpret_cast<unsigned long long>(a.src);
    rsrc[0] = (int)(unsigned)p;
    rsrc[1] = (int)(unsigned)(p >> 32);
    rsrc[2] = (int)a.src_bytes;
    rsrc[3] = 0x00020000;
  }
  const int tiles_x = (W + 31) >> 5, hp = H >> 1;
  const int total = a.B * hp * tiles_x;
  const unsigned plane = (unsigned)(H * W) * 32u;        // bytes of one 16-channel plane of one image
  // The workgroup's i-th tile is tile (i & 7) of its (i >> 3)-th group of 8 consecutive tiles.  Tiles are handed out
  // through an LDS counter: the two waves of a SIMD do not progress at the same rate (the older wave wins the issue
  // arbitration), and with a static split the faster waves idled for the last ~10 % of the launch.
  auto tile_of = [&](int i) { return ((i >> 3) * nparts + part) * 8 + (i & 7); };
  auto grab = [&]() {
    unsigned i = 0;
    if (lane == 0) i = __hip_atomic_fetch_add(s_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return tile_of((int)__builtin_amdgcn_readfirstlane(i));
  };
  int u = tile_of(wave), unext = 0;

  unsigned voff[5], soff0 = 0;
  int ub = 0, uy0 = 0, ux0 = 0;
  auto make_desc = [&](int uu) {
    const int xs = uu % tiles_x, t2 = uu / tiles_x;
    const int yp = t2 % hp;
    ub = t2 / hp; uy0 = yp * 2; ux0 = xs * 32;
    soff0 = (unsigned)ub * 4u * plane;
    const int base = ((uy0 - 1) * W + (ux0 - 1)) * 32;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int gy = uy0 - 1 + d_iy[k], gx = ux0 - 1 + d_ix[k];
      const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
      voff[k] = ok ? (unsigned)(base + d_rel[k]) : 0x80000000u;     // out of range => the DMA writes zeros
    }
  };

  const float slope = a.act == CDFO_ACT_NONE ? 1.f : (a.act == CDFO_ACT_LRELU ? 0.1f : 0.f);

  if (u < total) {
    make_desc(u);
    if (!(DBG & 2)) ws_dma5(voff, rsrc, soff0, stg_lds);
  }
  // Start-up stagger.  The waves run the same program with no barrier, so the two waves of a SIMD (w, w+4) would stay
  // in lockstep -- sharing the matrix pipe during their MFMA phases and leaving it idle during their simultaneous
  // epilogues.  A lead, once given, persists (neither wave waits for the other): waves 4-7 start half a tile's MFMA
  // time late, and the four SIMDs are spread over the rest so that their epilogues do not collide in LDS / the
  // store path either.
  if (!(DBG & 64)) {
    const int naps = (wave >> 2) * 4 + (wave & 3);          // x 576 cycles (tile MFMA time ~4,600 cycles)
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(9);
  }
  for (; u < total; u = unext) {
    const int cb = ub, cy0 = uy0, cx0 = ux0;     // this unit (make_desc below moves on to the next one)
    f32x16 acc[2][2];                            // [ni: 32-channel block][mi: image row]
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + WS_BIAS_OFF + (ni * 32 + 8 * j + 4 * h) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[ni][0][4 * j + k] = bv[k]; acc[ni][1][4 * j + k] = bv[k]; }
      }

    f32x4 rv[RES ? 16 : 1];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // wait for this chunk's image, issued during the previous chunk (below): no DMA is younger than it, so the wait
      // is vmcnt(0).  (A counted wait that lets the previous tile's epilogue stores stay in flight would assume that
      // stores and LDS-DMA loads retire in issue order; measured: VGPR loads and LDS-DMA loads do NOT -- a vmcnt(5)
      // behind 16 residual loads + 5 DMA pieces returned before the older residual loads had landed -- so counted waits
      // here only ever count younger DMA pieces, which do retire in order among themselves.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned char* sA = stg + (NBUF == 2 ? (c & 1) * WS_BUF : 0);
      const unsigned char* sWc = sWl + c * (9 * 2 * 64 * 16);
      f16x8_t fp[2][2], fw[2][2];                // [parity][mi / ni]: fragments are read one tap ahead
      auto load_frags = [&](int t, int par) {
        const int dy = t / 3, dx = t - dy * 3;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) fp[par][mi] = *reinterpret_cast<const f16x8_t*>(sA + p_off[(mi + dy) * 3 + dx]);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) fw[par][ni] = *reinterpret_cast<const f16x8_t*>(sWc + (t * 2 * 64 + ni * 32) * 16);
      };
      load_frags(0, 0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t < 8) load_frags(t + 1, (t & 1) ^ 1);
        // keep the NEXT tap's fragment reads in front of this tap's MFMAs: left to itself hipcc sinks them behind the MFMAs
        // and reads each weight fragment right before its use (a counted wait on a just-issued ds_read per tap)
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) {
          // Next chunk's DMA, into the OTHER ring buffer, whose last readers were the previous chunk's fragment reads.
          // Those must have returned before a DMA piece can land (one that hits in the CU's L1 lands within ~100
          // cycles; issuing without this wait corrupted ~1 tile per launch).  LDS operations of a wave return in
          // order and the 8 reads of taps 0 and 1 above are younger than all of them, so "at most 8 outstanding"
          // is enough -- and free, those 8 are about to be waited for anyway.
          asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
          if (RES && c == 3) {
            // 16 residual loads: pixel (row cy0 + mi, column cx0 + r), channels n0 + ni*32 + jj*16 + h*8 + 4*hf .. +3.
            // Always issued (a masked pixel reads its clamped neighbour), so that the wait below can count them.
            const int xc = cx0 + r < W ? cx0 + r : W - 1;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int mi = i >> 3, ni = (i >> 2) & 1, jj = (i >> 1) & 1, hf = i & 1;
              const float* rp = a.res1 + ((long long)(cb * H + cy0 + mi) * W + xc) * a.ldr1 + n0 + ni * 32 + jj * 16 + h * 8 + hf * 4;
              asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rv[i]) : "v"(rp) : "memory");
            }
          }
          if (NBUF == 2) {
            if (c < 3) {
              if (!(DBG & 2)) ws_dma5(voff, rsrc, soff0 + (unsigned)(c + 1) * plane, stg_lds + ((c + 1) & 1) * WS_BUF);
            } else if ((unext = grab()) < total) {
              make_desc(unext);
              if (!(DBG & 2)) ws_dma5(voff, rsrc, soff0, stg_lds);
            }
          }
        }
        if (NBUF == 1 && t == 8) {
          // single staging buffer: the next chunk's DMA may only be issued once EVERY fragment read of this chunk has
          // returned (tap 8's were issued one tap ago); it then flies behind tap 8's MFMAs and, between tiles, the epilogue,
          // while the SIMD's other two waves keep the matrix pipe busy
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (c < 3) {
            if (!(DBG & 2)) ws_dma5(voff, rsrc, soff0 + (unsigned)(c + 1) * plane, stg_lds);
          } else if ((unext = grab()) < total) {
            make_desc(unext);
            if (!(DBG & 2)) ws_dma5(voff, rsrc, soff0, stg_lds);
          }
        }
        const int par = t & 1;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            if (DBG & 1) acc[ni][mi][0] += (float)fw[par][ni][0] * (float)fp[par][mi][0];
            else acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[par][ni], fp[par][mi], acc[ni][mi], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- epilogue: act -> fp16 -> 16-byte stores.  acc[ni][mi][8jj + 4b + k] = channel ni*32 + jj*16 + h*8 + b*4 + k of
    // pixel r in image row mi, i.e. halves [h*8, h*8+8) of chunk nb*4 + ni*2 + jj.
    if (DBG & 8) {
      float t = 0.f;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int e = 0; e < 16; ++e) t += acc[ni][mi][e];
      if (t == 123.456f) a.out[0] = (_Float16)t;
      continue;
    }
    const int nck = a.Cout >> 4;                     // 16-channel chunks of the result
    const bool xok = cx0 + r < W;
    if (RES) {
      // Everything outstanding is waited for: the residual loads AND the DMA of the next tile's first chunk, issued right
      // after them a whole chunk of MFMAs ago (so it has landed; a counted vmcnt(5) here returned garbage residuals:
      // LDS-DMA loads and VGPR loads do not retire in issue order relative to each other).
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]), "+v"(rv[4]), "+v"(rv[5]), "+v"(rv[6]),
                   "+v"(rv[7]), "+v"(rv[8]), "+v"(rv[9]), "+v"(rv[10]), "+v"(rv[11]), "+v"(rv[12]), "+v"(rv[13]), "+v"(rv[14]),
                   "+v"(rv[15]) : : "memory");
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int y = cy0 + mi, x = cx0 + r;
        const long long pix = (long long)(cb * H + y) * W + x;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int n = n0 + ni * 32 + jj * 16 + h * 8;
            f32x4 v0, v1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float t0 = acc[ni][mi][8 * jj + k], t1 = acc[ni][mi][8 * jj + 4 + k];
              v0[k] = fmaxf(t0, slope * t0);
              v1[k] = fmaxf(t1, slope * t1);
            }
            v0 += rv[mi * 8 + ni * 4 + jj * 2];
            v1 += rv[mi * 8 + ni * 4 + jj * 2 + 1];
            if (!xok) continue;
            if (a.res2) {
              const float* p2 = a.res2 + pix * a.ldr2 + n;
              v0 += *reinterpret_cast<const f32x4*>(p2);
              v1 += *reinterpret_cast<const f32x4*>(p2 + 4);
            }
            *reinterpret_cast<f32x4*>(a.out32 + pix * a.ldo32 + n) = v0;
            *reinterpret_cast<f32x4*>(a.out32 + pix * a.ldo32 + n + 4) = v1;
            if (a.out) {
              f16x8_t hv;
#pragma unroll
              for (int k = 0; k < 4; ++k) { hv[k] = (_Float16)v0[k]; hv[4 + k] = (_Float16)v1[k]; }
              *reinterpret_cast<f16x8_t*>(a.out + ((((long long)cb * nck + nb * 4 + ni * 2 + jj) * H + y) * W + x) * 16 + h * 8) = hv;
            }
          }
      }
      continue;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int y = cy0 + mi, x = cx0 + r;
      long long pos;      // (chunk-plane 0 of this workgroup's block, pixel) in 16-half records
      long long cstride;  // records between consecutive chunk planes
      if (a.s2d) {
        const int ph = (y & 1) * 2 + (x & 1);
        cstride = (long long)hp * (W >> 1);
        pos = ((long long)cb * 4 * nck + ph * nck + nb * 4) * cstride + (long long)(y >> 1) * (W >> 1) + (x >> 1);
      } else {
        cstride = (long long)H * W;
        pos = ((long long)cb * nck + nb * 4) * cstride + (long long)y * W + x;
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          f16x8_t hv;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float v = acc[ni][mi][8 * jj + q];
            hv[q] = (_Float16)fmaxf(v, slope * v);      // slope in [0, 1]: identity / LeakyReLU / ReLU
          }
          if (xok) *reinterpret_cast<f16x8_t*>(a.out + (pos + (ni * 2 + jj) * cstride) * 16 + h * 8) = hv;
        }
    }
  }
  if ((DBG & 128) && a.clk && (tid & 63) == 0) {       // per wave: {shader cycles, start, end} (100 MHz real-time ticks)
    unsigned long long* c = a.clk + (blockIdx.x * 8 + wave) * 3;
    c[0] = __builtin_readcyclecounter() - clk0;
    c[1] = rt0;
    c[2] = __builtin_amdgcn_s_memrealtime();
  }
}

int ws_num_cus() { return cdfo_num_cus(); }

template <int DBG, bool RES = false>
int ws_launch(const ws_args& a, int grid, hipStream_t st) {
  // three waves per SIMD by default (same-box A/B: 1.21 -> 1.18 ms at 64 -> 256 on 8 x 544 x 960, 0.295 -> 0.286 at 272 x 480):
  // with two, the matrix pipe idles whenever both are outside their MFMA runs at once (counters: pipe busy 67 %);
  // CDFO_WS_WAVES=8 selects the two-per-SIMD form with its 2-deep staging rings (developer A/B switch)
  static const bool w12 = [] { const char* e = getenv("CDFO_WS_WAVES"); return !(e && atoi(e) == 8); }();
  // (four per SIMD -- 16 waves at 128 VGPRs, 8 of them spilled -- was 5 % slower than three)
  if (DBG == 0 && !RES && w12) {     // (the residual form needs 221 VGPRs: two waves per SIMD only)
    static CdfoAttrOnce once12;
    const hipError_t e = cdfo_set_max_lds(once12, reinterpret_cast<const void*>(conv3x3_c64_ws_kernel<DBG, RES, 12>), WsLds<12>::TOTAL);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv3x3_c64_ws_kernel<DBG, RES, 12>), dim3(grid), dim3(12 * 64), WsLds<12>::TOTAL, st, a);
    return 0;
  }
  static CdfoAttrOnce once;
  const hipError_t e = cdfo_set_max_lds(once, reinterpret_cast<const void*>(conv3x3_c64_ws_kernel<DBG, RES, 8>), WsLds<8>::TOTAL);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((conv3x3_c64_ws_kernel<DBG, RES, 8>), dim3(grid), dim3(8 * 64), WsLds<8>::TOTAL, st, a);
  return 0;
}

// fp32 pixel-major [B][P][ld] -> fp16 chunk-planar [B][C/16][P][16]
__global__ __launch_bounds__(256) void to_cp16_kernel(const float* __restrict__ in, int ldi, int B, long long P, int C,
                                                      _Float16* __restrict__ out) {
  const int nc = C >> 4;
  const long long total = (long long)B * nc * P * 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = i & 3;
    const long long t = i >> 2;
    const long long p = t % P;
    const int c = (t / P) % nc;
    const long long b = t / (P * nc);
    const f32x4 v = *reinterpret_cast<const f32x4*>(in + (b * P + p) * ldi + c * 16 + g * 4);
    f16x4_t hv;
#pragma unroll
    for (int k = 0; k < 4; ++k) hv[k] = (_Float16)v[k];
    *reinterpret_cast<f16x4_t*>(out + i * 4) = hv;
  }
}

}  // namespace

extern "C" int cdfo_conv3x3_c64_ws(const void* src_cp16, int B, int H, int W, const void* w_f16, int CoutP,
                                   const float* bias, int Cout, int act, void* out_cp16, int store_mode, int dbg,
                                   void* clk_probe, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || Cout <= 0 || Cout % 64 || CoutP < Cout || CoutP % 64) return CDFO_EINVAL;
  if (act == CDFO_ACT_SIGMOID || (store_mode != CDFO_STORE_PLAIN && store_mode != CDFO_STORE_S2D)) return CDFO_EINVAL;
  if (store_mode == CDFO_STORE_S2D && (W & 1)) return CDFO_EINVAL;
  const long long src_bytes = (long long)B * 4 * H * W * 32;
  if (src_bytes >= (1ll << 31)) return CDFO_EINVAL;      // 32-bit buffer offsets, out-of-range marker 0x80000000
  if (!aligned16(src_cp16) || !aligned16(w_f16) || !aligned16(out_cp16)) return CDFO_EALIGN;
  const int cus = ws_num_cus();
  if (cus < 8) return CDFO_EINVAL;
  const int nco = Cout / 64;
  int qn = (cus / 8) / nco;
  if (qn < 1) qn = 1;
  const int grid = 8 * qn * nco;
  ws_args a;
  a.src = src_cp16; a.src_bytes = (unsigned)src_bytes;
  a.B = B; a.H = H; a.W = W;
  a.w = static_cast<const unsigned short*>(w_f16); a.CoutP = CoutP;
  a.bias = bias; a.Cout = Cout; a.act = act;
  a.out = static_cast<_Float16*>(out_cp16); a.s2d = store_mode == CDFO_STORE_S2D;
  a.out32 = nullptr; a.ldo32 = 0; a.res1 = nullptr; a.ldr1 = 0; a.res2 = nullptr; a.ldr2 = 0;
  a.clk = static_cast<unsigned long long*>(clk_probe);
  const double px = (double)B * H * W;
  CdfoProfScope prof(st, KID_CONV3_WS, 2.0 * px * Cout * 64 * 9, 2.0 * (px * Cout + px * 64) + 2.0 * 9 * 64 * Cout);
  int rc;
  switch (dbg) {
    case 0: rc = ws_launch<0>(a, grid, st); break;
    case 1: rc = ws_launch<1>(a, grid, st); break;
    case 2: rc = ws_launch<2>(a, grid, st); break;
    case 3: rc = ws_launch<3>(a, grid, st); break;
    case 8: rc = ws_launch<8>(a, grid, st); break;
    case 9: rc = ws_launch<9>(a, grid, st); break;
    case 11: rc = ws_launch<11>(a, grid, st); break;
    case 16: rc = ws_launch<16>(a, grid, st); break;
    case 64: rc = ws_launch<64>(a, grid, st); break;
    case 72: rc = ws_launch<72>(a, grid, st); break;
    case 128: rc = ws_launch<128>(a, grid, st); break;
    case 136: rc = ws_launch<136>(a, grid, st); break;
    default: return CDFO_EINVAL;
  }
  if (rc) return rc;
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_conv3x3_c64_ws_res(const void* src_cp16, int B, int H, int W, const void* w_f16, int CoutP,
                                       const float* bias, int Cout, int act, float* out, int ldo, const float* res1,
                                       int ldr1, const float* res2, int ldr2, void* out2_cp16, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || Cout <= 0 || Cout % 64 || CoutP < Cout || CoutP % 64) return CDFO_EINVAL;
  if (act == CDFO_ACT_SIGMOID || !out || !res1 || ldo % 4 || ldo < Cout || ldr1 % 4 || ldr1 < Cout) return CDFO_EINVAL;
  if (res2 && (ldr2 % 4 || ldr2 < Cout)) return CDFO_EINVAL;
  const long long src_bytes = (long long)B * 4 * H * W * 32;
  if (src_bytes >= (1ll << 31)) return CDFO_EINVAL;
  if (!aligned16(src_cp16) || !aligned16(w_f16) || !aligned16(out) || !aligned16(res1) || (res2 && !aligned16(res2)) ||
      (out2_cp16 && !aligned16(out2_cp16)))
    return CDFO_EALIGN;
  const int cus = ws_num_cus();
  if (cus < 8) return CDFO_EINVAL;
  const int nco = Cout / 64;
  int qn = (cus / 8) / nco;
  if (qn < 1) qn = 1;
  ws_args a;
  a.src = src_cp16; a.src_bytes = (unsigned)src_bytes;
  a.B = B; a.H = H; a.W = W;
  a.w = static_cast<const unsigned short*>(w_f16); a.CoutP = CoutP;
  a.bias = bias; a.Cout = Cout; a.act = act;
  a.out = static_cast<_Float16*>(out2_cp16); a.s2d = 0;
  a.out32 = out; a.ldo32 = ldo; a.res1 = res1; a.ldr1 = ldr1; a.res2 = res2; a.ldr2 = ldr2;
  a.clk = nullptr;
  const double px = (double)B * H * W;
  CdfoProfScope prof(st, KID_CONV3_WS_RES, 2.0 * px * Cout * 64 * 9, px * (2.0 * 64 + 8.0 * Cout + (res2 ? 4.0 * Cout : 0.0)));
  const int rc = ws_launch<0, true>(a, 8 * qn * nco, st);
  if (rc) return rc;
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_to_cp16(const float* in, int ldi, int B, long long P, int C, void* out_cp16, void* stream) {
  if (B <= 0 || P <= 0 || C <= 0 || C % 16 || ldi % 4 || ldi < C) return CDFO_EINVAL;
  if (!aligned16(in) || !aligned16(out_cp16)) return CDFO_EALIGN;
  const long long threads = (long long)B * (C / 16) * P * 4;
  const long long blocks = (threads + 255) / 256;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_LAYOUT, 0, 6.0 * C * (double)B * P);
  hipLaunchKernelGGL(to_cp16_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, ldi, B, P, C, static_cast<_Float16*>(out_cp16));
  CDFO_LAUNCH_CHECK();
  return 0;
}
