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
  // offset / mask epilogue (conv3x3_c64_wsq_kernel<0, true>; cdfo_conv3x3_c64_ws_offmask): see cdfo_hip.h, CDFO_STORE_OFFMASK
  float* om_offset; float* om_mask; const float* om_flow; long long om_flow_bstride; float om_mag; int om_accumulate; int om_cout;
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

// ---------------------------------------------------------------------------------------------------------------------
// Ring-fed, wave-specialised form of the non-residual kernel (round 3).  Same product, same fp16 chunk-planar / space-to-depth
// result, same weight image in LDS; what changes is who does what:
//   * the workgroup (12 waves) owns a 16-row x 32-pixel tile at a time, as two 8-row HALF-TILES; a half-tile's 10 x 34 pixel halo
//     of a 16-channel chunk is staged once and shared by the four waves that consume it (the form above stages a private 4 x 34
//     halo per wave: 2.1x the tile's bytes through LDS-DMA instead of 1.3x);
//   * waves 0-7 are CONSUMERS, two per SIMD: waves 0-3 (group A) take the upper half-tile, waves 4-7 (group B) the lower one,
//     rows 2w, 2w + 1 of it each -- nothing in their instruction stream but fragment reads and MFMAs, then the 16-byte stores
//     of the epilogue.  Group B runs TWO CHUNKS (half a tile) behind group A: a SIMD's two consumers never reach their
//     epilogues (activation + fp16 pack + stores: ~18 % of the launch when all eight did it at once) together, one keeps the
//     matrix pipe busy while the other converts and stores;
//   * waves 8-11 (one per SIMD) are PRODUCERS: waves 8, 9 feed group A's ring (6 + 5 DMA pieces per chunk), waves 10, 11
//     group B's, each ring four stages deep (two chunks in flight behind the one being consumed), across tile boundaries;
//   * one workgroup barrier per step: a producer arrives when ITS pieces of the batch its group consumes next have landed
//     (counted vmcnt), a consumer when its fragment reads of the previous batch have returned; behind the barrier the
//     producers refill the stage of the batch just retired.
// Why: the timeline probe of the ring kernel (conv3x3_ring.hip, dbg 16) showed that a wave which interleaves DMA issue (100-200
// cycles per instruction) and ring bookkeeping with its MFMAs needs 250-400 cycles per tap for 128 cycles of matrix work, and
// that consumer-only waves reach 86 % matrix-pipe duty inside a chunk; the form above sits at 67 % over the launch.
// LDS: 2 rings x 4 stages x 11 KiB + the 72 KiB weight image = exactly 160 KiB (the bias lives in registers).
constexpr int WR_HT = 8, WR_IW = 34, WR_NPIX = 10 * 34, WR_PIECES = 11, WR_ACT = WR_PIECES * 1024, WR_NS = 4;
constexpr int WR_CONS = 8, WR_THREADS = 12 * 64, WR_LAG = 2;      // group B runs WR_LAG steps behind group A
#ifndef WR_PRIO
#define WR_PRIO 1
#endif
constexpr int WR_W_OFF = 2 * WR_NS * WR_ACT;           // 90,112
constexpr int WR_TOTAL = WR_W_OFF + WS_W_BYTES;        // 163,840 bytes
static_assert(WR_TOTAL <= 160 * 1024, "LDS budget");

// six / five 1 KiB LDS-DMA pieces into consecutive kilobytes, ONE asm block each (m0 is live across the pieces; see ws_dma5)
__device__ __forceinline__ void wr_dma6(const unsigned (&voff)[6], i32x4 rsrc, unsigned soff, unsigned lds) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %8\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %1, %7, %9 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %2, %7, %9 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %3, %7, %9 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %4, %7, %9 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %5, %7, %9 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %6, %7, %9 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "v"(voff[5]), "s"(rsrc), "s"(lds), "s"(soff)
      : "memory", "scc");
}
__device__ __forceinline__ void wr_dma5(const unsigned (&voff)[6], i32x4 rsrc, unsigned soff, unsigned lds) {
  const unsigned v5[5] = {voff[0], voff[1], voff[2], voff[3], voff[4]};
  ws_dma5(v5, rsrc, soff, lds);
}

// DBG: 1 = skip the MFMAs, 2 = skip the DMA, 8 = skip the epilogue (developer ablations); 32 = timeline probe: a.clk receives
// s_memtime stamps [workgroup][wave 0-11][4][8] of the consumers' third tile (per chunk: entry, barrier passed, after taps 2 / 5 /
// 8, end) and, in slot 6 of rows 0 / 1, the epilogue-start times of the fourth and fifth tile (undisturbed tile period)
template <int DBG>
__global__ __launch_bounds__(WR_THREADS) void conv3x3_c64_wsr_kernel(ws_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave < WR_CONS;
  const int H = a.H, W = a.W;
  // the nco output-channel blocks of a tile sequence sit on one XCD (shared L2); each XCD walks its own band of tiles
  const int nco = a.Cout >> 6;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int nq = nslots / nco, nb = slot % nco, q = slot / nco;
  const int tiles_x = (W + 31) >> 5, tiles_y = (H + 2 * WR_HT - 1) / (2 * WR_HT), tiles = tiles_x * tiles_y;
  const int all_tiles = a.B * tiles;
  const int band = (all_tiles + 7) >> 3, band0 = xcd * band;
  const int band_n = min(band, all_tiles - band0);
  const int my_tiles = (q < nq && band_n > q) ? (band_n - q + nq - 1) / nq : 0;
  if (my_tiles == 0) return;
  const int n0 = nb * 64;

  // ---- one-time: this workgroup's weight block (row permutation as in the kernel above)
  auto chan_of_row = [](int n) { const int m = n & 31; return (n & 32) + ((m >> 4) & 1) * 16 + ((m >> 2) & 1) * 8 + ((m >> 3) & 1) * 4 + (m & 3); };
  for (int i = tid; i < WS_W_BYTES / 16; i += WR_THREADS) {
    const int row = i >> 6, n = i & 63;          // row = (chunk*9 + tap)*2 + k-half
    *reinterpret_cast<u32x4*>(smem + WR_W_OFF + i * 16) =
        *reinterpret_cast<const u32x4*>(a.w + ((long long)row * a.CoutP + n0 + chan_of_row(n)) * 8);
  }
  __syncthreads();

  const unsigned lds0 = (unsigned)(unsigned long long)(smem);
  const unsigned plane = (unsigned)(H * W) * 32u;        // bytes of one 16-channel plane of one image
  const int nbatch = my_tiles * 4;                       // chunk batches per group; the workgroup runs nbatch + WR_LAG steps
  const int nsteps = nbatch + WR_LAG;
  auto tile_coords = [&](int ord, int& b, int& oy0, int& ox0) {
    const int t = band0 + q + ord * nq;
    b = t / tiles;
    const int tile = t - b * tiles, ty = tile / tiles_x;
    oy0 = ty * (2 * WR_HT); ox0 = (tile - ty * tiles_x) * 32;
  };

  if (!consumer) {
    // ================================================================================================ producers
    const int pw = wave - WR_CONS;
    const int grp = pw >> 1;                  // the consumer group this producer feeds
    const int first = (pw & 1) ? 6 : 0;       // its pieces of the half-tile halo: 0-5 (six) or 6-10 (five)
    const bool six = !(pw & 1);
    const int lag = grp ? WR_LAG : 0;
    int d_iy[6], d_ix[6], d_rel[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int s = (first + j) * 64 + lane, p = s >> 1, half = (s & 1) ^ ((p >> 3) & 1);
      const int iy = p / WR_IW, ix = p - iy * WR_IW;
      d_iy[j] = (p < WR_NPIX && first + j < WR_PIECES) ? iy : 1 << 20;        // pad slots: never inside the image
      d_ix[j] = ix;
      d_rel[j] = (iy * W + ix) * 32 + half * 16;
    }
    i32x4 rsrc;
    {
      const unsigned long long p = reinterpret_cast<unsigned long long>(a.src);
      rsrc[0] = (int)(unsigned)p; rsrc[1] = (int)(unsigned)(p >> 32); rsrc[2] = (int)a.src_bytes; rsrc[3] = 0x00020000;
    }
    int iu = 0, ic = 0, gi = 0;
    unsigned voff[6], soff0 = 0;
    auto issue_batch = [&]() {
      if (ic == 0) {
        int b, oy0, ox0;
        tile_coords(iu, b, oy0, ox0);
        oy0 += grp * WR_HT;
        soff0 = (unsigned)b * 4u * plane;
        const int base = ((oy0 - 1) * W + (ox0 - 1)) * 32;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int gy = oy0 - 1 + d_iy[j], gx = ox0 - 1 + d_ix[j];
          const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
          voff[j] = ok ? (unsigned)(base + d_rel[j]) : 0x80000000u;     // out of range => the DMA writes zeros
        }
      }
      const unsigned soff = __builtin_amdgcn_readfirstlane(soff0 + (unsigned)ic * plane);
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((grp * WR_NS + gi % WR_NS) * WR_ACT + first * 1024));
      if (!(DBG & 2)) {
        if (six) wr_dma6(voff, rsrc, soff, dst);
        else wr_dma5(voff, rsrc, soff, dst);
      }
      ++gi;
      if (++ic == 4) { ic = 0; ++iu; }
    };
#pragma unroll
    for (int k = 0; k < WR_NS - 1; ++k)
      if (k < nbatch) issue_batch();
    for (int s = 0; s < nsteps; ++s) {
      const int g = s - lag;                   // the batch my group consumes in this step (outside [0, nbatch): none)
      // my pieces of batch g have landed: all but the newest (NS - 2) batches of mine (DMA pieces retire in issue order among
      // themselves; a producer issues nothing else)
      if (g >= 0 && g < nbatch) {
        if (g + WR_NS - 2 >= nbatch) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (six) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        static_assert(WR_NS - 2 == 2, "counted wait immediates");
      }
      __builtin_amdgcn_s_barrier();
      if (g >= 0 && g + WR_NS - 1 < nbatch) issue_batch();      // into the stage of batch g-1: nobody reads it behind this barrier
    }
    return;
  }

  // ================================================================================================== consumers
  const int grp = wave >> 2, cw = wave & 3;
  const int lag = grp ? WR_LAG : 0;
  int p_off[12];        // fragment offsets: halo rows 2cw + rr (rr = 0..3), column offset dx, this lane's pixel r
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int p = (cw * 2 + rr) * WR_IW + dx + r;
      p_off[rr * 3 + dx] = grp * WR_NS * WR_ACT + (2 * p + (h ^ ((p >> 3) & 1))) * 16;
    }
  const unsigned char* sWl = smem + WR_W_OFF + (h * 64 + r) * 16;
  const float slope = a.act == CDFO_ACT_NONE ? 1.f : (a.act == CDFO_ACT_LRELU ? 0.1f : 0.f);
  const int nck = a.Cout >> 4, hp = H >> 1;
  // bias of this lane's accumulator rows: MFMA row 8j + 4h + k of block ni = channel chan_of_row(ni*32 + 8j + 4h + k)
  f32x4 bias[2][4];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) bias[ni][j][k] = a.bias ? a.bias[n0 + chan_of_row(ni * 32 + 8 * j + 4 * h + k)] : 0.f;

  for (int s = 0; s < lag; ++s) {            // group B idles through its lag (the barriers are the workgroup's)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  int g = 0;
  for (int ord = 0; ord < my_tiles; ++ord) {
    int cb, oy0, cx0;
    tile_coords(ord, cb, oy0, cx0);
    const int cy0 = oy0 + grp * WR_HT + cw * 2;
    f32x16 acc[2][2];                            // [ni: 32-channel block][mi: image row]
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[ni][0][4 * j + k] = bias[ni][j][k]; acc[ni][1][4 * j + k] = bias[ni][j][k]; }
    // ---- epilogue pieces: act -> fp16 -> one 16-byte store.  acc[ni][mi][8jj + 4b + k] = channel ni*32 + jj*16 + h*8 + b*4 + k
    // of pixel r in image row cy0 + mi, i.e. halves [h*8, h*8+8) of chunk nb*4 + ni*2 + jj.
    _Float16* orow[2];         // row mi: address of (chunk plane nb*4, this lane's pixel, halves h*8..)
    bool ook[2];
    long long cstride;         // 16-half records between consecutive chunk planes
    {
      const int x = cx0 + r;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int y = cy0 + mi;
        ook[mi] = y < H && x < W;
        long long pos;
        if (a.s2d) {
          const int ph = (y & 1) * 2 + (x & 1);
          cstride = (long long)hp * (W >> 1);
          pos = ((long long)cb * 4 * nck + ph * nck + nb * 4) * cstride + (long long)(y >> 1) * (W >> 1) + (x >> 1);
        } else {
          cstride = (long long)H * W;
          pos = ((long long)cb * nck + nb * 4) * cstride + (long long)y * W + x;
        }
        orow[mi] = a.out + pos * 16 + h * 8;
      }
    }
    auto epilogue_piece = [&](int mi, int k) {       // k = ni*2 + jj
      const int ni = k >> 1, jj = k & 1;
      f16x8_t hv;
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) {
        const float v = acc[ni][mi][8 * jj + qq];
        hv[qq] = (_Float16)fmaxf(v, slope * v);      // slope in [0, 1]: identity / LeakyReLU / ReLU
      }
      if (ook[mi]) *reinterpret_cast<f16x8_t*>(orow[mi] + (long long)k * cstride * 16) = hv;
    };
#pragma unroll
    for (int c = 0; c < 4; ++c, ++g) {
      unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};      // dbg 32: step entry, barrier passed, after taps 2 / 5 / 8, step end
      if (DBG & 32) asm volatile("s_memtime %0" : "=s"(ts[0]) : : "memory");
      // my fragment reads of batch g-1 have returned (its stage is refilled behind the barrier)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // Issue priority.  A SIMD's two consumers do not share the matrix pipe evenly: the older wave (group A) takes its 36 MFMAs
      // first, the younger one (group B) finishes ~1 200 cycles later (dbg 32 timeline).  That hides A's epilogue behind B's
      // tail -- but B's own epilogue came after everything else in its step, with A already waiting at the barrier.  So a
      // consumer raises its priority for the LAST chunk of its tile: it finishes first and converts / stores while the other
      // group, half a tile away, computes.
      if (WR_PRIO) {
        if (c == 3) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
      }
      if (DBG & 32) asm volatile("s_memtime %0" : "=s"(ts[1]) : : "memory");
      const unsigned char* sA = smem + (g % WR_NS) * WR_ACT;
      const unsigned char* sWc = sWl + c * (9 * 2 * 64 * 16);
      f16x8_t fp[2][2], fw[2][2];                // [parity][mi / ni]: fragments are read one tap ahead
      {
        auto load_frags = [&](int t, int par) {
          const int dy = t / 3, dx = t - dy * 3;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) fw[par][ni] = *reinterpret_cast<const f16x8_t*>(sWc + (t * 2 * 64 + ni * 32) * 16);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) fp[par][mi] = *reinterpret_cast<const f16x8_t*>(sA + p_off[(mi + dy) * 3 + dx]);
        };
        load_frags(0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if (t < 8) load_frags(t + 1, (t & 1) ^ 1);
          __builtin_amdgcn_sched_barrier(0);
          const int par = t & 1;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
              if (DBG & 1) acc[ni][mi][0] += (float)fw[par][ni][0] * (float)fp[par][mi][0];
              else if (DBG & 4) {      // clock experiment: the 16x16x32 shape at equal FLOPs and fragment reads (results are NOT the convolution)
                f32x4 lo = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
                f32x4 hi = {acc[ni][mi][4], acc[ni][mi][5], acc[ni][mi][6], acc[ni][mi][7]};
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[par][ni], fp[par][mi], lo, 0, 0, 0);
                hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[par][ni], fp[par][mi], hi, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[ni][mi][e] = lo[e]; acc[ni][mi][4 + e] = hi[e]; }
              } else acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[par][ni], fp[par][mi], acc[ni][mi], 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
          if ((DBG & 32) && (t == 2 || t == 5 || t == 8)) asm volatile("s_memtime %0" : "=s"(ts[2 + t / 3]) : : "memory");
        }
      }
      if (DBG & 32) {
        asm volatile("s_memtime %0" : "=s"(ts[5]) : : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (a.clk && ord == 2 && lane == 0) {
          unsigned long long* cbuf = a.clk + (((long long)blockIdx.x * 12 + wave) * 4 + c) * 8;
#pragma unroll
          for (int i = 0; i < 6; ++i) cbuf[i] = ts[i];
        }
      }
    }
    if (DBG & 32) {
      unsigned long long te = 0;
      asm volatile("s_memtime %0" : "=s"(te) : : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (a.clk && (ord == 3 || ord == 4) && lane == 0) a.clk[(((long long)blockIdx.x * 12 + wave) * 4 + (ord - 3)) * 8 + 6] = te;   // tile period, undisturbed
    }
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
    // (tried: the last chunk row by row with row 0's epilogue pieces between row 1's taps -- tile period 12 870 -> 13 440 cycles, removed)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int k = 0; k < 4; ++k) epilogue_piece(mi, k);
  }
  for (int s = lag; s < WR_LAG; ++s) {       // group A waits out group B's lag: every wave runs nbatch + WR_LAG barriers
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same ring-fed, wave-specialised kernel on v_mfma_f32_16x16x32_f16 (round 3).
// Why another MFMA shape: every structure tried for this convolution -- private halos with three drifting waves per SIMD, the
// shared-halo ring above in lock step, with lagged consumer groups, with the epilogue inside the last chunk -- delivers the
// same 1.04-1.07 PFLOP/s, because the loop is POWER-limited: the chip holds 1.5-1.65 GHz under it and gives an in-kernel
// cycle saving back as a lower clock (MI355X_MICROARCH.md, DVFS give-back).  What moves the number is energy per FLOP, and the
// guide measures the 16x16x32 shape at ~1.13x the FLOP/s of 32x32x16 at equal cycles on random data; swapping the shape in the
// kernel above at equal FLOPs and fragment reads (dbg 4, wrong arithmetic) ran 8-13 % faster.  So:
//   * K = 32 per MFMA = both 16-channel planes of a 32-channel SUPERCHUNK at one tap: a step stages two planes per group
//     (2 x 11 KiB), a tile is two steps, the rings are two stages deep (same bytes in flight as 4 x 11 KiB), group B runs one
//     step = half a tile behind group A;
//   * a consumer's 2 rows x 32 pixels x 64 channels are 4 x 4 accumulators of 16 x 16: per tap 4 weight fragments + 4 pixel
//     fragments for 16 MFMAs (the same LDS bytes per FLOP as before); lane l holds k-group l >> 4 = (plane, 8-channel half) of
//     pixel / output channel l & 15, which needs NO bank swizzle for ds_read_b128 with natural [pixel][16 ch] records and the
//     weight image in natural channel order;
//   * accumulator registers are 4 consecutive output channels of one pixel: the epilogue stores 8 bytes per lane and 16 x 16
//     block, 512 contiguous bytes per wave-instruction.
constexpr int WQ_PLANE = WR_PIECES * 1024, WQ_ACT = 2 * WQ_PLANE, WQ_NS = 2, WQ_LAG = 1;
constexpr int WQ_W_OFF = 2 * WQ_NS * WQ_ACT;           // 90,112
constexpr int WQ_TOTAL = WQ_W_OFF + WS_W_BYTES;        // 163,840 bytes
static_assert(WQ_TOTAL <= 160 * 1024, "LDS budget");

// OFFMASK: the convolution is MVDualAttAlignment's conv_offset[2] (a.om_cout = 27 dg real output channels inside a.Cout padded ones)
// and the epilogue is the module's offset / mask assembly into the DCN operator's NCHW planes (arch.py:3336-3350), like
// conv3x3_bf16.hip's CDFO_STORE_OFFMASK: a lane owns 16 output channels of 4 pixels whose 16 lane-neighbours are consecutive along x,
// so every access is a 4-byte element of a 64-byte run of one plane.  All of a tile's prior values (the motion field for the
// first head, the planes themselves for the second) are requested before the first store.
template <int DBG, bool OFFMASK = false>
__global__ __launch_bounds__(WR_THREADS) void conv3x3_c64_wsq_kernel(ws_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave < WR_CONS;
  const int H = a.H, W = a.W;
  const int nco = a.Cout >> 6;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int nq = nslots / nco, nb = slot % nco, q = slot / nco;
  const int tiles_x = (W + 31) >> 5, tiles_y = (H + 2 * WR_HT - 1) / (2 * WR_HT), tiles = tiles_x * tiles_y;
  const int all_tiles = a.B * tiles;
  const int band = (all_tiles + 7) >> 3, band0 = xcd * band;
  const int band_n = min(band, all_tiles - band0);
  const int my_tiles = (q < nq && band_n > q) ? (band_n - q + nq - 1) / nq : 0;
  if (my_tiles == 0) return;
  const int n0 = nb * 64;

  // ---- one-time: this workgroup's weight block: row = (chunk*9 + tap)*2 + k-half, 64 x 16 bytes.  MFMA row i of 16-row block mb
  // holds output channel (mb>>1)*32 + (i>>2)*8 + (mb&1)*4 + (i&3): a lane's accumulators of blocks 2e and 2e+1 (rows 4 kg ..
  // 4 kg + 3 each) are then 8 CONSECUTIVE channels e*32 + 8 kg .. + 7 -- one 16-byte store per lane, pixel block and block pair
  auto chan_q = [](int n) { return (n >> 5) * 32 + ((n & 15) >> 2) * 8 + ((n >> 4) & 1) * 4 + (n & 3); };
  for (int i = tid; i < WS_W_BYTES / 16; i += WR_THREADS) {
    const int row = i >> 6, n = i & 63;
    *reinterpret_cast<u32x4*>(smem + WQ_W_OFF + i * 16) =
        *reinterpret_cast<const u32x4*>(a.w + ((long long)row * a.CoutP + n0 + chan_q(n)) * 8);
  }
  __syncthreads();

  const unsigned lds0 = (unsigned)(unsigned long long)(smem);
  const unsigned plane = (unsigned)(H * W) * 32u;        // bytes of one 16-channel plane of one image
  const int nbatch = my_tiles * 2;                       // superchunk batches per group; the workgroup runs nbatch + WQ_LAG steps
  const int nsteps = nbatch + WQ_LAG;
  auto tile_coords = [&](int ord, int& b, int& oy0, int& ox0) {
    const int t = band0 + q + ord * nq;
    b = t / tiles;
    const int tile = t - b * tiles, ty = tile / tiles_x;
    oy0 = ty * (2 * WR_HT); ox0 = (tile - ty * tiles_x) * 32;
  };

  if (!consumer) {
    // ================================================================================================ producers
    const int pw = wave - WR_CONS;
    const int grp = pw >> 1;
    const int first = (pw & 1) ? 6 : 0;       // pieces 0-5 (six) or 6-10 (five) of BOTH planes of the half-tile halo
    const bool six = !(pw & 1);
    const int lag = grp ? WQ_LAG : 0;
    int d_iy[6], d_ix[6], d_rel[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int s = (first + j) * 64 + lane, p = s >> 1, half = s & 1;      // natural records: LDS byte 16 s = 32 p + 16 half
      const int iy = p / WR_IW, ix = p - iy * WR_IW;
      d_iy[j] = (p < WR_NPIX && first + j < WR_PIECES) ? iy : 1 << 20;
      d_ix[j] = ix;
      d_rel[j] = (iy * W + ix) * 32 + half * 16;
    }
    i32x4 rsrc;
    {
      const unsigned long long p = reinterpret_cast<unsigned long long>(a.src);
      rsrc[0] = (int)(unsigned)p; rsrc[1] = (int)(unsigned)(p >> 32); rsrc[2] = (int)a.src_bytes; rsrc[3] = 0x00020000;
    }
    int iu = 0, ic = 0, gi = 0;
    unsigned voff[6], soff0 = 0;
    auto issue_batch = [&]() {
      if (ic == 0) {
        int b, oy0, ox0;
        tile_coords(iu, b, oy0, ox0);
        oy0 += grp * WR_HT;
        soff0 = (unsigned)b * 4u * plane;
        const int base = ((oy0 - 1) * W + (ox0 - 1)) * 32;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int gy = oy0 - 1 + d_iy[j], gx = ox0 - 1 + d_ix[j];
          const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
          voff[j] = ok ? (unsigned)(base + d_rel[j]) : 0x80000000u;
        }
      }
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((grp * WQ_NS + gi % WQ_NS) * WQ_ACT + first * 1024));
      if (!(DBG & 2)) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const unsigned soff = __builtin_amdgcn_readfirstlane(soff0 + (unsigned)(ic * 2 + pl) * plane);
          if (six) wr_dma6(voff, rsrc, soff, dst + pl * WQ_PLANE);
          else wr_dma5(voff, rsrc, soff, dst + pl * WQ_PLANE);
        }
      }
      ++gi;
      if (++ic == 2) { ic = 0; ++iu; }
    };
    if (nbatch > 0) issue_batch();
    for (int s = 0; s < nsteps; ++s) {
      const int g = s - lag;
      // two stages: batch g is the only thing of mine in flight when I wait for it
      if (g >= 0 && g < nbatch) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (g >= 0 && g + 1 < nbatch) issue_batch();      // into the stage of batch g-1: nobody reads it behind this barrier
    }
    return;
  }

  // ================================================================================================== consumers
  const int grp = wave >> 2, cw = wave & 3;
  const int lag = grp ? WQ_LAG : 0;
  const int kplane = kg >> 1, khalf = kg & 1;
  int p_off[12];        // pixel fragments: halo rows 2cw + rr (rr = 0..3), column offset dx, pixel l16 (+16: immediate), this lane's k-group
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
      p_off[rr * 3 + dx] = grp * WQ_NS * WQ_ACT + kplane * WQ_PLANE + ((cw * 2 + rr) * WR_IW + dx + l16) * 32 + khalf * 16;
  // weight fragments: row (superchunk*2 + kplane)*9*2 + tap*2 + khalf, output channel mb*16 + l16
  const unsigned char* sWl = smem + WQ_W_OFF + (kplane * 9 * 2 + khalf) * 1024 + l16 * 16;
  const float slope = a.act == CDFO_ACT_NONE ? 1.f : (a.act == CDFO_ACT_LRELU ? 0.1f : 0.f);
  const int nck = a.Cout >> 4, hp = H >> 1;
  f32x4 bias[4];        // accumulator rows 4 kg .. 4 kg + 3 of block mb
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int n = n0 + chan_q(mb * 16 + 4 * kg + k);
      bias[mb][k] = (a.bias && (!OFFMASK || n < a.om_cout)) ? a.bias[n] : 0.f;
    }

  for (int s = 0; s < lag; ++s) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  int g = 0;
  for (int ord = 0; ord < my_tiles; ++ord) {
    int cb, oy0, cx0;
    tile_coords(ord, cb, oy0, cx0);
    const int cy0 = oy0 + grp * WR_HT + cw * 2;
    f32x4 acc[4][4];                             // [mb: 16 output channels][nbk = row*2 + pixel half]
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nbk = 0; nbk < 4; ++nbk) acc[mb][nbk] = bias[mb];
#pragma unroll
    for (int c = 0; c < 2; ++c, ++g) {
      unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};      // dbg 32: step entry, barrier passed, after taps 2 / 5 / 8, step end
      if (DBG & 32) asm volatile("s_memtime %0" : "=s"(ts[0]) : : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my fragment reads of batch g-1 have returned
      __builtin_amdgcn_s_barrier();
      if (DBG & 32) asm volatile("s_memtime %0" : "=s"(ts[1]) : : "memory");
      if (WR_PRIO) {      // last step of my tile: finish first, convert / store while the other group computes (see the kernel above)
        if (c == 1) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
      }
      const unsigned char* sA = smem + (g % WQ_NS) * WQ_ACT;
      const unsigned char* sWc = sWl + c * (2 * 9 * 2 * 1024);
      f16x8_t fp[2][4], fw[2][4];                // [parity][pixel block / channel block]: fragments are read one tap ahead
      auto load_frags = [&](int t, int par) {
        const int dy = t / 3, dx = t - dy * 3;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) fw[par][mb] = *reinterpret_cast<const f16x8_t*>(sWc + t * 2048 + mb * 256);
#pragma unroll
        for (int nbk = 0; nbk < 4; ++nbk)
          fp[par][nbk] = *reinterpret_cast<const f16x8_t*>(sA + p_off[((nbk >> 1) + dy) * 3 + dx] + (nbk & 1) * 512);
      };
      load_frags(0, 0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t < 8) load_frags(t + 1, (t & 1) ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const int par = t & 1;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int nbk = 0; nbk < 4; ++nbk) {
            if (DBG & 1) acc[mb][nbk][0] += (float)fw[par][mb][0] * (float)fp[par][nbk][0];
            else acc[mb][nbk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[par][mb], fp[par][nbk], acc[mb][nbk], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
        if ((DBG & 32) && (t == 2 || t == 5 || t == 8)) asm volatile("s_memtime %0" : "=s"(ts[2 + t / 3]) : : "memory");
      }
      if (DBG & 32) {
        asm volatile("s_memtime %0" : "=s"(ts[5]) : : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (a.clk && ord == 2 && lane == 0) {
          unsigned long long* cbuf = a.clk + (((long long)blockIdx.x * 12 + wave) * 4 + c) * 8;
#pragma unroll
          for (int i = 0; i < 6; ++i) cbuf[i] = ts[i];
        }
      }
    }
    if (DBG & 32) {
      unsigned long long te = 0;
      asm volatile("s_memtime %0" : "=s"(te) : : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (a.clk && (ord == 3 || ord == 4 || ord == 24) && lane == 0)       // epilogue-start times: undisturbed tile period(s)
        a.clk[(((long long)blockIdx.x * 12 + wave) * 4 + (ord == 24 ? 2 : ord - 3)) * 8 + 6] = te;
    }
    if (DBG & 8) {
      float t = 0.f;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nbk = 0; nbk < 4; ++nbk)
#pragma unroll
          for (int e = 0; e < 4; ++e) t += acc[mb][nbk][e];
      if (t == 123.456f) a.out[0] = (_Float16)t;
      continue;
    }
    if (OFFMASK) {
      // acc[mb][nbk][k] (mb = 2e + hf) = channel n0 + e*32 + 8 kg + 4 hf + k of pixel (cy0 + (nbk >> 1), cx0 + 16 (nbk & 1) + l16).
      // 2 * third and om_cout are multiples of 32 resp. 8 (launcher), so a 32-channel half-block e is offset or mask as a whole and
      // a lane's eight channels of it are inside the real output or outside together.  Addresses: one buffer descriptor per plane
      // set and image, the per-lane part (8 kg planes + the pixel) in a 32-bit lane offset that carries the out-of-range marker,
      // the wave-uniform channel in the scalar offset -- 8 address registers for the 64 elements of a lane.
      const int third = a.om_cout / 3;
      const unsigned P4 = (unsigned)(H * W) * 4u;
      const __amdgpu_buffer_rsrc_t r_off = __builtin_amdgcn_make_buffer_rsrc(a.om_offset + (long long)cb * 2 * third * (H * W), 0,
                                                                             (int)(2 * third * P4), 0x00020000);
      const __amdgpu_buffer_rsrc_t r_msk = __builtin_amdgcn_make_buffer_rsrc(a.om_mask + (long long)cb * third * (H * W), 0,
                                                                             (int)(third * P4), 0x00020000);
      bool ok[4];
      unsigned vo[2][4];
#pragma unroll
      for (int nbk = 0; nbk < 4; ++nbk) {
        const int y = cy0 + (nbk >> 1), x = cx0 + (nbk & 1) * 16 + l16;
        ok[nbk] = y < H && x < W;
#pragma unroll
        for (int e = 0; e < 2; ++e)
          vo[e][nbk] = (ok[nbk] && n0 + e * 32 + 8 * kg < a.om_cout) ? (unsigned)(8 * kg) * P4 + (unsigned)(y * W + x) * 4u : 0x80000000u;
      }
      bool e_off[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) e_off[e] = n0 + e * 32 < 2 * third;
      auto soff_of = [&](int e, int hf, int k) {      // wave-uniform byte offset of channel n0 + e*32 + 4 hf + k inside its plane set
        return (unsigned)(n0 + e * 32 + 4 * hf + k - (e_off[e] ? 0 : 2 * third)) * P4;
      };
      // the accumulators as plain scalars (they are dead after this; element-wise updates of the vector registers in place made
      // hipcc store element 0 for every k), and the offset channels' own term first: vector work that needs nothing from memory
      float val[4][4][4];                      // [mb][nbk][k]
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nbk = 0; nbk < 4; ++nbk) {
          const f32x4 t = acc[mb][nbk];
          val[mb][nbk][0] = t[0]; val[mb][nbk][1] = t[1]; val[mb][nbk][2] = t[2]; val[mb][nbk][3] = t[3];
        }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        if (!e_off[mb >> 1]) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int nbk = 0; nbk < 4; ++nbk)
            val[mb][nbk][k] = a.om_mag * (1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(val[mb][nbk][k] * 2.885390081777927f)));
      }
      auto store_batch = [&](int e) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int nbk = 0; nbk < 4; ++nbk)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val[2 * e + hf][nbk][k]), e_off[e] ? r_off : r_msk,
                                                    (int)vo[e][nbk], (int)soff_of(e, hf, k), 0);
      };
      if (!a.om_accumulate) {
        // first head: offset = term + flipped motion field (the (y, x) pairs take (flow_y, flow_x): channel parity = k & 1), mask = raw sums
        float fl[2][4];
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
          for (int nbk = 0; nbk < 4; ++nbk) {
            const int y = cy0 + (nbk >> 1), x = cx0 + (nbk & 1) * 16 + l16;
            fl[par][nbk] = ok[nbk] ? a.om_flow[(long long)cb * a.om_flow_bstride + (long long)(1 - par) * (H * W) + y * W + x] : 0.f;
          }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          if (!e_off[mb >> 1]) continue;
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int nbk = 0; nbk < 4; ++nbk) val[mb][nbk][k] += fl[k & 1][nbk];
        }
        store_batch(0);
        store_batch(1);
        continue;
      }
      // second head, in place: two batches of 32 prior values; batch 1 is requested before batch 0 is stored (the stores may alias
      // the loads as far as the compiler knows, so a load -> store loop would pay one memory latency per run)
      float prior[2][4][4];                    // [hf][nbk][k]
      auto load_batch = [&](int e) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int nbk = 0; nbk < 4; ++nbk)
              prior[hf][nbk][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(e_off[e] ? r_off : r_msk, (int)vo[e][nbk],
                                                                                                (int)soff_of(e, hf, k), 0));
      };
      auto finish_batch = [&](int e) {          // val <- final values
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int nbk = 0; nbk < 4; ++nbk) {
              const float t = prior[hf][nbk][k] + val[2 * e + hf][nbk][k];
              val[2 * e + hf][nbk][k] = e_off[e] ? t : __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f));
            }
      };
      load_batch(0);
      finish_batch(0);
      load_batch(1);
      store_batch(0);
      finish_batch(1);
      store_batch(1);
      continue;
    }
    // ---- epilogue: act -> fp16 -> 16-byte stores.  acc[2e][nbk][k], acc[2e+1][nbk][k] = channels e*32 + 8 kg + k, + 4 + k of
    // pixel (row cy0 + (nbk >> 1), column cx0 + 16 (nbk & 1) + l16): halves [8 (kg & 1), + 8) of chunk nb*4 + 2e + (kg >> 1).
#pragma unroll
    for (int nbk = 0; nbk < 4; ++nbk) {
      const int y = cy0 + (nbk >> 1), x = cx0 + (nbk & 1) * 16 + l16;
      const bool ok = y < H && x < W;
      long long pos, cstride;
      if (a.s2d) {
        const int ph = (y & 1) * 2 + (x & 1);
        cstride = (long long)hp * (W >> 1);
        pos = ((long long)cb * 4 * nck + ph * nck + nb * 4) * cstride + (long long)(y >> 1) * (W >> 1) + (x >> 1);
      } else {
        cstride = (long long)H * W;
        pos = ((long long)cb * nck + nb * 4) * cstride + (long long)y * W + x;
      }
      _Float16* o = a.out + (pos + (kg >> 1) * cstride) * 16 + (kg & 1) * 8;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        f16x8_t hv;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v0 = acc[2 * e][nbk][k], v1 = acc[2 * e + 1][nbk][k];
          hv[k] = (_Float16)fmaxf(v0, slope * v0);
          hv[4 + k] = (_Float16)fmaxf(v1, slope * v1);
        }
        if (ok) *reinterpret_cast<f16x8_t*>(o + (long long)(2 * e) * cstride * 16) = hv;
      }
    }
  }
  for (int s = lag; s < WQ_LAG; ++s) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

int ws_num_cus() { return cdfo_num_cus(); }

// CDFO_WS_RING=0 selects the private-halo form above for the non-residual calls; default: the ring-fed wave-specialised form
bool ws_ring_form() {
  static const bool v = [] { const char* s = getenv("CDFO_WS_RING"); return !(s && s[0] == '0'); }();
  return v;
}

// CDFO_WS_MFMA16=0 keeps the 32x32x16 ring-fed form; default: the 16x16x32 one
bool ws_mfma16() {
  static const bool v = [] { const char* s = getenv("CDFO_WS_MFMA16"); return !(s && s[0] == '0'); }();
  return v;
}

template <int DBG>
int wsr_launch(const ws_args& a, hipStream_t st) {
  if constexpr ((DBG & ~43) == 0) {
    if (ws_mfma16()) {
      static CdfoAttrOnce onceq;
      const hipError_t e = cdfo_set_max_lds(onceq, reinterpret_cast<const void*>(conv3x3_c64_wsq_kernel<DBG>), WQ_TOTAL);
      if (e != hipSuccess) return (int)e;
      const int grid = ws_num_cus() / 8 * 8;
      hipLaunchKernelGGL((conv3x3_c64_wsq_kernel<DBG>), dim3(grid), dim3(WR_THREADS), WQ_TOTAL, st, a);
      return 0;
    }
  }
  static CdfoAttrOnce once;
  const hipError_t e = cdfo_set_max_lds(once, reinterpret_cast<const void*>(conv3x3_c64_wsr_kernel<DBG>), WR_TOTAL);
  if (e != hipSuccess) return (int)e;
  const int grid = ws_num_cus() / 8 * 8;
  hipLaunchKernelGGL((conv3x3_c64_wsr_kernel<DBG>), dim3(grid), dim3(WR_THREADS), WR_TOTAL, st, a);
  return 0;
}

template <int DBG, bool RES = false>
int ws_launch(const ws_args& a, int grid, hipStream_t st) {
  if constexpr (!RES && (DBG & ~47) == 0) {        // ablation bits 1, 2, 8 exist in both forms; 32 = this form's timeline probe
    if (ws_ring_form() && (a.Cout >> 6) <= (ws_num_cus() / 8)) return wsr_launch<DBG>(a, st);
  }
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
    case 4: rc = ws_launch<4>(a, grid, st); break;
    case 12: rc = ws_launch<12>(a, grid, st); break;
    case 8: rc = ws_launch<8>(a, grid, st); break;
    case 9: rc = ws_launch<9>(a, grid, st); break;
    case 11: rc = ws_launch<11>(a, grid, st); break;
    case 16: rc = ws_launch<16>(a, grid, st); break;
    case 32: rc = ws_launch<32>(a, grid, st); break;
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

extern "C" int cdfo_conv3x3_c64_ws_offmask(const void* src_cp16, int B, int H, int W, const void* w_f16, int CoutP, const float* bias,
                                           int Cout, float* offset, float* mask, const float* flow, long long flow_bstride,
                                           float mag, int accumulate, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || Cout <= 0 || Cout % 3 || Cout % 8 || (2 * Cout / 3) % 32 || CoutP < Cout || CoutP % 64 || !offset ||
      !mask || !flow || (long long)H * W * (2 * Cout / 3) * 4 >= (1ll << 31))
    return CDFO_EINVAL;      // (32-bit plane-set offsets; a half-block of 32 channels is offset or mask as a whole)
  const long long src_bytes = (long long)B * 4 * H * W * 32;
  if (src_bytes >= (1ll << 31)) return CDFO_EINVAL;      // 32-bit buffer offsets, out-of-range marker 0x80000000
  if (!aligned16(src_cp16) || !aligned16(w_f16)) return CDFO_EALIGN;
  const int cus = ws_num_cus();
  if (cus < 8 || (CoutP >> 6) > cus / 8) return CDFO_EINVAL;
  ws_args a{};
  a.src = src_cp16; a.src_bytes = (unsigned)src_bytes;
  a.B = B; a.H = H; a.W = W;
  a.w = static_cast<const unsigned short*>(w_f16); a.CoutP = CoutP;
  a.bias = bias; a.Cout = CoutP; a.act = CDFO_ACT_NONE;      // the kernel walks the padded 64-channel blocks; om_cout bounds the real ones
  a.om_offset = offset; a.om_mask = mask; a.om_flow = flow; a.om_flow_bstride = flow_bstride; a.om_mag = mag;
  a.om_accumulate = accumulate; a.om_cout = Cout;
  const double px = (double)B * H * W;
  CdfoProfScope prof(st, KID_CONV3_WS, 2.0 * px * Cout * 64 * 9, (accumulate ? 8.0 : 4.0) * px * Cout + 2.0 * px * 64 + 2.0 * 9 * 64 * Cout);
  static CdfoAttrOnce once;
  const hipError_t e = cdfo_set_max_lds(once, reinterpret_cast<const void*>(conv3x3_c64_wsq_kernel<0, true>), WQ_TOTAL);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((conv3x3_c64_wsq_kernel<0, true>), dim3(cus / 8 * 8), dim3(WR_THREADS), WQ_TOTAL, st, a);
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
