// 3x3 stride-1 convolution of an fp16 chunk-planar source, many input channels -> 64-wide output blocks, as a persistent
// kernel fed by an LDS-DMA ring (fp16 MFMA, fp32 accumulate).
//
// Block_.body[2] (arch/SIDECVSR_our.py:383-387: Conv2d(256, 64, 3, 1, 1)) and the composed stride-2 convolution of the
// block's double-resolution branch (1024 space-to-depth channels, 4 of 9 taps per chunk) have too many input channels
// for weights-stationary LDS (295 / 524 KB per 64 outputs), so both operands stream:
//   * one 512-thread workgroup per CU walks over 16-row x 32-pixel output tiles; per 16-channel chunk the 18 x 34 pixel
//     halo (20 pieces of 1 KiB) and the chunk's weight slab (18 pieces; 8 when only four taps carry weights) are
//     copied global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), four or five instructions per wave and chunk,
//     issued one at a time BETWEEN the taps' MFMAs -- no staging registers, no ds_write pass;
//   * a ring of four stages (dense form) or three (four-tap form), i.e. three or two chunks in flight, ONE workgroup
//     barrier per chunk (the tiled kernel needs two and restages through VGPRs); the ring runs across tile
//     boundaries, so the epilogue of one tile overlaps the loads of the next;
//   * four-tap form: the half-resolution residual of Block_'s x1/2 branch (added after bilinear x2 by the epilogue) is
//     copied into LDS by DMA at the tile's first chunk, so its taps cost no global loads at the tile boundary;
//   * optional source plane wrap (chunk c reads plane c % wrap): a split-fp16, fp32-grade product
//     (a_hi | a_lo | a_hi) x (w_hi | w_hi | w_lo) runs as one K-expanded convolution without duplicated planes;
//   * chunk-planar source [B][Cin/16][H][W][16]: a halo row is 34 x 32 contiguous bytes; out-of-image pixels get an
//     out-of-range buffer offset = hardware zero fill = the zero padding;
//   * the dense 32-byte pixel records are made conflict-free for ds_read_b128 by swapping the two 16-byte halves of
//     every second group of 8 pixels (on the DMA source side and on the read side);
//   * transposed product (M = output channels, weight rows permuted by the DMA source address) so that a lane's
//     accumulators are 8 consecutive channels of one pixel: the epilogue (bias, activation, residuals, bilinear x2 of a
//     half-resolution residual, fp32 / fp16 pixel-major store, optional fp16 chunk-planar copy) works straight from
//     registers, 32 contiguous bytes per lane -- no LDS transpose, no extra barrier; the residual values are fetched
//     while the tile's last chunk computes.
// DMA completion is hand-counted (hipcc does not see inline-asm memory operations); the rules sit next to each wait.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int RG_THREADS = 512;
constexpr int RG_TH = 16, RG_IW = 34, RG_NPIX = 18 * 34;   // 612 staged pixels per chunk
constexpr int RG_ACT = 20 * 1024;                          // 20 DMA pieces (612 x 32 B = 19,584 B + pad slots)
constexpr int RG_ET_ROWS = 10, RG_ET_COLS = 18;            // half-resolution residual region of a 16 x 32 tile (+ taps)
constexpr int RG_ET_PIECES = 48;                           // 180 pixels x 256 B = 45 KiB, 6 DMA pieces per wave
// LDS map: NS ring stages (dense: 4 x 38 KB; four-tap form: 3 x 28 KB) | [four-tap form: 48 KB half-resolution residual
// tile] | 1 KB dump | 4 KB bias
template <bool SPARSE> struct RingLds {
  static constexpr int NS = SPARSE ? 3 : 4;                                       // NS - 1 chunk batches in flight
  static constexpr int WGT = (SPARSE ? 8 : 18) * 1024;
  static constexpr int STAGE = RG_ACT + WGT;
  static constexpr int ETILE = NS * STAGE;
  static constexpr int DUMP = ETILE + (SPARSE ? RG_ET_PIECES * 1024 : 0);         // target of the padding DMA pieces
  static constexpr int BIAS = DUMP + 1024;
  static constexpr int WINTAB = BIAS + 4096;                                      // four-tap form: window origin per chunk (bytes)
  static constexpr int TOTAL = WINTAB + (SPARSE ? 1024 : 0);
};

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ring_extra {
  const void* src;            // fp16 chunk-planar [B][nc][H][W][16]
  int nc;                     // 16-channel chunks = Cin / 16
  const void* w;              // fp16 [nc][taps][2][CoutP][8], taps = 9 (dense) or 4 (the chunk's active taps, ascending)
  int CoutP;
  const unsigned* tap_mask;   // nullptr = dense
  int w_bytes;                // size of the weight buffer
  int plane_wrap;             // chunk c reads source plane c % plane_wrap (0: plane c)
  int src_planes;             // planes per image in the source tensor
  int touch;                  // wave-specialised form: producers touch the tile's res1 lines ahead of the epilogue
  int halfsplit;              // source rows are [8-channel half][W][8] (cdfo_conv_args.src_halfsplit): the two 16-byte halves of a pixel's
                              // record lie W * 16 bytes apart instead of side by side
};
__device__ __forceinline__ bool rg_touch_on(const ring_extra& e) { return e.touch != 0; }

// one 1 KiB LDS-DMA piece: lane l writes LDS bytes lds + 16 l from (buffer base + voff + soff).  rsrc / soff / lds must
// be wave-uniform values the compiler can keep in SGPRs (readfirstlane them).
__device__ __forceinline__ void rg_dma1(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds)
      : "memory");
}

__device__ __forceinline__ i32x4 rg_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  i32x4 r;
  r[0] = (int)(unsigned)p;
  r[1] = (int)(unsigned)(p >> 32);
  r[2] = (int)bytes;
  r[3] = 0x00020000;
  return r;
}

constexpr int RG_PROBE_UNIT = 3;      // dbg 16: the tile (ordinal within the workgroup) whose chunks are stamped
template <int V> struct IntC { static constexpr int value = V; };

template <int N> __device__ __forceinline__ void rg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void rg_wait_vm_lgkm0() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// DBG (developer ablations, tools/bench_ring.py): 1 = skip the MFMAs, 2 = skip the DMA, 4 = MFMA-shape clock experiment (wrong
// arithmetic), 8 = skip the epilogue, 16 = timeline
// probe: a.res2 is NOT a residual but a u64 buffer [workgroup][wave][4 chunks][10 stamps] that receives s_memtime stamps of
// chunks 8..11 of each workgroup's fourth tile (entry, after the counted wait, after the barrier, before the first tap, after
// each of four tap groups, chunk end)
template <bool SPARSE, int DBG>
__global__ __launch_bounds__(RG_THREADS) void conv3x3_ring_kernel(cdfo_conv_args a, ring_extra e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TAPS = SPARSE ? 4 : 9;
  using L = RingLds<SPARSE>;
  constexpr int STAGE = L::STAGE, RG_NS = L::NS;
  // DMA pieces per wave and chunk.  dense: waves 0-3 copy the 20 activation pieces, waves 4-7 the 18 weight pieces
  // (+2 padding pieces into the dump kilobyte, so that every loader issues exactly PPW instructions per chunk);
  // four-tap form: waves 0-4 the activations, waves 5-6 the 8 weight pieces, wave 7 none.
  constexpr int PPW = SPARSE ? 4 : 5, ACT_WAVES = 20 / PPW;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader_w = wave >= ACT_WAVES;
  const bool loader = !loader_w || (wave - ACT_WAVES) * PPW < TAPS * 2;
  const int H = a.H, W = a.W, nc = e.nc;
  const int tiles_x = (W + 31) >> 5, tiles_y = (H + RG_TH - 1) / RG_TH, tiles = tiles_x * tiles_y;
  const int nco = a.CoutP >> 6;
  const int units = a.B * tiles * nco;             // unit = (image, tile, 64-channel output block), block fastest
  // each XCD (private L2) takes a contiguous band of units; its workgroups stride through the band
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int band = (units + 7) >> 3, band0 = xcd * band;
  const int band_n = min(band, units - band0);     // may be <= 0
  const int my_units = band_n > slot ? (band_n - slot + nslots - 1) / nslots : 0;
  if (my_units == 0) return;
  const unsigned lds0 = (unsigned)(unsigned long long)(smem);
  for (int i = tid; i < a.CoutP; i += RG_THREADS)
    reinterpret_cast<float*>(smem + L::BIAS)[i] = (a.bias && i < a.Cout) ? a.bias[i] : 0.f;
  // (made visible to the other waves by the barriers of the chunk loop, long before the first epilogue)
  // four-tap form: the chunk's four taps are a 2x2 window of the 3x3 stencil (checked by the host wrapper); its origin
  // (2 y0 + x0) per chunk goes into an LDS table ONCE.  Reading e.tap_mask[c] inside the chunk loop (rounds 2-3) put a
  // vector global load and the compiler's s_waitcnt vmcnt(0) -- which also drains every LDS-DMA piece in flight -- between
  // the barrier and the chunk's first MFMA: one exposed memory latency per chunk in every wave at once.
  auto win_of = [](unsigned tm) { return ((tm & 0x7u) ? 0 : 2) + ((tm & 0x49u) ? 0 : 1); };
  if (SPARSE)
    for (int i = tid; i < nc; i += RG_THREADS) smem[L::WINTAB + i] = (unsigned char)win_of(e.tap_mask[i]);

  // ---- per-lane DMA descriptors
  // activation piece q = PPW*wave + j: slot s = 64 q + lane -> pixel p = s >> 1 of the 18 x 34 halo, k-half
  // (s & 1) ^ ((p >> 3) & 1);  weight piece q = slab row (tap*2 + k-half), lane = output channel of the block
  int d_iy[PPW], d_ix[PPW], d_rel[PPW], dst_off[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    if (!loader_w) {
      const int q = wave * PPW + j, s = q * 64 + lane, p = s >> 1, half = (s & 1) ^ ((p >> 3) & 1);
      const int iy = p / RG_IW, ix = p - iy * RG_IW;
      d_iy[j] = p < RG_NPIX ? iy : 1 << 20;
      d_ix[j] = ix;
      d_rel[j] = e.halfsplit ? ((iy * 2 + half) * W + ix) * 16 : (iy * W + ix) * 32 + half * 16;
      dst_off[j] = q * 1024;
    } else {
      const int q = (wave - ACT_WAVES) * PPW + j;
      d_iy[j] = q < TAPS * 2 ? 0 : 1 << 20;
      d_ix[j] = 0;
      // LDS position `lane` of a slab row = MFMA row m of 32-channel block (lane >> 5); it holds output channel
      // (m>>4)*16 + ((m>>2)&1)*8 + ((m>>3)&1)*4 + (m&3), so that the accumulator registers of a lane are 8 consecutive
      // channels of each 16-channel chunk (see the epilogue)
      const int m = lane & 31;
      const int chan = (lane & 32) + ((m >> 4) & 1) * 16 + ((m >> 2) & 1) * 8 + ((m >> 3) & 1) * 4 + (m & 3);
      d_rel[j] = (q * e.CoutP + chan) * 16;
      dst_off[j] = q < TAPS * 2 ? RG_ACT + q * 1024 : -1;      // padding piece -> dump
    }
  }
  const unsigned plane = (unsigned)(H * W) * 32u;                 // bytes of one 16-channel plane
  const unsigned wchunk = (unsigned)(TAPS * 2 * e.CoutP) * 16u;   // bytes of one chunk's weight slab (all output blocks)
  const unsigned img_bytes = plane * (unsigned)e.src_planes;
  const i32x4 rsrc_w = rg_rsrc(e.w, (unsigned)e.w_bytes);

  // ---- fragment read offsets: tile rows 2w..2w+3 of the halo, column offset dx, this lane's pixel r
  int p_off[12];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int p = (wave * 2 + rr) * RG_IW + dx + r;
      p_off[rr * 3 + dx] = (2 * p + (h ^ ((p >> 3) & 1))) * 16;
    }
  const int w_off = RG_ACT + (h * 64 + r) * 16;

  auto unit_coords = [&](int ord, int& b, int& oy0, int& ox0, int& n0) {
    const int uidx = band0 + slot + ord * nslots;
    const int nb = uidx % nco, t = uidx / nco;
    const int tile = t % tiles;
    b = t / tiles;
    const int ty = tile / tiles_x;
    oy0 = ty * RG_TH; ox0 = (tile - ty * tiles_x) * 32; n0 = nb * 64;
  };

  // ---- issue cursor: (unit, chunk) of the next DMA batch; a batch is issued piece by piece between the MFMAs
  int iu = 0, ic = 0, gi = 0;         // unit ordinal, chunk, batches issued so far (stage = gi % RG_NS)
  unsigned voff[PPW];
  i32x4 rsrc_i = rsrc_w;
  unsigned soff_i = 0, stage_i = 0;
  auto issue_begin = [&]() {
    if (ic == 0) {                    // descriptors of unit `iu`
      int b, oy0, ox0, n0;
      unit_coords(iu, b, oy0, ox0, n0);
      if (!loader_w) {
        rsrc_i = rg_rsrc(static_cast<const unsigned char*>(e.src) + (unsigned long long)b * img_bytes, img_bytes);
        const int base = e.halfsplit ? ((oy0 - 1) * 2 * W + (ox0 - 1)) * 16 : ((oy0 - 1) * W + (ox0 - 1)) * 32;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
          const int gy = oy0 - 1 + d_iy[j], gx = ox0 - 1 + d_ix[j];
          const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
          voff[j] = ok ? (unsigned)(base + d_rel[j]) : 0x80000000u;      // out of range => the DMA writes zeros
        }
      } else {
#pragma unroll
        for (int j = 0; j < PPW; ++j) voff[j] = d_iy[j] == 0 ? (unsigned)(d_rel[j] + n0 * 16) : 0x80000000u;
      }
      soff_i = 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) rsrc_i[k] = __builtin_amdgcn_readfirstlane(rsrc_i[k]);
    soff_i = __builtin_amdgcn_readfirstlane(soff_i);
    stage_i = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((gi % RG_NS) * STAGE));
  };
  auto issue_piece = [&](int j) {     // j is a compile-time constant at every call site
    if (DBG & 2) return;
    if (!loader) return;
    const unsigned dst = __builtin_amdgcn_readfirstlane(dst_off[j] >= 0 ? stage_i + (unsigned)dst_off[j] : lds0 + (unsigned)L::DUMP);
    rg_dma1(voff[j], rsrc_i, soff_i, dst);
  };
  auto issue_end = [&]() {
    ++gi;
    if (++ic == nc) { ic = 0; ++iu; }
    soff_i = loader_w ? (unsigned)ic * wchunk : (unsigned)(e.plane_wrap ? ic % e.plane_wrap : ic) * plane;
  };
  const int total = my_units * nc;    // chunk batches of this workgroup
#pragma unroll
  for (int k = 0; k < RG_NS - 1; ++k)
    if (k < total) {
      issue_begin();
#pragma unroll
      for (int j = 0; j < PPW; ++j) issue_piece(j);
      issue_end();
    }

  const float slope = a.act == CDFO_ACT_NONE ? 1.f : (a.act == CDFO_ACT_LRELU ? 0.1f : 0.f);

  int g = 0;                          // batch being consumed
  // four-tap form: LDS byte addresses (stage base included) of the six fragment positions of the chunk about to be consumed --
  // halo rows 2w + rr + y0, columns x0 + dx + r.  Computed one chunk AHEAD, behind the previous chunk's last MFMAs: nothing but
  // the fragment reads themselves sits between a chunk's barrier and its first MFMA.
  int f_cur[3][2] = {{0, 0}, {0, 0}, {0, 0}};
  auto frag_addresses = [&](int win, int stage_base) {
    const int y0 = win >> 1, x0 = win & 1;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int p = (wave * 2 + rr + y0) * RG_IW + x0 + dx + r;
        f_cur[rr][dx] = stage_base + (2 * p + (h ^ ((p >> 3) & 1))) * 16;
      }
  };
  if (SPARSE) frag_addresses(__builtin_amdgcn_readfirstlane(win_of(e.tap_mask[0])), 0);
  for (int ord = 0; ord < my_units; ++ord) {
    // transposed product: M = output channels (weights = A operand), N = pixels -> acc[ni][mi][8jj + q] is channel
    // ni*32 + jj*16 + h*8 + q of pixel r in tile row 2w + mi
    f32x16 acc[2][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[ni][mi][q] = 0.f;
    int b, oy0, ox0, n0;
    unit_coords(ord, b, oy0, ox0, n0);
    const int oyw = oy0 + wave * 2;
    // half-resolution residual tile: local pixel (i, j) = clamped source pixel (oy0/2 - 1 + i, ox0/2 - 1 + j); LDS slot
    // (pixel, 16-byte part q ^ (pixel & 15)) so that the epilogue's b128 reads of neighbouring pixels do not conflict
    unsigned evoff[RG_ET_PIECES / 8];
    i32x4 ersrc = rsrc_w;
    const unsigned edst = lds0 + (unsigned)L::ETILE + (unsigned)(wave * (RG_ET_PIECES / 8)) * 1024u;
    if (SPARSE && a.res_up2) {
      const int Hd = H >> 1, Wd = W >> 1;
      ersrc = rg_rsrc(a.res_up2 + (long long)b * Hd * Wd * a.ldru, (unsigned)(Hd * Wd * a.ldru) * 4u);
#pragma unroll
      for (int k = 0; k < 4; ++k) ersrc[k] = __builtin_amdgcn_readfirstlane(ersrc[k]);
#pragma unroll
      for (int j = 0; j < RG_ET_PIECES / 8; ++j) {
        const int slot = (wave * (RG_ET_PIECES / 8) + j) * 64 + lane;      // 16-byte slot of the tile image
        const int px = slot >> 4, q = (slot & 15) ^ (px & 15);
        const int ei = px / RG_ET_COLS, ej = px - ei * RG_ET_COLS;
        int gy = (oy0 >> 1) - 1 + ei, gx = (ox0 >> 1) - 1 + ej;
        gy = gy < 0 ? 0 : (gy >= Hd ? Hd - 1 : gy);
        gx = gx < 0 ? 0 : (gx >= Wd ? Wd - 1 : gx);
        evoff[j] = px < RG_ET_ROWS * RG_ET_COLS ? (unsigned)((gy * Wd + gx) * a.ldru + n0 + q * 4) * 4u : 0x80000000u;
      }
    }

    // one chunk: wait + barrier, then the taps' MFMAs with the pieces of batch g+3 issued between them (a DMA
    // instruction takes 100-200 cycles to issue; behind a tap's four MFMAs that time is covered by the matrix pipe)
    unsigned long long ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int i) {           // i is a compile-time constant at every call site
      if (DBG & 16) asm volatile("s_memtime %0" : "=s"(ts[i]) : : "memory");
    };
    auto chunk = [&](int c) {
      stamp(0);
      // (1) my pieces of batch g have landed: the younger DMA batches are (NS - 2) PPW instructions, and DMA pieces
      //     retire in issue order among themselves.  Other vector-memory operations (epilogue stores, residual loads) are
      //     NOT counted: LDS-DMA loads do not retire in order relative to VGPR loads (measured in conv3x3_ws.hip), so
      //     "at most (NS - 2) PPW outstanding" is the only bound that implies batch g has landed whatever else is in flight.
      //     (2) my fragment reads of batch g-1 have RETURNED (its stage is overwritten after the barrier).  Then the
      //     barrier: everyone's pieces of g are in LDS, nobody still reads the stage of g-1.
      if (g + RG_NS - 2 >= total) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else rg_wait_vm_lgkm0<(RG_NS - 2) * PPW>();
      static_assert((RG_NS - 2) * PPW < 64, "vmcnt is a 6-bit counter");
      stamp(1);
      __builtin_amdgcn_s_barrier();
      stamp(2);
      const bool do_issue = g + RG_NS - 1 < total;     // batch g+3 -> the stage batch g-1 lived in
      if (do_issue) issue_begin();
      const unsigned char* st = smem + (g % RG_NS) * STAGE;
      const unsigned char* sW = st + w_off;
      f16x8_t fa[2][2], fb[2][2];                      // [parity][tile]: fragments are read one tap ahead
      auto mma_tap = [&](int par) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            if (DBG & 1) acc[ni][mi][0] += (float)fa[par][mi][0] * (float)fb[par][ni][0];
            else if (DBG & 4) {      // clock experiment: the 16x16x32 shape at equal FLOPs and fragment reads (NOT the convolution)
              f32x4 lo = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
              f32x4 hi = {acc[ni][mi][4], acc[ni][mi][5], acc[ni][mi][6], acc[ni][mi][7]};
              lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[par][ni], fa[par][mi], lo, 0, 0, 0);
              hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[par][ni], fa[par][mi], hi, 0, 0, 0);
#pragma unroll
              for (int q = 0; q < 4; ++q) { acc[ni][mi][q] = lo[q]; acc[ni][mi][4 + q] = hi[q]; }
            } else acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[par][ni], fa[par][mi], acc[ni][mi], 0, 0, 0);
          }
      };
      if (SPARSE) {
        // (round 3: no switch over the four windows -- with four copies of the tap loop hipcc sits at 256 VGPRs and spills; the
        // window origin enters through the six fragment addresses f_cur)
        auto load_frags = [&](int j, int par) {      // j = dy*2 + dx inside the window = slab slot
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) fa[par][mi] = *reinterpret_cast<const f16x8_t*>(smem + f_cur[mi + (j >> 1)][j & 1]);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) fb[par][ni] = *reinterpret_cast<const f16x8_t*>(sW + (j * 2 * 64 + ni * 32) * 16);
        };
        load_frags(0, 0);
        // window origin of the NEXT chunk (the next tile's chunk 0 after the last one): an LDS byte, consumed after the taps
        const int win_n = smem[L::WINTAB + (c + 1 == nc ? 0 : c + 1)];
        stamp(3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < 3) load_frags(j + 1, (j & 1) ^ 1);
          // (scheduling fences: keep the next tap's fragment reads in FRONT of this tap's MFMAs -- hipcc otherwise sinks
          // them behind and waits on a just-issued ds_read at every tap)
          __builtin_amdgcn_sched_barrier(0);
          mma_tap(j & 1);
          __builtin_amdgcn_sched_barrier(0);
          if (do_issue) issue_piece(j);
          stamp(4 + j);
        }
        frag_addresses(__builtin_amdgcn_readfirstlane(win_n), ((g + 1) % RG_NS) * STAGE);
      } else {
        auto load_frags = [&](int t, int par) {
          const int dy = t / 3, dx = t - dy * 3;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) fa[par][mi] = *reinterpret_cast<const f16x8_t*>(st + p_off[(mi + dy) * 3 + dx]);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) fb[par][ni] = *reinterpret_cast<const f16x8_t*>(sW + (t * 2 * 64 + ni * 32) * 16);
        };
        load_frags(0, 0);
        stamp(3);
#pragma unroll
        for (int t = 0; t < 9; ++t) {     // (no scheduling fences here: measured 2 % slower with them in the dense form,
          if (t < 8) load_frags(t + 1, (t & 1) ^ 1);      //  1.7 % faster in the four-tap form above, same box)
          mma_tap(t & 1);
          if (do_issue && t < PPW) issue_piece(t);
          if (DBG & 16) {
            if (t == 1 || t == 3 || t == 5 || t == 8) {
              __builtin_amdgcn_sched_barrier(0);
              stamp(t == 8 ? 7 : 4 + (t >> 1));
            }
          }
        }
      }
      if (do_issue) issue_end();
      if (DBG & 16) {
        stamp(8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ord == RG_PROBE_UNIT && c >= 8 && c < 12 && lane == 0) {
          unsigned long long* cb = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res2)) +
                                   (((long long)blockIdx.x * 8 + wave) * 4 + (c - 8)) * 10;
#pragma unroll
          for (int i = 0; i < 9; ++i) cb[i] = ts[i];
        }
      }
      // half-resolution residual tile of THIS output tile (consumed by its epilogue, 60+ chunks from now): 48 DMA
      // pieces, 6 per wave.  Younger than this chunk's batch, so the counted waits above (which only allow the newest
      // (NS - 2) PPW pieces to be pending) retire them within the next two chunks.
      if (SPARSE && c == 0 && a.res_up2 && !(DBG & 2)) {
#pragma unroll
        for (int j = 0; j < RG_ET_PIECES / 8; ++j) rg_dma1(evoff[j], ersrc, 0u, __builtin_amdgcn_readfirstlane(edst + j * 1024));
      }
    };

    for (int c = 0; c < nc - 1; ++c, ++g) chunk(c);
    // ---- last chunk of the tile (peeled so that the residual registers are live only here): the residual values are
    // fetched while its MFMAs run.  A lane owns pixel (row oyw + mi, column ox0 + r) and, per (ni, jj), the 8 channels
    // n0 + ni*32 + jj*16 + h*8 .. + 7 = 32 contiguous bytes of every fp32 pixel-major operand.
    const bool has_res = a.res1 != nullptr && !(DBG & 8);
    const bool px_ok[2] = {oyw < H && ox0 + r < W, oyw + 1 < H && ox0 + r < W};
    f32x4 rv[2][2][2][2];                     // [mi][ni][jj][half]
#pragma unroll
    for (int i = 0; i < 16; ++i) (&rv[0][0][0][0])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_res = [&]() {
      if (!has_res) return;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const float* rp = a.res1 + ((long long)(b * H + oyw + mi) * W + ox0 + r) * a.ldr1 + n0 + h * 8;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            if (px_ok[mi] && n0 + ni * 32 + jj * 16 + h * 8 < a.Cout) {
              rv[mi][ni][jj][0] = *reinterpret_cast<const f32x4*>(rp + ni * 32 + jj * 16);
              rv[mi][ni][jj][1] = *reinterpret_cast<const f32x4*>(rp + ni * 32 + jj * 16 + 4);
            }
      }
    };
    // (four-tap form: 64 chunks per tile, the window switch needs the registers -> fetch after the last chunk)
    if (!SPARSE) load_res();
    chunk(nc - 1);
    ++g;
    if (SPARSE) load_res();

    // ---- epilogue straight from the accumulators: +bias -> act -> +res1 -> +res2 -> +bilinear x2 of res_up2 -> stores
    // (32 contiguous bytes per lane and channel group; the fp16 chunk-planar copy: 16 bytes per lane, 1 KiB per wave)
    if (DBG & 8) {
      float t = 0.f;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int q = 0; q < 16; ++q) t += acc[ni][mi][q];
      if (t == 123.456f) a.out[0] = t;
      continue;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int oy = oyw + mi, X = ox0 + r;
      const long long pix = (long long)(b * H + oy) * W + X;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int n = n0 + ni * 32 + jj * 16 + h * 8;
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(smem + L::BIAS + n * 4);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(smem + L::BIAS + n * 4 + 16);
          f32x4 v0, v1;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float t0 = acc[ni][mi][8 * jj + k] + b0[k], t1 = acc[ni][mi][8 * jj + 4 + k] + b1[k];
            v0[k] = fmaxf(t0, slope * t0);      // slope in [0, 1]: identity / LeakyReLU / ReLU
            v1[k] = fmaxf(t1, slope * t1);
          }
          v0 += rv[mi][ni][jj][0];              // unconditional use: no residual load may stay "pending" across tiles
          v1 += rv[mi][ni][jj][1];
          if (!px_ok[mi] || n >= a.Cout) continue;
          if (a.res2 && !(DBG & 16)) {
            const float* p2 = a.res2 + pix * a.ldr2 + n;
            v0 += *reinterpret_cast<const f32x4*>(p2);
            v1 += *reinterpret_cast<const f32x4*>(p2 + 4);
          }
          if (a.res_up2) {      // + bilinear x2 of a half-resolution tensor: taps (Q-1, Q) x (P-1, P), clamped
            const float ly = (oy & 1) ? 0.25f : 0.75f, lx = (X & 1) ? 0.25f : 0.75f;
            if (SPARSE) {       // from the staged tile: local tap rows Q - oy0/2 (+1), columns P - ox0/2 (+1)
              const int li = ((oy + 1) >> 1) - (oy0 >> 1), lj = ((X + 1) >> 1) - (ox0 >> 1);
              const unsigned char* et = smem + L::ETILE;
              const int cq = (ni * 32 + jj * 16 + h * 8) >> 2;          // 16-byte part of the tile's 64 channels
#pragma unroll
              for (int hf = 0; hf < 2; ++hf) {
                auto tap = [&](int i, int j) {
                  const int px = i * RG_ET_COLS + j;
                  return *reinterpret_cast<const f32x4*>(et + px * 256 + (((cq + hf) ^ (px & 15)) << 4));
                };
                (hf ? v1 : v0) += (1.f - ly) * ((1.f - lx) * tap(li, lj) + lx * tap(li, lj + 1)) +
                                  ly * ((1.f - lx) * tap(li + 1, lj) + lx * tap(li + 1, lj + 1));
              }
            } else {
              const int Hd = H >> 1, Wd = W >> 1;
              const int Q = (oy + 1) >> 1, P = (X + 1) >> 1;
              const int ya = Q > 0 ? Q - 1 : 0, yb = Q < Hd ? Q : Hd - 1, xa = P > 0 ? P - 1 : 0, xb = P < Wd ? P : Wd - 1;
              const float* eb = a.res_up2 + (long long)b * Hd * Wd * a.ldru + n;
              const float* paa = eb + ((long long)ya * Wd + xa) * a.ldru;
              const float* pab = eb + ((long long)ya * Wd + xb) * a.ldru;
              const float* pba = eb + ((long long)yb * Wd + xa) * a.ldru;
              const float* pbb = eb + ((long long)yb * Wd + xb) * a.ldru;
#pragma unroll
              for (int hf = 0; hf < 2; ++hf) {
                const f32x4 eaa = *reinterpret_cast<const f32x4*>(paa + 4 * hf), eab = *reinterpret_cast<const f32x4*>(pab + 4 * hf);
                const f32x4 eba = *reinterpret_cast<const f32x4*>(pba + 4 * hf), ebb = *reinterpret_cast<const f32x4*>(pbb + 4 * hf);
                (hf ? v1 : v0) += (1.f - ly) * ((1.f - lx) * eaa + lx * eab) + ly * ((1.f - lx) * eba + lx * ebb);
              }
            }
          }
          f16x8_t hv;
#pragma unroll
          for (int k = 0; k < 4; ++k) { hv[k] = (_Float16)v0[k]; hv[4 + k] = (_Float16)v1[k]; }
          if (a.out_f16) {
            *reinterpret_cast<f16x8_t*>(reinterpret_cast<_Float16*>(a.out) + pix * a.ldo + n) = hv;
          } else {
            *reinterpret_cast<f32x4*>(a.out + pix * a.ldo + n) = v0;
            *reinterpret_cast<f32x4*>(a.out + pix * a.ldo + n + 4) = v1;
          }
          if (a.out2_cp16) { // chunk-planar fp16 copy: record (image, chunk n/16, pixel), halves h*8 .. h*8+7
            const int npl = a.out2_lo ? (a.Cout >> 3) : (a.Cout >> 4);      // planes per image: hi | lo, or hi only
            _Float16* o2 = static_cast<_Float16*>(a.out2_cp16) + (((long long)b * npl + (n >> 4)) * H * W + (long long)oy * W + X) * 16 + h * 8;
            *reinterpret_cast<f16x8_t*>(o2) = hv;
            if (a.out2_lo) {   // + the remainders: the pair is the source of a split-fp16 (hi + lo activations) convolution
              f16x8_t lv;
#pragma unroll
              for (int k = 0; k < 4; ++k) { lv[k] = (_Float16)(v0[k] - (float)hv[k]); lv[4 + k] = (_Float16)(v1[k] - (float)hv[4 + k]); }
              *reinterpret_cast<f16x8_t*>(o2 + (long long)(a.Cout >> 4) * H * W * 16) = lv;
            }
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised form of the same kernel (round 3; same LDS map, stage layout, DMA pieces, unit order and epilogue
// arithmetic as conv3x3_ring_kernel above -- the results are bit-identical).
//
// Why.  s_memtime stamps inside the kernel above (dbg 16, tools/ring_timeline.py) show where a chunk's cycles go: with eight
// identical waves -- two per SIMD, each owning 2 tile rows and issuing its share of the DMA pieces between its taps -- a wave
// needs 250-400 cycles per tap for 128 cycles of MFMA (a DMA instruction costs 100-200 issue cycles, every wave carries the
// ring's scalar bookkeeping), the younger wave of each SIMD loses the issue arbitration and finishes a chunk 600-900 cycles
// after the older one, which then sits at the barrier; 290-400 cycles pass between the barrier and a chunk's first MFMA.  The
// matrix pipe is busy 36 % (four-tap form) / 47 % (dense) of the launch (SQ counters, profiles/r03_pmc_counters_conv3x3_ring.txt).
// Here the two jobs are separated:
//   * waves 0-3 (one per SIMD) are CONSUMERS: 4 tile rows x 32 pixels x 64 channels each (8 accumulators), nothing in their
//     instruction stream but fragment reads and MFMAs (0.75 ds_read_b128 per MFMA instead of 1);
//   * waves 4-7 (the other wave of each SIMD) are PRODUCERS: all DMA pieces of a chunk (dense: 10 per wave; four-tap form: 8),
//     the ring cursor, the half-resolution residual tile -- scalar work and DMA issue that now overlap the consumer's MFMAs
//     instead of interrupting them.
// One workgroup barrier per chunk, as before: producers arrive once THEIR pieces of batch g have landed (counted vmcnt),
// consumers once their fragment reads of batch g-1 have returned; behind it the producers refill the stage of batch g-1.
// Both role loops execute exactly my_units * nc barriers.
// MF16 (four-tap form only): the consumers run v_mfma_f32_16x16x32_f16 instead of 32x32x16 -- the loop is power-limited and the
// chip holds a higher clock under that shape (see conv3x3_ws.hip, conv3x3_c64_wsq_kernel): K = 32 = the two taps of a window row
// x 16 channels; a lane holds k-group l >> 4 = (tap of the pair, 8-channel half) of pixel / output channel l & 15; natural
// [pixel][16 ch] records (no half swizzle: this read pattern is conflict-free without it) and a weight-row order in which a
// lane's accumulators of channel blocks 2e, 2e+1 are 8 consecutive channels (the epilogue keeps its 32-byte granularity).
template <bool SPARSE, int DBG, bool MF16 = false>
__global__ __launch_bounds__(RG_THREADS) void conv3x3_ring_split_kernel(cdfo_conv_args a, ring_extra e) {
  static_assert(!MF16 || SPARSE, "the 16x16x32 consumer exists for the four-tap form");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TAPS = SPARSE ? 4 : 9;
  using L = RingLds<SPARSE>;
  constexpr int STAGE = L::STAGE, RG_NS = L::NS;
  constexpr int CR = 4;                               // tile rows per consumer wave
  // producer pieces per chunk: dense 10 + 10 activation pieces, 9 + 9 weight pieces (+1 pad each); four-tap form 7 + 7 + 6
  // activation pieces (+1, +1, +2 pads into the dump kilobyte) and 8 weight pieces: every producer issues exactly PPW
  // instructions per batch, which is what the counted waits count
  constexpr int PPW = SPARSE ? 8 : 10, ACT_PW = SPARSE ? 3 : 2, ACT_PER = SPARSE ? 7 : 10, WGT_PER = SPARSE ? 8 : 9;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave < 4;
  const int H = a.H, W = a.W, nc = e.nc;
  const int tiles_x = (W + 31) >> 5, tiles_y = (H + RG_TH - 1) / RG_TH, tiles = tiles_x * tiles_y;
  const int nco = a.CoutP >> 6;
  const int units = a.B * tiles * nco;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int band = (units + 7) >> 3, band0 = xcd * band;
  const int band_n = min(band, units - band0);
  const int my_units = band_n > slot ? (band_n - slot + nslots - 1) / nslots : 0;
  if (my_units == 0) return;
  const unsigned lds0 = (unsigned)(unsigned long long)(smem);
  for (int i = tid; i < a.CoutP; i += RG_THREADS)
    reinterpret_cast<float*>(smem + L::BIAS)[i] = (a.bias && i < a.Cout) ? a.bias[i] : 0.f;
  auto win_of = [](unsigned tm) { return ((tm & 0x7u) ? 0 : 2) + ((tm & 0x49u) ? 0 : 1); };
  if (SPARSE)
    for (int i = tid; i < nc; i += RG_THREADS) smem[L::WINTAB + i] = (unsigned char)win_of(e.tap_mask[i]);
  // (bias and window table become visible to the consumers through the chunk barriers: the first read of either comes after
  // the first barrier)
  const int total = my_units * nc;                    // chunk batches = barriers of this workgroup
  auto unit_coords = [&](int ord, int& b, int& oy0, int& ox0, int& n0) {
    const int uidx = band0 + slot + ord * nslots;
    const int nb = uidx % nco, t = uidx / nco;
    const int tile = t % tiles;
    b = t / tiles;
    const int ty = tile / tiles_x;
    oy0 = ty * RG_TH; ox0 = (tile - ty * tiles_x) * 32; n0 = nb * 64;
  };

  if (!consumer) {
    // ================================================================================================ producers
    const int pw = wave - 4;
    const bool loader_w = pw >= ACT_PW;
    int d_iy[PPW], d_ix[PPW], d_rel[PPW], dst_off[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      if (!loader_w) {
        const int q = pw * ACT_PER + j;
        const bool real = j < ACT_PER && q < 20;
        const int s = q * 64 + lane, p = s >> 1, half = MF16 ? (s & 1) : (s & 1) ^ ((p >> 3) & 1);
        const int iy = p / RG_IW, ix = p - iy * RG_IW;
        d_iy[j] = (real && p < RG_NPIX) ? iy : 1 << 20;
        d_ix[j] = ix;
        d_rel[j] = e.halfsplit ? ((iy * 2 + half) * W + ix) * 16 : (iy * W + ix) * 32 + half * 16;
        dst_off[j] = real ? q * 1024 : -1;
      } else {
        const int q = (pw - ACT_PW) * WGT_PER + j;
        const bool real = j < WGT_PER && q < TAPS * 2;
        d_iy[j] = real ? 0 : 1 << 20;
        d_ix[j] = 0;
        const int m = lane & 31;      // slab row position `lane` = MFMA row m of block (lane >> 5), see the kernel above
        const int chan = MF16 ? (lane >> 5) * 32 + ((lane & 15) >> 2) * 8 + ((lane >> 4) & 1) * 4 + (lane & 3)
                              : (lane & 32) + ((m >> 4) & 1) * 16 + ((m >> 2) & 1) * 8 + ((m >> 3) & 1) * 4 + (m & 3);
        d_rel[j] = (q * e.CoutP + chan) * 16;
        dst_off[j] = real ? RG_ACT + q * 1024 : -1;
      }
    }
    const unsigned plane = (unsigned)(H * W) * 32u;
    const unsigned wchunk = (unsigned)(TAPS * 2 * e.CoutP) * 16u;
    const unsigned img_bytes = plane * (unsigned)e.src_planes;
    const i32x4 rsrc_w = rg_rsrc(e.w, (unsigned)e.w_bytes);
    int iu = 0, ic = 0, gi = 0;
    unsigned voff[PPW];
    i32x4 rsrc_i = rsrc_w;
    unsigned soff_i = 0;
    auto issue_batch = [&]() {
      if (ic == 0) {                    // descriptors of unit `iu`
        int b, oy0, ox0, n0;
        unit_coords(iu, b, oy0, ox0, n0);
        if (!loader_w) {
          rsrc_i = rg_rsrc(static_cast<const unsigned char*>(e.src) + (unsigned long long)b * img_bytes, img_bytes);
          const int base = e.halfsplit ? ((oy0 - 1) * 2 * W + (ox0 - 1)) * 16 : ((oy0 - 1) * W + (ox0 - 1)) * 32;
#pragma unroll
          for (int j = 0; j < PPW; ++j) {
            const int gy = oy0 - 1 + d_iy[j], gx = ox0 - 1 + d_ix[j];
            const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
            voff[j] = ok ? (unsigned)(base + d_rel[j]) : 0x80000000u;
          }
        } else {
#pragma unroll
          for (int j = 0; j < PPW; ++j) voff[j] = d_iy[j] == 0 ? (unsigned)(d_rel[j] + n0 * 16) : 0x80000000u;
        }
        soff_i = 0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) rsrc_i[k] = __builtin_amdgcn_readfirstlane(rsrc_i[k]);
      soff_i = __builtin_amdgcn_readfirstlane(soff_i);
      const unsigned stage_i = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((gi % RG_NS) * STAGE));
      if (!(DBG & 2)) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
          const unsigned dst = __builtin_amdgcn_readfirstlane(dst_off[j] >= 0 ? stage_i + (unsigned)dst_off[j] : lds0 + (unsigned)L::DUMP);
          rg_dma1(voff[j], rsrc_i, soff_i, dst);
        }
      }
      ++gi;
      if (++ic == nc) { ic = 0; ++iu; }
      soff_i = loader_w ? (unsigned)ic * wchunk : (unsigned)(e.plane_wrap ? ic % e.plane_wrap : ic) * plane;
    };
#pragma unroll
    for (int k = 0; k < RG_NS - 1; ++k)
      if (k < total) issue_batch();

    int g = 0;
    int ex_prev1 = 0, ex_prev2 = 0;     // extras (pieces issued behind a batch) of the previous / second-previous iteration
    for (int ord = 0; ord < my_units; ++ord) {
      // Extras of unit `ord`, issued AFTER a chunk's batch (so that the counted wait of the next chunk can allow exactly
      // "the newest batch(es) + the extras issued behind them" to be pending and never asks a just-issued batch to land):
      //   * four-tap form: the half-resolution residual tile, 48 pieces, 12 per producer, two per chunk over the first six chunks;
      //   * both forms: TOUCHES of the tile's res1 lines -- 128 KiB that the consumers' epilogue reads with one exposed memory
      //     latency per tile row (all 256 workgroups reach their epilogues together: 4-5 us per row from HBM under that burst,
      //     18-34 k cycles per tile, dbg 16) -- as DMA pieces into the dump kilobyte over the tile's last eight chunks: the lines
      //     are then in L2 / the Infinity Cache when the epilogue asks for them, and their HBM reads fall into the MFMA phase.
      constexpr int EPP = RG_ET_PIECES / 4, EPC = 2;       // residual-tile pieces per producer, per chunk
      constexpr int TCH = 8, TPC = 4;                      // touch chunks per tile, touch pieces per producer and chunk (4 x 8 x 4 = 128)
      unsigned evoff[SPARSE ? EPP : 1];
      i32x4 ersrc = rsrc_w, trsrc = rsrc_w;
      const bool etile = SPARSE && a.res_up2 && !(DBG & 2);
      const bool touch = a.res1 != nullptr && !(DBG & (2 | 8)) && rg_touch_on(e);
      int ub, uoy0, uox0, un0;
      unit_coords(ord, ub, uoy0, uox0, un0);
      if (etile) {
        const int Hd = H >> 1, Wd = W >> 1;
        ersrc = rg_rsrc(a.res_up2 + (long long)ub * Hd * Wd * a.ldru, (unsigned)(Hd * Wd * a.ldru) * 4u);
#pragma unroll
        for (int k = 0; k < 4; ++k) ersrc[k] = __builtin_amdgcn_readfirstlane(ersrc[k]);
#pragma unroll
        for (int j = 0; j < EPP; ++j) {
          const int eslot = (pw * EPP + j) * 64 + lane;      // 16-byte slot of the tile image
          const int px = eslot >> 4, q = (eslot & 15) ^ (px & 15);
          const int ei = px / RG_ET_COLS, ej = px - ei * RG_ET_COLS;
          int gy = (uoy0 >> 1) - 1 + ei, gx = (uox0 >> 1) - 1 + ej;
          gy = gy < 0 ? 0 : (gy >= Hd ? Hd - 1 : gy);
          gx = gx < 0 ? 0 : (gx >= Wd ? Wd - 1 : gx);
          evoff[j] = px < RG_ET_ROWS * RG_ET_COLS ? (unsigned)((gy * Wd + gx) * a.ldru + un0 + q * 4) * 4u : 0x80000000u;
        }
      }
      if (touch) {
        trsrc = rg_rsrc(a.res1 + (long long)ub * H * W * a.ldr1, (unsigned)(H * W * a.ldr1) * 4u);
#pragma unroll
        for (int k = 0; k < 4; ++k) trsrc[k] = __builtin_amdgcn_readfirstlane(trsrc[k]);
      }
      const unsigned edst = lds0 + (unsigned)L::ETILE + (unsigned)(pw * EPP) * 1024u;
      const unsigned dump = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)L::DUMP);
      const int t0 = nc - TCH;                        // first touch chunk (tiles of fewer chunks are touched in part)
      for (int c = 0; c < nc; ++c, ++g) {
        unsigned long long ts[4] = {0, 0, 0, 0};     // dbg 16: entry, pieces landed, barrier passed, batch issued
        if (DBG & 16) asm volatile("s_memtime %0" : "=s"(ts[0]) : : "memory");
        // my pieces of batch g have landed: everything but the newest (NS - 2) batches and the extras issued behind them in
        // the last NS - 2 iterations may be pending (DMA pieces retire in issue order among themselves; a producer issues
        // nothing else)
        {
          const int allow = (RG_NS - 2) * PPW + ex_prev1 + (RG_NS > 3 ? ex_prev2 : 0);
          if (g + RG_NS - 2 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if (allow == (RG_NS - 2) * PPW) rg_wait_vm<(RG_NS - 2) * PPW>();
          else if (allow == (RG_NS - 2) * PPW + EPC) rg_wait_vm<(RG_NS - 2) * PPW + EPC>();
          else if (allow == (RG_NS - 2) * PPW + TPC) rg_wait_vm<(RG_NS - 2) * PPW + TPC>();
          else if (allow == (RG_NS - 2) * PPW + 2 * EPC) rg_wait_vm<(RG_NS - 2) * PPW + 2 * EPC>();
          else if (allow == (RG_NS - 2) * PPW + 2 * TPC) rg_wait_vm<(RG_NS - 2) * PPW + 2 * TPC>();
          else rg_wait_vm<(RG_NS - 2) * PPW>();      // (any other mix: the plain bound is always safe)
          static_assert((RG_NS - 2) * PPW + 2 * TPC < 64, "vmcnt is a 6-bit counter");
        }
        if (DBG & 16) asm volatile("s_memtime %0" : "=s"(ts[1]) : : "memory");
        __builtin_amdgcn_s_barrier();
        if (DBG & 16) asm volatile("s_memtime %0" : "=s"(ts[2]) : : "memory");
        // behind the barrier nobody reads the stage of batch g-1 any more (and, at c == 0, the previous tile's epilogue is
        // through with the residual tile)
        if (g + RG_NS - 1 < total) issue_batch();
        int extras = 0;
        if (SPARSE && etile && c < EPP / EPC) {
          // (c is a loop variable: the two pieces are selected by a uniform switch so that evoff[] stays in registers)
#pragma unroll
          for (int k = 0; k < EPP / EPC; ++k)
            if (c == k) {
              rg_dma1(evoff[2 * k], ersrc, 0u, __builtin_amdgcn_readfirstlane(edst + (2 * k) * 1024));
              rg_dma1(evoff[2 * k + 1], ersrc, 0u, __builtin_amdgcn_readfirstlane(edst + (2 * k + 1) * 1024));
            }
          extras += EPC;
        }
        if (touch && c >= t0 && c >= 0) {
#pragma unroll
          for (int j = 0; j < TPC; ++j) {
            // piece pid of the tile's 128: tile row pid >> 3, four pixels x 256 bytes from column 4 (pid & 7); lane = (pixel, 16-byte part)
            const int pid = pw * (TCH * TPC) + (c - t0) * TPC + j;
            const int gy = uoy0 + (pid >> 3), gx = uox0 + (pid & 7) * 4 + (lane >> 4);
            const unsigned vo = (gy < H && gx < W) ? (unsigned)((gy * W + gx) * a.ldr1 + un0 + (lane & 15) * 4) * 4u : 0x80000000u;
            rg_dma1(vo, trsrc, 0u, dump);
          }
          extras += TPC;
        }
        ex_prev2 = ex_prev1;
        ex_prev1 = extras;
        if (DBG & 16) {
          asm volatile("s_memtime %0" : "=s"(ts[3]) : : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (ord == RG_PROBE_UNIT && c >= 8 && c < 12 && lane == 0) {
            unsigned long long* cb = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res2)) +
                                     (((long long)blockIdx.x * 8 + wave) * 4 + (c - 8)) * 10;
            cb[0] = ts[0]; cb[1] = ts[1]; cb[2] = ts[2]; cb[3] = ts[3];
#pragma unroll
            for (int i = 4; i < 9; ++i) cb[i] = ts[3];
          }
        }
      }
    }
    return;
  }

  // ================================================================================================== consumers
  const float slope = a.act == CDFO_ACT_NONE ? 1.f : (a.act == CDFO_ACT_LRELU ? 0.1f : 0.f);
  if constexpr (MF16) {
    const int l16 = lane & 15, kg = lane >> 4, ktap = kg >> 1, khalf = kg & 1;
    // weight fragment of (tap pair, channel block mb): slab row (pair*2 + ktap)*2 + khalf, position mb*16 + l16
    const int w16 = RG_ACT + (ktap * 2 + khalf) * 1024 + l16 * 16;
    // pixel fragments of the chunk about to be consumed (stage base included): halo rows 4w + rr + y0 (rr = row + pair = 0..4),
    // column x0 + ktap + l16 (+16: immediate); computed one chunk ahead
    int f16a[5] = {0, 0, 0, 0, 0};
    auto frag16 = [&](int win, int stage_base) {
      const int y0 = win >> 1, x0 = win & 1;
#pragma unroll
      for (int rr = 0; rr < 5; ++rr)
        f16a[rr] = stage_base + ((wave * CR + rr + y0) * RG_IW + x0 + ktap + l16) * 32 + khalf * 16;
    };
    frag16(__builtin_amdgcn_readfirstlane(win_of(e.tap_mask[0])), 0);
    int g = 0;
    for (int ord = 0; ord < my_units; ++ord) {
      // acc[mb][nbk][k]: block mb row 4 kg + k = channel (mb>>1)*32 + 8 kg + (mb&1)*4 + k; nbk = row*2 + pixel half
      f32x4 acc[4][2 * CR];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nbk = 0; nbk < 2 * CR; ++nbk) acc[mb][nbk] = f32x4{0.f, 0.f, 0.f, 0.f};
      int b, oy0, ox0, n0;
      unit_coords(ord, b, oy0, ox0, n0);
      const int oyw = oy0 + wave * CR;
      auto chunk = [&](int c) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my fragment reads of batch g-1 have returned
        __builtin_amdgcn_s_barrier();
        const unsigned char* sW = smem + (g % RG_NS) * STAGE + w16;
        // four sub-steps: (tap pair, rows 0-1), (pair, rows 2-3); the weight fragments of a pair serve both
        f16x8_t fw[2][4], fp[2][4];                 // [pair parity / sub-step parity][channel block / (row, pixel half)]
        auto load_w = [&](int pair) {
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) fw[pair & 1][mb] = *reinterpret_cast<const f16x8_t*>(sW + pair * 4096 + mb * 256);
        };
        auto load_p = [&](int ss) {
          const int pair = ss >> 1, hr = ss & 1;
#pragma unroll
          for (int q = 0; q < 4; ++q)              // q = (row within the half)*2 + pixel half
            fp[ss & 1][q] = *reinterpret_cast<const f16x8_t*>(smem + f16a[2 * hr + (q >> 1) + pair] + (q & 1) * 512);
        };
        load_w(0);
        load_p(0);
        const int win_n = smem[L::WINTAB + (c + 1 == nc ? 0 : c + 1)];
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
          if (ss < 3) load_p(ss + 1);
          if (ss == 1) load_w(1);
          __builtin_amdgcn_sched_barrier(0);
          const int pair = ss >> 1, hr = ss & 1;
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
              if (DBG & 1) acc[mb][hr * 4 + q][0] += (float)fw[pair & 1][mb][0] * (float)fp[ss & 1][q][0];
              else acc[mb][hr * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[pair & 1][mb], fp[ss & 1][q], acc[mb][hr * 4 + q], 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        frag16(__builtin_amdgcn_readfirstlane(win_n), ((g + 1) % RG_NS) * STAGE);
      };
      // residual values of one tile row: [pixel half][e][4-channel half] -- 32 contiguous bytes per (pixel, e)
      const bool has_res = a.res1 != nullptr && !(DBG & 8);
      auto px_ok = [&](int mi, int nh) { return oyw + mi < H && ox0 + nh * 16 + l16 < W; };
      auto load_res_row = [&](int mi, f32x4 (&rv)[2][2][2]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) (&rv[0][0][0])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!has_res) return;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          if (!px_ok(mi, nh)) continue;
          const float* rp = a.res1 + ((long long)(b * H + oyw + mi) * W + ox0 + nh * 16 + l16) * a.ldr1 + n0 + kg * 8;
#pragma unroll
          for (int ee = 0; ee < 2; ++ee)
            if (n0 + ee * 32 + kg * 8 < a.Cout) {
              rv[nh][ee][0] = *reinterpret_cast<const f32x4*>(rp + ee * 32);
              rv[nh][ee][1] = *reinterpret_cast<const f32x4*>(rp + ee * 32 + 4);
            }
        }
      };
      for (int c = 0; c < nc - 1; ++c, ++g) chunk(c);
      f32x4 rva[2][2][2], rvb[2][2][2];
      load_res_row(0, rva);
      chunk(nc - 1);
      ++g;
      if (DBG & 8) {
        float t = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int nbk = 0; nbk < 2 * CR; ++nbk)
#pragma unroll
            for (int k = 0; k < 4; ++k) t += acc[mb][nbk][k] + rva[nbk & 1][mb & 1][0][k];
        if (t == 123.456f) a.out[0] = t;
        continue;
      }
      auto epilogue_row = [&](int mi, const f32x4 (&rv)[2][2][2]) {
        const int oy = oyw + mi;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          const int X = ox0 + nh * 16 + l16;
          const bool ok = px_ok(mi, nh);
          const long long pix = (long long)(b * H + oy) * W + X;
#pragma unroll
          for (int ee = 0; ee < 2; ++ee) {
            const int n = n0 + ee * 32 + kg * 8;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(smem + L::BIAS + n * 4);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(smem + L::BIAS + n * 4 + 16);
            f32x4 v0, v1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float t0 = acc[2 * ee][mi * 2 + nh][k] + b0[k], t1 = acc[2 * ee + 1][mi * 2 + nh][k] + b1[k];
              v0[k] = fmaxf(t0, slope * t0);
              v1[k] = fmaxf(t1, slope * t1);
            }
            v0 += rv[nh][ee][0];
            v1 += rv[nh][ee][1];
            if (!ok || n >= a.Cout) continue;
            if (a.res2) {
              const float* p2 = a.res2 + pix * a.ldr2 + n;
              v0 += *reinterpret_cast<const f32x4*>(p2);
              v1 += *reinterpret_cast<const f32x4*>(p2 + 4);
            }
            if (a.res_up2) {      // + bilinear x2 of the half-resolution tensor, from the staged tile (see the form above)
              const float ly = (oy & 1) ? 0.25f : 0.75f, lx = (X & 1) ? 0.25f : 0.75f;
              const int li = ((oy + 1) >> 1) - (oy0 >> 1), lj = ((X + 1) >> 1) - (ox0 >> 1);
              const unsigned char* et = smem + L::ETILE;
              const int cq = (ee * 32 + kg * 8) >> 2;
#pragma unroll
              for (int hf = 0; hf < 2; ++hf) {
                auto tap = [&](int i, int j) {
                  const int px = i * RG_ET_COLS + j;
                  return *reinterpret_cast<const f32x4*>(et + px * 256 + (((cq + hf) ^ (px & 15)) << 4));
                };
                (hf ? v1 : v0) += (1.f - ly) * ((1.f - lx) * tap(li, lj) + lx * tap(li, lj + 1)) +
                                  ly * ((1.f - lx) * tap(li + 1, lj) + lx * tap(li + 1, lj + 1));
              }
            }
            f16x8_t hv;
#pragma unroll
            for (int k = 0; k < 4; ++k) { hv[k] = (_Float16)v0[k]; hv[4 + k] = (_Float16)v1[k]; }
            if (a.out_f16) {
              *reinterpret_cast<f16x8_t*>(reinterpret_cast<_Float16*>(a.out) + pix * a.ldo + n) = hv;
            } else {
              *reinterpret_cast<f32x4*>(a.out + pix * a.ldo + n) = v0;
              *reinterpret_cast<f32x4*>(a.out + pix * a.ldo + n + 4) = v1;
            }
            if (a.out2_cp16) {
              const int npl = a.out2_lo ? (a.Cout >> 3) : (a.Cout >> 4);
              _Float16* o2 = static_cast<_Float16*>(a.out2_cp16) + (((long long)b * npl + (n >> 4)) * H * W + (long long)oy * W + X) * 16 + (n & 15);
              *reinterpret_cast<f16x8_t*>(o2) = hv;
              if (a.out2_lo) {
                f16x8_t lv;
#pragma unroll
                for (int k = 0; k < 4; ++k) { lv[k] = (_Float16)(v0[k] - (float)hv[k]); lv[4 + k] = (_Float16)(v1[k] - (float)hv[4 + k]); }
                *reinterpret_cast<f16x8_t*>(o2 + (long long)(a.Cout >> 4) * H * W * 16) = lv;
              }
            }
          }
        }
      };
      load_res_row(1, rvb);
      epilogue_row(0, rva);
      load_res_row(2, rva);
      epilogue_row(1, rvb);
      load_res_row(3, rvb);
      epilogue_row(2, rva);
      epilogue_row(3, rvb);
    }
    return;
  }
  const int w_off = RG_ACT + (h * 64 + r) * 16;
  // dense: fragment offsets of halo rows 4w + rr (rr = 0..5), column offset dx, this lane's pixel r
  int p_off[SPARSE ? 1 : 18];
  if (!SPARSE) {
#pragma unroll
    for (int rr = 0; rr < 6; ++rr)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int p = (wave * CR + rr) * RG_IW + dx + r;
        p_off[SPARSE ? 0 : rr * 3 + dx] = (2 * p + (h ^ ((p >> 3) & 1))) * 16;
      }
  }
  // four-tap form: LDS byte addresses (stage base included) of the chunk about to be consumed: halo rows 4w + rr + y0
  // (rr = 0..4), columns x0 + dx + r -- computed one chunk ahead
  int f_cur[5][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
  auto frag_addresses = [&](int win, int stage_base) {
    const int y0 = win >> 1, x0 = win & 1;
#pragma unroll
    for (int rr = 0; rr < 5; ++rr)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int p = (wave * CR + rr + y0) * RG_IW + x0 + dx + r;
        f_cur[rr][dx] = stage_base + (2 * p + (h ^ ((p >> 3) & 1))) * 16;
      }
  };
  if (SPARSE) frag_addresses(__builtin_amdgcn_readfirstlane(win_of(e.tap_mask[0])), 0);

  int g = 0;
  for (int ord = 0; ord < my_units; ++ord) {
    // acc[ni][mi][8jj + q] = channel ni*32 + jj*16 + h*8 + q of pixel r in tile row 4w + mi
    f32x16 acc[2][CR];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < CR; ++mi)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[ni][mi][q] = 0.f;
    int b, oy0, ox0, n0;
    unit_coords(ord, b, oy0, ox0, n0);
    const int oyw = oy0 + wave * CR;

    unsigned long long ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int i) {
      if (DBG & 16) asm volatile("s_memtime %0" : "=s"(ts[i]) : : "memory");
    };
    auto chunk = [&](int c) {
      stamp(0);
      if ((DBG & 16) && ord == RG_PROBE_UNIT + 1 && (c == 2 || c == nc - 1)) {     // undisturbed chunk period: two entry stamps
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0)
          (reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res2)) +
           (((long long)blockIdx.x * 8 + wave) * 4 + (c == 2 ? 0 : 1)) * 10)[9] = ts[0];
      }
      // my fragment reads of batch g-1 have returned (its stage is refilled behind the barrier)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(1);
      __builtin_amdgcn_s_barrier();
      stamp(2);
      const unsigned char* st = smem + (g % RG_NS) * STAGE;
      const unsigned char* sW = st + w_off;
      f16x8_t fa[2][CR], fb[2][2];                     // [parity][row / channel block]: fragments are read one tap ahead
      auto mma_tap = [&](int par) {
#pragma unroll
        for (int mi = 0; mi < CR; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            if (DBG & 1) acc[ni][mi][0] += (float)fa[par][mi][0] * (float)fb[par][ni][0];
            else if (DBG & 4) {      // clock experiment: the 16x16x32 shape at equal FLOPs and fragment reads (NOT the convolution)
              f32x4 lo = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
              f32x4 hi = {acc[ni][mi][4], acc[ni][mi][5], acc[ni][mi][6], acc[ni][mi][7]};
              lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[par][ni], fa[par][mi], lo, 0, 0, 0);
              hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[par][ni], fa[par][mi], hi, 0, 0, 0);
#pragma unroll
              for (int q = 0; q < 4; ++q) { acc[ni][mi][q] = lo[q]; acc[ni][mi][4 + q] = hi[q]; }
            } else acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[par][ni], fa[par][mi], acc[ni][mi], 0, 0, 0);
          }
      };
      if (SPARSE) {
        auto load_frags = [&](int j, int par) {      // j = dy*2 + dx inside the window = slab slot
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) fb[par][ni] = *reinterpret_cast<const f16x8_t*>(sW + (j * 2 * 64 + ni * 32) * 16);
#pragma unroll
          for (int mi = 0; mi < CR; ++mi) fa[par][mi] = *reinterpret_cast<const f16x8_t*>(smem + f_cur[mi + (j >> 1)][j & 1]);
        };
        load_frags(0, 0);
        const int win_n = smem[L::WINTAB + (c + 1 == nc ? 0 : c + 1)];
        stamp(3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < 3) load_frags(j + 1, (j & 1) ^ 1);
          __builtin_amdgcn_sched_barrier(0);
          mma_tap(j & 1);
          __builtin_amdgcn_sched_barrier(0);
          stamp(4 + j);
        }
        frag_addresses(__builtin_amdgcn_readfirstlane(win_n), ((g + 1) % RG_NS) * STAGE);
      } else {
        auto load_frags = [&](int t, int par) {
          const int dy = t / 3, dx = t - dy * 3;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) fb[par][ni] = *reinterpret_cast<const f16x8_t*>(sW + (t * 2 * 64 + ni * 32) * 16);
#pragma unroll
          for (int mi = 0; mi < CR; ++mi) fa[par][mi] = *reinterpret_cast<const f16x8_t*>(st + p_off[SPARSE ? 0 : (mi + dy) * 3 + dx]);
        };
        load_frags(0, 0);
        stamp(3);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if (t < 8) load_frags(t + 1, (t & 1) ^ 1);
          __builtin_amdgcn_sched_barrier(0);
          mma_tap(t & 1);
          __builtin_amdgcn_sched_barrier(0);
          if (t == 1 || t == 3 || t == 5 || t == 8) stamp(t == 8 ? 7 : 4 + (t >> 1));
        }
      }
      if (DBG & 16) {
        stamp(8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ord == RG_PROBE_UNIT && c >= 8 && c < 12 && lane == 0) {
          unsigned long long* cb = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res2)) +
                                   (((long long)blockIdx.x * 8 + wave) * 4 + (c - 8)) * 10;
#pragma unroll
          for (int i = 0; i < 9; ++i) cb[i] = ts[i];
        }
      }
    };

    // residual values of one tile row: [ni][jj][half] x 4 channels = 32 contiguous bytes per (ni, jj)
    const bool has_res = a.res1 != nullptr && !(DBG & 8);
    auto row_ok = [&](int mi) { return oyw + mi < H && ox0 + r < W; };
    auto load_res_row = [&](int mi, f32x4 (&rv)[2][2][2]) {
#pragma unroll
      for (int i = 0; i < 8; ++i) (&rv[0][0][0])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!has_res || !row_ok(mi)) return;
      const float* rp = a.res1 + ((long long)(b * H + oyw + mi) * W + ox0 + r) * a.ldr1 + n0 + h * 8;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          if (n0 + ni * 32 + jj * 16 + h * 8 < a.Cout) {
            rv[ni][jj][0] = *reinterpret_cast<const f32x4*>(rp + ni * 32 + jj * 16);
            rv[ni][jj][1] = *reinterpret_cast<const f32x4*>(rp + ni * 32 + jj * 16 + 4);
          }
    };

    for (int c = 0; c < nc - 1; ++c, ++g) chunk(c);
    // last chunk of the tile (peeled): the first row's residual values are requested in front of its MFMAs
    f32x4 rva[2][2][2], rvb[2][2][2];
    load_res_row(0, rva);
    chunk(nc - 1);
    ++g;

    if (DBG & 8) {
      float t = 0.f;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < CR; ++mi)
#pragma unroll
          for (int q = 0; q < 16; ++q) t += acc[ni][mi][q] + rva[ni & 1][mi & 1][0][q & 3];
      if (t == 123.456f) a.out[0] = t;
      continue;
    }
    // ---- epilogue, one tile row at a time; row mi + 1's residual values are requested before row mi is processed
    auto epilogue_row = [&](int mi, const f32x4 (&rv)[2][2][2]) {
      const int oy = oyw + mi, X = ox0 + r;
      const bool ok = row_ok(mi);
      const long long pix = (long long)(b * H + oy) * W + X;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int n = n0 + ni * 32 + jj * 16 + h * 8;
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(smem + L::BIAS + n * 4);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(smem + L::BIAS + n * 4 + 16);
          f32x4 v0, v1;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float t0 = acc[ni][mi][8 * jj + k] + b0[k], t1 = acc[ni][mi][8 * jj + 4 + k] + b1[k];
            v0[k] = fmaxf(t0, slope * t0);
            v1[k] = fmaxf(t1, slope * t1);
          }
          v0 += rv[ni][jj][0];
          v1 += rv[ni][jj][1];
          if (!ok || n >= a.Cout) continue;
          if (a.res2 && !(DBG & 16)) {
            const float* p2 = a.res2 + pix * a.ldr2 + n;
            v0 += *reinterpret_cast<const f32x4*>(p2);
            v1 += *reinterpret_cast<const f32x4*>(p2 + 4);
          }
          if (a.res_up2) {      // + bilinear x2 of a half-resolution tensor: taps (Q-1, Q) x (P-1, P), clamped
            const float ly = (oy & 1) ? 0.25f : 0.75f, lx = (X & 1) ? 0.25f : 0.75f;
            if (SPARSE) {       // from the staged tile: local tap rows Q - oy0/2 (+1), columns P - ox0/2 (+1)
              const int li = ((oy + 1) >> 1) - (oy0 >> 1), lj = ((X + 1) >> 1) - (ox0 >> 1);
              const unsigned char* et = smem + L::ETILE;
              const int cq = (ni * 32 + jj * 16 + h * 8) >> 2;
#pragma unroll
              for (int hf = 0; hf < 2; ++hf) {
                auto tap = [&](int i, int j) {
                  const int px = i * RG_ET_COLS + j;
                  return *reinterpret_cast<const f32x4*>(et + px * 256 + (((cq + hf) ^ (px & 15)) << 4));
                };
                (hf ? v1 : v0) += (1.f - ly) * ((1.f - lx) * tap(li, lj) + lx * tap(li, lj + 1)) +
                                  ly * ((1.f - lx) * tap(li + 1, lj) + lx * tap(li + 1, lj + 1));
              }
            } else {
              const int Hd = H >> 1, Wd = W >> 1;
              const int Q = (oy + 1) >> 1, P = (X + 1) >> 1;
              const int ya = Q > 0 ? Q - 1 : 0, yb = Q < Hd ? Q : Hd - 1, xa = P > 0 ? P - 1 : 0, xb = P < Wd ? P : Wd - 1;
              const float* eb = a.res_up2 + (long long)b * Hd * Wd * a.ldru + n;
              const float* paa = eb + ((long long)ya * Wd + xa) * a.ldru;
              const float* pab = eb + ((long long)ya * Wd + xb) * a.ldru;
              const float* pba = eb + ((long long)yb * Wd + xa) * a.ldru;
              const float* pbb = eb + ((long long)yb * Wd + xb) * a.ldru;
#pragma unroll
              for (int hf = 0; hf < 2; ++hf) {
                const f32x4 eaa = *reinterpret_cast<const f32x4*>(paa + 4 * hf), eab = *reinterpret_cast<const f32x4*>(pab + 4 * hf);
                const f32x4 eba = *reinterpret_cast<const f32x4*>(pba + 4 * hf), ebb = *reinterpret_cast<const f32x4*>(pbb + 4 * hf);
                (hf ? v1 : v0) += (1.f - ly) * ((1.f - lx) * eaa + lx * eab) + ly * ((1.f - lx) * eba + lx * ebb);
              }
            }
          }
          f16x8_t hv;
#pragma unroll
          for (int k = 0; k < 4; ++k) { hv[k] = (_Float16)v0[k]; hv[4 + k] = (_Float16)v1[k]; }
          if (a.out_f16) {
            *reinterpret_cast<f16x8_t*>(reinterpret_cast<_Float16*>(a.out) + pix * a.ldo + n) = hv;
          } else {
            *reinterpret_cast<f32x4*>(a.out + pix * a.ldo + n) = v0;
            *reinterpret_cast<f32x4*>(a.out + pix * a.ldo + n + 4) = v1;
          }
          if (a.out2_cp16) {
            const int npl = a.out2_lo ? (a.Cout >> 3) : (a.Cout >> 4);
            _Float16* o2 = static_cast<_Float16*>(a.out2_cp16) + (((long long)b * npl + (n >> 4)) * H * W + (long long)oy * W + X) * 16 + h * 8;
            *reinterpret_cast<f16x8_t*>(o2) = hv;
            if (a.out2_lo) {
              f16x8_t lv;
#pragma unroll
              for (int k = 0; k < 4; ++k) { lv[k] = (_Float16)(v0[k] - (float)hv[k]); lv[4 + k] = (_Float16)(v1[k] - (float)hv[4 + k]); }
              *reinterpret_cast<f16x8_t*>(o2 + (long long)(a.Cout >> 4) * H * W * 16) = lv;
            }
          }
        }
    };
    unsigned long long te0 = 0, te1 = 0;
    if (DBG & 16) asm volatile("s_memtime %0" : "=s"(te0) : : "memory");
    load_res_row(1, rvb);
    epilogue_row(0, rva);
    load_res_row(2, rva);
    epilogue_row(1, rvb);
    load_res_row(3, rvb);
    epilogue_row(2, rva);
    epilogue_row(3, rvb);
    if (DBG & 16) {      // epilogue length of the undisturbed tile: slots [chunk 2][9], [chunk 3][9]
      asm volatile("s_memtime %0" : "=s"(te1) : : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (ord == RG_PROBE_UNIT + 1 && lane == 0) {
        unsigned long long* cb = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res2)) + ((long long)blockIdx.x * 8 + wave) * 40;
        cb[29] = te0;
        cb[39] = te1;
      }
    }
  }
}

int rg_num_cus() { return cdfo_num_cus(); }

// CDFO_RING_SPLIT=0 selects the round-2 form (eight identical waves); default: the wave-specialised form
bool rg_split() {
  static const bool v = [] { const char* s = getenv("CDFO_RING_SPLIT"); return !(s && s[0] == '0'); }();
  return v;
}

template <bool SPARSE, int DBG>
int rg_launch(const cdfo_conv_args& a, const ring_extra& e, int grid, hipStream_t st) {
  static CdfoAttrOnce once, once_split;
  // The wave-specialised form runs the four-tap convolutions only (same-box op table, 8 x 272x480: 0.801 -> 0.711 ms per call,
  // 1.8 ms per forward).  The dense form is NOT faster with it (256 -> 64: 0.354 -> 0.366 ms; the short-K split-fp16
  // convolutions 192 -> 64 and 48 -> 64 on 56 frames +12 % / +17 %: a tile of 3-16 chunks ends in an epilogue that four
  // consumer waves take through 4 rows each, one wave per SIMD), so it keeps the eight-identical-waves kernel.
  // (with a half-resolution residual the producers spread its 48 tile pieces over a tile's first six chunks: nc >= 8)
  if constexpr (SPARSE) {
    if (rg_split() && !(a.res_up2 && e.nc < 8)) {
      static const bool mf16 = [] { const char* v = getenv("CDFO_RING_MFMA16"); return !(v && v[0] == '0'); }();
      if (mf16 && !(DBG & (4 | 16))) {
        static CdfoAttrOnce once16;
        const hipError_t err = cdfo_set_max_lds(once16, reinterpret_cast<const void*>(conv3x3_ring_split_kernel<SPARSE, DBG, true>), RingLds<SPARSE>::TOTAL);
        if (err != hipSuccess) return (int)err;
        hipLaunchKernelGGL((conv3x3_ring_split_kernel<SPARSE, DBG, true>), dim3(grid), dim3(RG_THREADS), RingLds<SPARSE>::TOTAL, st, a, e);
        return 0;
      }
      const hipError_t err = cdfo_set_max_lds(once_split, reinterpret_cast<const void*>(conv3x3_ring_split_kernel<SPARSE, DBG>), RingLds<SPARSE>::TOTAL);
      if (err != hipSuccess) return (int)err;
      hipLaunchKernelGGL((conv3x3_ring_split_kernel<SPARSE, DBG>), dim3(grid), dim3(RG_THREADS), RingLds<SPARSE>::TOTAL, st, a, e);
      return 0;
    }
  }
  const hipError_t err = cdfo_set_max_lds(once, reinterpret_cast<const void*>(conv3x3_ring_kernel<SPARSE, DBG>), RingLds<SPARSE>::TOTAL);
  if (err != hipSuccess) return (int)err;
  hipLaunchKernelGGL((conv3x3_ring_kernel<SPARSE, DBG>), dim3(grid), dim3(RG_THREADS), RingLds<SPARSE>::TOTAL, st, a, e);
  return 0;
}

}  // namespace

extern "C" int cdfo_conv3x3_ring(const cdfo_conv_args* pa, void* stream) {
  const cdfo_conv_args& a = *pa;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a.nsrc != 1 || a.B <= 0 || a.ks != 3 || a.stride != 1 || a.pad != 1 || !a.src_f16) return CDFO_EINVAL;
  if (a.act == CDFO_ACT_SIGMOID || a.Cin <= 0 || a.Cin % 16 || a.cs[0] != a.Cin || a.ld[0] != 16) return CDFO_EINVAL;
  if (a.CoutP % 64 || a.CoutP > 1024 || a.CoutP < a.Cout || a.Cout <= 0 || a.Ho != a.H || a.Wo != a.W || a.w_bstride != 0) return CDFO_EINVAL;
  if (a.store_mode != CDFO_STORE_PLAIN || a.res2_pixscale) return CDFO_EINVAL;
  if (a.Cout % 8) return CDFO_EINVAL;
  if (a.out2_cp16 && (a.Cout % 16 || !aligned16(a.out2_cp16))) return CDFO_EINVAL;
  if (a.res_up2 && ((a.H | a.W) & 1 || a.ldru % 4 || a.ldru < a.Cout || !aligned16(a.res_up2))) return CDFO_EINVAL;
  if (!aligned16(a.src[0]) || !aligned16(a.w) || a.Cout % 4 || a.ldo % 4 || !aligned16(a.out) ||
      (a.bias && !aligned16(a.bias)))
    return CDFO_EALIGN;
  if ((a.res1 && (a.ldr1 % 4 || !aligned16(a.res1))) || (a.res2 && (a.ldr2 % 4 || !aligned16(a.res2)))) return CDFO_EALIGN;
  const int nc = a.Cin / 16;
  if ((long long)nc * a.H * a.W * 32 >= (1ll << 31) || (long long)a.B * a.H * a.W >= (1ll << 31)) return CDFO_EINVAL;
  const int taps = a.tap_mask ? 4 : 9;
  const long long w_bytes = (long long)nc * taps * 2 * a.CoutP * 16;
  if (w_bytes >= (1ll << 31)) return CDFO_EINVAL;
  const int cus = rg_num_cus();
  if (cus < 8) return CDFO_EINVAL;
  ring_extra e;
  e.src = a.src[0]; e.nc = nc; e.w = a.w; e.CoutP = a.CoutP; e.tap_mask = a.tap_mask; e.w_bytes = (int)w_bytes;
  e.plane_wrap = a.src_plane_wrap;
  e.halfsplit = a.src_halfsplit ? 1 : 0;
  // (off by default: the touched lines do not survive until the epilogue -- FETCH_SIZE of the launch rose by exactly the res1
  // tile bytes, 3.10 -> 3.44 GB, i.e. the epilogue fetched them again -- and the epilogue got 7 % shorter at best; CDFO_RING_TOUCH=1)
  static const int touch = [] { const char* v = getenv("CDFO_RING_TOUCH"); return (v && v[0] == '1') ? 1 : 0; }();
  e.touch = (touch && a.res1 && (long long)a.H * a.W * a.ldr1 * 4 < (1ll << 31)) ? 1 : 0;
  e.src_planes = a.src_plane_wrap ? a.src_plane_wrap : nc;
  if (a.src_plane_wrap < 0 || a.src_plane_wrap > nc) return CDFO_EINVAL;
  const int grid = cus / 8 * 8;
  const double px = (double)a.B * a.Ho * a.Wo;
  CdfoProfScope prof(st, a.tap_mask ? KID_CONV3_RING4 : KID_CONV3_RING, 2.0 * px * a.Cout * a.Cin * taps,
                     2.0 * px * a.Cin + (a.out_f16 ? 2.0 : 4.0) * px * a.Cout + 2.0 * taps * a.Cin * a.Cout);
  int rc;
  switch (a.prec >> 8) {
    case 0: rc = a.tap_mask ? rg_launch<true, 0>(a, e, grid, st) : rg_launch<false, 0>(a, e, grid, st); break;
    case 1: rc = a.tap_mask ? rg_launch<true, 1>(a, e, grid, st) : rg_launch<false, 1>(a, e, grid, st); break;
    case 2: rc = a.tap_mask ? rg_launch<true, 2>(a, e, grid, st) : rg_launch<false, 2>(a, e, grid, st); break;
    case 4: rc = a.tap_mask ? rg_launch<true, 4>(a, e, grid, st) : rg_launch<false, 4>(a, e, grid, st); break;
    case 8: rc = a.tap_mask ? rg_launch<true, 8>(a, e, grid, st) : rg_launch<false, 8>(a, e, grid, st); break;
    case 9: rc = a.tap_mask ? rg_launch<true, 9>(a, e, grid, st) : rg_launch<false, 9>(a, e, grid, st); break;
    case 10: rc = a.tap_mask ? rg_launch<true, 10>(a, e, grid, st) : rg_launch<false, 10>(a, e, grid, st); break;
    case 16: rc = a.tap_mask ? rg_launch<true, 16>(a, e, grid, st) : rg_launch<false, 16>(a, e, grid, st); break;
    default: return CDFO_EINVAL;
  }
  if (rc) return rc;
  CDFO_LAUNCH_CHECK();
  return 0;
}
