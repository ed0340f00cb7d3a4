// 1x1 convolution as a persistent HBM stream: LDS-DMA rings, no workgroup barrier in the steady state.
//
// conv1x1_bf16x3.hip (one 128-pixel tile per workgroup: load -> split -> LDS -> barrier -> MFMA -> LDS transpose -> store)
// reaches 2.2-3.4 TB/s of algorithmic traffic: its phases are separated by barriers, only two workgroups fit a CU, and
// HBM is idle whenever both are computing or storing.  The 1x1 convolutions of the CVSR_V8 forward (arch.py: attention
// apply of MDTA :1573-1575 and DualAttAlignment :3459-3491 as per-image folded weights, RDAB.input_conv / fuse
// :2196,2246, conv_du_re.0 :2148, fusion_out :3441, Block_.up.0 :381) move 4*(Cin+Cout) bytes per pixel for 2*Cin*Cout
// FLOP: pure streams.  Here
//   * one 256-thread workgroup per CU is persistent over a contiguous range of 128-pixel tiles; each of its four waves owns
//     32 pixels of the tile and a PRIVATE ring of 8 KB LDS stages that it fills itself by LDS-DMA
//     (buffer_load_dwordx4 ... lds): the fp32 pixel-major activations of one 64-channel K block (32 pixels x 256 B), and
//     after the last K block the tile's residual operands, arrive as 8 pieces of 1 KiB with NS-1 stages in flight behind a
//     counted s_waitcnt -- no staging registers, no barrier, the waves drift freely;
//   * the LDS image is [pixel][16-byte part ^ (pixel & 15)] (the swizzle is applied on the DMA's per-lane source
//     address), so the fragment reads -- a lane's 8 consecutive input channels of its pixel, two ds_read_b128 -- are
//     conflict-free; the lane splits them to bf16 hi / lo in registers and feeds the MFMA directly (no LDS write pass);
//   * same arithmetic as conv1x1_bf16x3: split-bf16, a_lo*w_hi + a_hi*w_lo + a_hi*w_hi, fp32 accumulation (fp32-grade);
//     the weights (also the per-image folded attention weights) are split once per image into LDS, rows permuted so that
//     a lane's accumulators are 8 consecutive output channels of its pixel (transposed product, as conv3x3_ws.hip): the
//     epilogue (bias, activation, up to two residuals read from their LDS stages) stores 32 contiguous bytes per lane
//     straight from registers.
// Contract: cdfo_conv_args as cdfo_conv1x1_bf16x3 with plain store, Cin % 64 == 0 per source, Cout % 8 == 0,
// CoutP in {64, 128}, weights + ring within the CU's LDS (Cin * CoutP <= 192 * 64); everything else stays on that kernel.
#include "common.h"

namespace {

constexpr int ST_THREADS = 256;
constexpr int ST_STAGE = 8192;                 // 32 pixels x 64 channels x 4 B
constexpr int ST_MAXNS = 4;

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct st_item { const float* base; int ld; int ch0; };     // one 64-channel slice of a pixel-major operand

struct st_args {
  st_item act[CDFO_MAXSRC * 4];    // K blocks in order (<= 12 used: Cin <= 768 guarded by the LDS test anyway)
  int nkb;
  st_item res[2];                  // residual operands (ch0 = 0), res[i].base == nullptr: absent
  const float* res_scale;          // optional per-pixel factor [B][P] of the LAST residual operand (a spatial gate folded into the add)
  const float* w; long long w_bstride; const float* bias;
  int Cin, Cout, CoutP, act_fn;
  float* out; int ldo;
  int B; long long P;              // pixels per image
  int tiles_per_image; long long tiles;
  int ns;                          // ring stages per wave
  // optional second output (Cout == 64): LayerNorm64(result) as fp16 hi | lo chunk-planar planes [B][8][P][16] -- the source of
  // the split-fp16 3x3 convolution that follows the attention in the feature extractor (arch.py:1470-1474): the values are in
  // the epilogue's registers, a separate LayerNorm pass would read the 64-channel result back from HBM
  _Float16* ln_hl; const float* ln_g; const float* ln_b;
  // TAPS form (upconv2 of arch.py:4474-4476, Cout = 256 = 4 sub-pixels x 64): instead of the pixel-shuffled 64-channel HR map, out
  // receives for every HR pixel the nine per-tap channel sums of the 3x3 conv_last that follows (see conv1x1_bf16x3.hip, TAPS):
  // out[HR pixel][ldo], HR pixel = (b, 2 y + (cb >> 1), 2 x + (cb & 1)) of the W-pixel-wide input row (y, x)
  const float* w_last;             // conv_last.weight as [64][9]
  int W;                           // input image width (pixels per row) for the pixel shuffle
};

__device__ __forceinline__ unsigned st_pack_bf16(float a, float b) {
  const __bf16 ha = (__bf16)a, hb = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
}
__device__ __forceinline__ float st_bf16_round(float a) { return (float)(__bf16)a; }

// eight 1 KiB pieces of one stage: lane l of piece k writes LDS bytes lds + 1024 k + 16 l from (buffer base + voff[k])
__device__ __forceinline__ void st_dma8(const unsigned (&voff)[8], i32x4 rsrc, unsigned lds) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %10\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %1, %9, 0 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %2, %9, 0 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %3, %9, 0 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %4, %9, 0 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %5, %9, 0 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %6, %9, 0 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %7, %9, 0 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %8, %9, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "v"(voff[5]), "v"(voff[6]), "v"(voff[7]),
        "s"(rsrc), "s"(lds)
      : "memory", "scc");
}

// one dword per lane: lane l writes LDS bytes lds + 4 l from (buffer base + voff)
__device__ __forceinline__ void st_dma1(unsigned voff, i32x4 rsrc, unsigned lds) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "buffer_load_dword %1, %2, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds)
      : "memory", "scc");
}

// two fp32 -> packed fp16 hi and packed fp16 lo = fp16(v - hi): v_cvt_pk_f16_f32, two v_fma_mix_f32 (exact remainders, the fp16
// operand read from its half of the packed register), v_cvt_pk_f16_f32 (see attention.hip: full-register results only)
__device__ __forceinline__ void st_split_pair_f16(float x, float y, unsigned& hi, unsigned& lo) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 hv = {(_Float16)x, (_Float16)y};
  hi = __builtin_bit_cast(unsigned, hv);
  float rx, ry;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(rx) : "v"(x), "v"(hi));
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(ry) : "v"(y), "v"(hi));
  const h2 lv = {(_Float16)rx, (_Float16)ry};
  lo = __builtin_bit_cast(unsigned, lv);
}

// NCB: 64-wide output-channel blocks (1, 2; 4 in the TAPS form)
template <int NCB, bool TAPS = false>
__global__ __launch_bounds__(ST_THREADS) void conv1x1_stream_kernel(st_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nkb = a.nkb, NS = a.ns;
  // LDS map: weights hi [nkb*4 chunks][2 k-halves][NCB*64 rows][8 bf16] | weights lo (same) | bias [NCB*64 floats] |
  //          4 waves x NS stages
  const int w_half = nkb * 4 * 2 * NCB * 64 * 16;
  unsigned char* sWh = smem;
  unsigned char* sWl = smem + w_half;
  float* sBias = reinterpret_cast<float*>(smem + 2 * w_half);
  const int ring_off = 2 * w_half + NCB * 64 * 4;
  unsigned char* ring = smem + ring_off + wave * NS * ST_STAGE;
  const unsigned ring_lds = (unsigned)(unsigned long long)(smem) + ring_off + wave * NS * ST_STAGE;
  // res_scale: two 256-byte slots per wave (tile parity) behind the rings
  const unsigned char* scale_slots = smem + ring_off + 4 * NS * ST_STAGE + wave * 512;
  const unsigned scale_lds = (unsigned)(unsigned long long)(smem) + ring_off + 4 * NS * ST_STAGE + wave * 512;

  // MFMA row m of a 32-channel block holds channel (m>>4)*16 + ((m>>2)&1)*8 + ((m>>3)&1)*4 + (m&3) (see conv3x3_ws.hip):
  // a lane's accumulator registers then are 8 consecutive channels of each 16-channel group
  auto chan_of_row = [](int n) { const int m = n & 31; return (n & ~31) + ((m >> 4) & 1) * 16 + ((m >> 2) & 1) * 8 + ((m >> 3) & 1) * 4 + (m & 3); };

  // this workgroup's contiguous tile range
  const long long t_lo = a.tiles * blockIdx.x / gridDim.x, t_hi = a.tiles * (blockIdx.x + 1) / gridDim.x;
  if (t_lo >= t_hi) return;
  const int nres = (a.res[0].base ? 1 : 0) + (a.res[1].base ? 1 : 0);
  const int items_per_tile = nkb + nres * NCB;
  const float slope = a.act_fn == CDFO_ACT_NONE ? 1.f : (a.act_fn == CDFO_ACT_LRELU ? 0.1f : 0.f);

  // per-lane DMA slot geometry: piece k, lane l -> slot s = 64 k + l -> pixel p = s >> 4 (0..31), part (s & 15) ^ (p & 15)
  int d_px[8], d_part[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int s = 64 * k + lane, p = s >> 4;
    d_px[k] = p;
    d_part[k] = (s & 15) ^ (p & 15);
  }
  // fragment read offsets of this lane's pixel r: 16-byte part q lives at r*256 + ((q ^ (r & 15)) << 4)
  auto part_off = [&](int q) { return r * 256 + ((q ^ (r & 15)) << 4); };

  // TAPS: A operand of the second product = w_last^T (row = tap, rows 9..31 zero), fp16 hi | lo, in registers for the whole launch:
  // lane (r = tap, h) holds channels 16 s4 + 8 h .. + 7
  typedef _Float16 st_h8 __attribute__((ext_vector_type(8)));
  typedef unsigned st_u4 __attribute__((ext_vector_type(4)));
  st_h8 twh[TAPS ? 4 : 1], twl[TAPS ? 4 : 1];
  if (TAPS) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float wv = r < 9 ? a.w_last[(16 * s4 + 8 * h + j) * 9 + r] : 0.f;
        twh[s4][j] = (_Float16)wv;
        twl[s4][j] = (_Float16)(wv - (float)twh[s4][j]);
      }
  }

  for (long long img_t0 = t_lo; img_t0 < t_hi;) {
    const int b = (int)(img_t0 / a.tiles_per_image);
    const long long img_end = (long long)(b + 1) * a.tiles_per_image;
    const long long seg_hi = img_end < t_hi ? img_end : t_hi;           // tiles [img_t0, seg_hi) belong to image b
    // ---- this image's weights -> LDS (split bf16 hi | lo, rows permuted).  No DMA is in flight here.
    __syncthreads();                                                     // previous image's MFMAs are done with sW
    {
      const float* wb = a.w + (long long)b * a.w_bstride;
      const int ngrp = (a.Cin >> 2) * NCB * 64;                          // (group of 4 input channels, row position)
      for (int i = tid; i < ngrp; i += ST_THREADS) {
        const int rowpos = i % (NCB * 64), kg = i / (NCB * 64);
        const int n = chan_of_row(rowpos);
        f32x4 wv = {0.f, 0.f, 0.f, 0.f};
        if (n < a.CoutP) wv = *reinterpret_cast<const f32x4*>(wb + ((long long)kg * a.CoutP + n) * 4);
        const int c = kg >> 2, hh = (kg >> 1) & 1, j0 = (kg & 1) * 4;
        u32x2 hi, lo;
        hi[0] = st_pack_bf16(wv[0], wv[1]); hi[1] = st_pack_bf16(wv[2], wv[3]);
        lo[0] = st_pack_bf16(wv[0] - st_bf16_round(wv[0]), wv[1] - st_bf16_round(wv[1]));
        lo[1] = st_pack_bf16(wv[2] - st_bf16_round(wv[2]), wv[3] - st_bf16_round(wv[3]));
        const int off = ((c * 2 + hh) * (NCB * 64) + rowpos) * 16 + j0 * 2;
        *reinterpret_cast<u32x2*>(sWh + off) = hi;
        *reinterpret_cast<u32x2*>(sWl + off) = lo;
      }
      for (int i = tid; i < NCB * 64; i += ST_THREADS) sBias[i] = (a.bias && i < a.Cout) ? a.bias[i] : 0.f;   // by channel
    }
    __syncthreads();

    // ---- the item stream of this wave for tiles [img_t0, seg_hi): per tile nkb K blocks, then the residual stages
    const long long n_items = (seg_hi - img_t0) * items_per_tile;
    long long issued = 0;
    auto issue = [&](long long it) {
      const long long tile = img_t0 + it / items_per_tile;
      const int k = (int)(it % items_per_tile);
      const long long p0 = (tile - (long long)b * a.tiles_per_image) * 128 + wave * 32;      // first pixel (inside image b)
      if (a.res_scale && k == 0) {
        // the tile's per-pixel residual factors: one more LDS-DMA piece (lane l -> dword l of the tile-parity slot), requested AHEAD
        // of the tile's first item.  An ordinary load would be waited for with vmcnt(0) at its use and drain the ring; memory
        // operations retire in order, so the slot is valid once any item of this tile has landed (the hand-counted waits below,
        // which one more older piece only makes more conservative).
        const long long rows = a.P - p0;
        const unsigned long long pb = reinterpret_cast<unsigned long long>(a.res_scale + (long long)b * a.P + (rows > 0 ? p0 : 0));
        i32x4 rs;
        rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)pb);
        rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(pb >> 32));
        rs[2] = __builtin_amdgcn_readfirstlane((int)(rows > 0 ? (rows < 32 ? rows : 32) * 4 : 0));      // beyond the image: zeros
        rs[3] = 0x00020000;
        st_dma1((unsigned)r * 4, rs, __builtin_amdgcn_readfirstlane(scale_lds + (unsigned)(tile & 1) * 256));
      }
      st_item src;
      int chan0;
      if (k < nkb) { src = a.act[k]; chan0 = src.ch0; }
      else {                     // residual operands are packed from slot 0 by the host wrapper
        const int j = k - nkb;
        src = a.res[j / NCB];
        chan0 = (j % NCB) * 64;
      }
      // descriptor based at the wave's first pixel: offsets stay small whatever the tensor's size
      const long long rows_left = a.P - p0;                               // may be <= 0: everything out of range
      const float* base = src.base + ((long long)b * a.P + (rows_left > 0 ? p0 : 0)) * src.ld + chan0;
      const unsigned long long pb = reinterpret_cast<unsigned long long>(base);
      i32x4 rsrc;
      rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)pb);
      rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(pb >> 32));
      const long long bytes = rows_left > 0 ? (rows_left < 32 ? rows_left : 32) * (long long)src.ld * 4 : 0;
      rsrc[2] = __builtin_amdgcn_readfirstlane((int)bytes);               // lanes beyond the image read zeros
      rsrc[3] = 0x00020000;
      unsigned voff[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) voff[q] = (unsigned)(d_px[q] * src.ld * 4 + d_part[q] * 16);
      st_dma8(voff, rsrc, __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)(it % NS) * ST_STAGE));
    };
    for (; issued < NS - 1 && issued < n_items; ++issued) issue(issued);

    f32x16 acc[NCB][2];
    for (long long it = 0; it < n_items; ++it) {
      const int k = (int)(it % items_per_tile);
      // item `it` has landed when at most the 8 * (younger items in flight) youngest DMA pieces are outstanding
      // (pieces retire in order among themselves; stores in flight only make the wait more conservative)
      const long long younger = TAPS ? 0 : issued - it - 1;
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned char* st = ring + (it % NS) * ST_STAGE;
      if (TAPS) {
        // two-stage ring (the 64 KB weight image leaves room for no more): the stage of item it - 1 was consumed in the previous
        // trip, so item it + 1 is requested NOW and lands behind this tile's products and tap epilogue
        if (issued < n_items) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); issue(issued); ++issued; }
      }
      if (k == 0) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[cb][ni][e] = 0.f;
      }
      if (k < nkb) {
        // ---- one K block: 4 chunks of 16 channels; this lane's operand = channels c*16 + h*8 .. +7 of pixel r
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(st + part_off(c * 4 + h * 2));
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(st + part_off(c * 4 + h * 2 + 1));
          union { unsigned u[4]; bf16x8_t v; } ph, pl;
          ph.u[0] = st_pack_bf16(v0[0], v0[1]); ph.u[1] = st_pack_bf16(v0[2], v0[3]);
          ph.u[2] = st_pack_bf16(v1[0], v1[1]); ph.u[3] = st_pack_bf16(v1[2], v1[3]);
          pl.u[0] = st_pack_bf16(v0[0] - st_bf16_round(v0[0]), v0[1] - st_bf16_round(v0[1]));
          pl.u[1] = st_pack_bf16(v0[2] - st_bf16_round(v0[2]), v0[3] - st_bf16_round(v0[3]));
          pl.u[2] = st_pack_bf16(v1[0] - st_bf16_round(v1[0]), v1[1] - st_bf16_round(v1[1]));
          pl.u[3] = st_pack_bf16(v1[2] - st_bf16_round(v1[2]), v1[3] - st_bf16_round(v1[3]));
          const int wrow = ((k * 4 + c) * 2 + h) * (NCB * 64) + r;
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              const bf16x8_t wh = *reinterpret_cast<const bf16x8_t*>(sWh + (wrow + cb * 64 + ni * 32) * 16);
              const bf16x8_t wl = *reinterpret_cast<const bf16x8_t*>(sWl + (wrow + cb * 64 + ni * 32) * 16);
              acc[cb][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, pl.v, acc[cb][ni], 0, 0, 0);
              acc[cb][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ph.v, acc[cb][ni], 0, 0, 0);
              acc[cb][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ph.v, acc[cb][ni], 0, 0, 0);
            }
        }
        if (k == nkb - 1) {      // bias + activation, in place: acc[cb][ni][8 jj + q] = channel cb*64 + ni*32 + jj*16 + h*8 + q
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const float t = acc[cb][ni][e] + sBias[cb * 64 + ni * 32 + (e >> 3) * 16 + h * 8 + (e & 7)];
                acc[cb][ni][e] = fmaxf(t, 0.f) + slope * fminf(t, 0.f);
              }
        }
      } else {
        // ---- a residual stage: 64 channels of block cb of the tile's residual operand
        const int cb = (k - nkb) % NCB;
        float sc = 1.f;
        if (a.res_scale && (k - nkb) / NCB == nres - 1)
          sc = *reinterpret_cast<const float*>(scale_slots + ((img_t0 + it / items_per_tile) & 1) * 256 + lane * 4);
#pragma unroll
        for (int cc = 0; cc < NCB; ++cc) {
          if (cc != cb) continue;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              const int q = (ni * 32 + jj * 16 + h * 8) >> 2;
              const f32x4 r0 = *reinterpret_cast<const f32x4*>(st + part_off(q));
              const f32x4 r1 = *reinterpret_cast<const f32x4*>(st + part_off(q + 1));
#pragma unroll
              for (int e = 0; e < 4; ++e) { acc[cc][ni][8 * jj + e] = fmaf(sc, r0[e], acc[cc][ni][8 * jj + e]); acc[cc][ni][8 * jj + 4 + e] = fmaf(sc, r1[e], acc[cc][ni][8 * jj + 4 + e]); }
            }
        }
      }
      // the stage may be overwritten once its reads have returned: the next DMA below targets the stage of item it-1... wait
      // for this item's own LDS reads too (they were consumed by the MFMAs / adds above, so they have returned)
      if (!TAPS && issued < n_items) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); issue(issued); ++issued; }
      if (k == items_per_tile - 1) {
        // ---- store: 32 contiguous bytes per lane and 16-channel group
        const long long tile = img_t0 + it / items_per_tile;
        const long long pin = (tile - (long long)b * a.tiles_per_image) * 128 + wave * 32 + r;
        if (NCB == 1 && a.ln_hl && !a.ln_g) {
          // plain fp16 chunk-planar copy [B][4][P][16] of the result (the source layout of the weights-stationary 3x3 kernel)
          typedef _Float16 st_f16x8 __attribute__((ext_vector_type(8)));
          if (pin < a.P) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int jj = 0; jj < 2; ++jj) {
                const int n = ni * 32 + jj * 16 + h * 8;
                st_f16x8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (_Float16)acc[0][ni][8 * jj + e];
                *reinterpret_cast<st_f16x8*>(a.ln_hl + (((long long)b * 4 + (n >> 4)) * a.P + pin) * 16 + h * 8) = hv;
              }
          }
        } else if (NCB == 1 && a.ln_hl) {
          // per-pixel LayerNorm over the 64 channels this lane (32 of them) and lane ^ 32 hold; biased variance, eps 1e-5
          typedef _Float16 st_f16x8 __attribute__((ext_vector_type(8)));
          float sm = 0.f;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) sm += acc[0][ni][e];
          sm += __shfl_xor(sm, 32, 64);
          const float mu = sm * (1.f / 64.f);
          float sq = 0.f;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float d = acc[0][ni][e] - mu; sq = fmaf(d, d, sq); }
          sq += __shfl_xor(sq, 32, 64);
          const float rstd = 1.f / sqrtf(sq * (1.f / 64.f) + 1e-5f);
          if (pin < a.P) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int jj = 0; jj < 2; ++jj) {
                const int n = ni * 32 + jj * 16 + h * 8;               // 8 consecutive channels = half of chunk n >> 4
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln_g + n), g1 = *reinterpret_cast<const f32x4*>(a.ln_g + n + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.ln_b + n), b1 = *reinterpret_cast<const f32x4*>(a.ln_b + n + 4);
                st_f16x8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const float y = (acc[0][ni][8 * jj + e] - mu) * rstd * (e < 4 ? g0[e & 3] : g1[e & 3]) + (e < 4 ? b0[e & 3] : b1[e & 3]);
                  hi[e] = (_Float16)y;
                  lo[e] = (_Float16)(y - (float)hi[e]);
                }
                _Float16* o16 = a.ln_hl + (((long long)b * 8 + (n >> 4)) * a.P + pin) * 16 + h * 8;
                *reinterpret_cast<st_f16x8*>(o16) = hi;
                *reinterpret_cast<st_f16x8*>(o16 + 4 * a.P * 16) = lo;
              }
          }
        }
        if (TAPS) {
          // Block cb = sub-pixel (cb >> 1, cb & 1) of the pixel-shuffled map.  acc[cb][ni][8 jj + q] = act(y)[channel 16 (2 ni + jj) +
          // 8 h + q] of this lane's pixel: exactly the B fragment (k = 8 h + q of chunk 2 ni + jj) of taps[k][pixel] = sum_c
          // w_last[c][k] act(y)[pixel][c] -- the accumulators feed the second product straight from registers.  Scaled per PIXEL by a
          // power of two into fp16's range (a column of the product may carry its own scale), split hi | lo: hi*hi + lo*hi + hi*lo.
          const int oy = (int)(pin / a.W), ox = (int)(pin - (long long)oy * a.W);
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) {
            float amax = 0.f;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int e = 0; e < 16; ++e) amax = fmaxf(amax, fabsf(acc[cb][ni][e]));
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));              // the pixel's other 32 channels sit in lane ^ 32
            int ex = 0;
            if (amax > 0.f && amax < INFINITY) frexpf(amax, &ex);
            ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
            const float sc = ldexpf(1.f, 14 - ex), inv = ldexpf(1.f, ex - 14);
            f32x16 tacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) tacc[e] = 0.f;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              unsigned hh[4], ll[4];
#pragma unroll
              for (int j = 0; j < 4; ++j)
                st_split_pair_f16(acc[cb][s4 >> 1][8 * (s4 & 1) + 2 * j] * sc, acc[cb][s4 >> 1][8 * (s4 & 1) + 2 * j + 1] * sc, hh[j], ll[j]);
              const st_u4 hu = {hh[0], hh[1], hh[2], hh[3]}, lu = {ll[0], ll[1], ll[2], ll[3]};
              const st_h8 yh = __builtin_bit_cast(st_h8, hu), yl = __builtin_bit_cast(st_h8, lu);
              tacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(twl[s4], yh, tacc, 0, 0, 0);
              tacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(twh[s4], yl, tacc, 0, 0, 0);
              tacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(twh[s4], yh, tacc, 0, 0, 0);
            }
            // lane (pixel r, h): registers 0..3 = taps 4 h .. 4 h + 3, register 4 = tap 8 (h = 0)
            if (pin < a.P) {
              const long long opix = ((long long)b * 2 * (a.P / a.W) + 2 * oy + (cb >> 1)) * (2 * a.W) + 2 * ox + (cb & 1);
              float* op = a.out + opix * a.ldo;
              const f32x4 t4 = {tacc[0] * inv, tacc[1] * inv, tacc[2] * inv, tacc[3] * inv};
              *reinterpret_cast<f32x4*>(op + 4 * h) = t4;
              if (h == 0) op[8] = tacc[4] * inv;
            }
          }
        } else if (pin < a.P) {
          float* op = a.out + ((long long)b * a.P + pin) * a.ldo;
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int jj = 0; jj < 2; ++jj) {
                const int n = cb * 64 + ni * 32 + jj * 16 + h * 8;
                if (n < a.Cout) {
                  f32x4 v0, v1;
#pragma unroll
                  for (int e = 0; e < 4; ++e) { v0[e] = acc[cb][ni][8 * jj + e]; v1[e] = acc[cb][ni][8 * jj + 4 + e]; }
                  *reinterpret_cast<f32x4*>(op + n) = v0;
                  *reinterpret_cast<f32x4*>(op + n + 4) = v1;
                }
              }
        }
      }
    }
    img_t0 = seg_hi;
  }
}

}  // namespace

// Returns 1 when the streaming kernel took the launch, 0 when the shapes are outside its contract (the caller falls back to
// cdfo_conv1x1_bf16x3), < 0 / hipError_t on errors.  Same argument block as cdfo_conv1x1_bf16x3.
int cdfo_conv1x1_stream_try(const cdfo_conv_args& a, hipStream_t st) {
  const bool ln_out = a.out2_cp16 != nullptr;      // post-LayerNorm hi | lo second output (checked by the caller: Cout == 64)
  const bool taps = a.store_mode == CDFO_STORE_TAPS9;
  if (taps) {
    // upconv2 + conv_last's tap sums (checked by the caller: Cout = CoutP = 256, res2 = conv_last.weight [64][9], no res1)
    if (a.Cin != 64 || a.nsrc != 1 || a.ldo % 4 || a.ldo < 12 || a.ln_gamma || ln_out || (long long)a.ld[0] * 4 * 32 >= (1ll << 31)) return 0;
    st_args s{};
    s.act[0].base = a.src[0]; s.act[0].ld = a.ld[0]; s.act[0].ch0 = 0;
    s.nkb = 1;
    s.w = a.w; s.w_bstride = a.w_bstride; s.bias = a.bias;
    s.Cin = 64; s.Cout = 256; s.CoutP = 256; s.act_fn = a.act;
    s.out = a.out; s.ldo = a.ldo; s.B = a.B; s.P = (long long)a.H * a.W;
    s.w_last = a.res2; s.W = a.W;
    s.tiles_per_image = (int)((s.P + 127) / 128);
    s.tiles = (long long)a.B * s.tiles_per_image;
    s.ns = 2;
    const int w_bytes = 2 * 1 * 4 * 2 * 4 * 64 * 16 + 4 * 64 * 4;
    const int lds = w_bytes + 4 * s.ns * ST_STAGE;
    const int cus = cdfo_num_cus();
    if (cus <= 0) return CDFO_EINVAL;
    const int grid = (int)(s.tiles < cus ? s.tiles : cus);
    const double px = (double)a.B * s.P;
    CdfoProfScope prof(st, KID_CONV1, 2.0 * px * 256 * 64 + 2.0 * px * 4 * 9 * 64, 4.0 * (px * 64 + px * 4 * a.ldo + 64.0 * 256));
    static CdfoAttrOnce once;
    const hipError_t e = cdfo_set_max_lds(once, reinterpret_cast<const void*>(conv1x1_stream_kernel<4, true>), 160 * 1024 - 256);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv1x1_stream_kernel<4, true>), dim3(grid), dim3(ST_THREADS), lds, st, s);
    const hipError_t e2 = hipGetLastError();
    return e2 == hipSuccess ? 1 : (int)e2;
  }
  if (a.store_mode != CDFO_STORE_PLAIN || (a.ln_gamma && !ln_out) || a.CoutP % 64 || a.CoutP > 128 || a.Cout % 8) return 0;      // (ln_out without ln_gamma: plain fp16 copy)
  if (ln_out && (a.CoutP != 64 || a.Cout != 64)) return 0;
  const int ncb = a.CoutP / 64, nkb = a.Cin / 64;
  if (nkb < 1 || nkb > CDFO_MAXSRC * 4) return 0;
  const int w_bytes = 2 * nkb * 4 * 2 * ncb * 64 * 16 + ncb * 64 * 4;
  const int scale_bytes = (a.res2 && a.res2_pixscale) ? 4 * 512 : 0;      // two tile-parity slots per wave
  int ns = (160 * 1024 - 256 - scale_bytes - w_bytes) / (4 * ST_STAGE);
  if (ns > ST_MAXNS) ns = ST_MAXNS;
  if (scale_bytes) {
    // the per-pixel factors of tile t + 2 are requested with item (t + 2, 0), NS - 1 items ahead of consumption, into the slot tile t
    // used (two tile-parity slots): safe only while NS <= items_per_tile + 2, or tile t's residual stage would read tile t + 2's factors
    const int items_per_tile = nkb + ((a.res1 ? 1 : 0) + 1) * ncb;      // as the kernel counts them: K blocks + residual stages
    if (ns > items_per_tile + 2) ns = items_per_tile + 2;
  }
  if (ns < 3) return 0;
  const long long P = (long long)a.H * a.W;
  st_args s{};
  int k = 0;
  for (int i = 0; i < a.nsrc; ++i) {
    if ((long long)a.ld[i] * 4 * 32 >= (1ll << 31)) return 0;
    for (int c = 0; c < a.cs[i]; c += 64) { s.act[k].base = a.src[i]; s.act[k].ld = a.ld[i]; s.act[k].ch0 = c; ++k; }
  }
  if (k != nkb) return 0;
  s.nkb = nkb;
  // residual operands are staged 64 channels at a time: they must be at least NCB*64 channels wide in memory terms only
  // where Cout needs it (a narrower tensor would be read past its pitch): require ld >= CoutP
  int nr = 0;
  if (a.res1) { if (a.ldr1 < a.CoutP) return 0; s.res[nr].base = a.res1; s.res[nr].ld = a.ldr1; ++nr; }
  if (a.res2) { if (a.ldr2 < a.CoutP) return 0; s.res[nr].base = a.res2; s.res[nr].ld = a.ldr2; ++nr; }
  s.res_scale = a.res2 ? a.res2_pixscale : nullptr;
  s.w = a.w; s.w_bstride = a.w_bstride; s.bias = a.bias;
  s.Cin = a.Cin; s.Cout = a.Cout; s.CoutP = a.CoutP; s.act_fn = a.act;
  s.out = a.out; s.ldo = a.ldo; s.B = a.B; s.P = P;
  s.ln_hl = ln_out ? static_cast<_Float16*>(a.out2_cp16) : nullptr; s.ln_g = a.ln_gamma; s.ln_b = a.ln_beta;
  s.tiles_per_image = (int)((P + 127) / 128);
  s.tiles = (long long)a.B * s.tiles_per_image;
  s.ns = ns;
  const int cus = cdfo_num_cus();
  if (cus <= 0) return CDFO_EINVAL;
  const int grid = (int)(s.tiles < cus ? s.tiles : cus);
  const int lds = w_bytes + 4 * ns * ST_STAGE + scale_bytes;
  const double px = (double)a.B * P;
  CdfoProfScope prof(st, KID_CONV1, 2.0 * px * a.Cout * a.Cin, 4.0 * (px * a.Cout * (1 + nr) + px * a.Cin + (double)a.Cin * a.Cout));
  if (ncb == 1) {
    static CdfoAttrOnce once;
    const hipError_t e = cdfo_set_max_lds(once, reinterpret_cast<const void*>(conv1x1_stream_kernel<1>), 160 * 1024 - 256);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv1x1_stream_kernel<1>, dim3(grid), dim3(ST_THREADS), lds, st, s);
  } else {
    static CdfoAttrOnce once;
    const hipError_t e = cdfo_set_max_lds(once, reinterpret_cast<const void*>(conv1x1_stream_kernel<2>), 160 * 1024 - 256);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv1x1_stream_kernel<2>, dim3(grid), dim3(ST_THREADS), lds, st, s);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 1 : (int)e;
}
