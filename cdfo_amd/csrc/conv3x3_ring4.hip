// The four-tap ("sparse") form of the LDS-DMA ring convolution (conv3x3_ring.hip) on TALL tiles: 32 rows x 32 pixels per unit
// where the image has them, 16 x 32 for what is left, in ONE persistent launch.
//
// Block_'s double-resolution branch ends in conv2 . mean2x2 . down.0, composed into a 3x3 convolution over the 1024 space-to-
// depth channels of body[0]'s output with four active taps per 16-channel chunk (arch/SIDECVSR_our.py:383-406; cvsr_v8.py
// _weights).  The 16 x 32 form moves 28 KiB through the LDS-DMA engine per chunk and tile (20 KiB of activations for an 18 x 34
// halo, 8 KiB of weights) for 16 MFMAs per wave: 3.66 GB per launch at 8 x 272 x 480, DMA-bound (0.61 ms of DMA alone, 0.57 ms of
// MFMAs alone, 0.89 ms together: DESIGN 5.1).  A 32-row tile shares the chunk's weight slab between twice the pixels and
// overlaps two halo rows less often: 45 KiB per chunk for 32 MFMAs per wave = 0.78x the bytes per pixel, 0.75 LDS fragment
// reads per MFMA instead of 1, half the barriers per MFMA.  What kept round 2 from building it is the work granularity (8 clips
// of 272 x 480 are 4.2 rounds of 1024-pixel units on 256 CUs): here the 16 leftover rows of a 272-row image become 16 x 32 units
// of the same launch, handed first to the workgroups that received one 32 x 32 unit fewer.
//
// Per unit and chunk: 37 activation pieces (34 x 34 pixels x 32 B) + 8 weight pieces of 1 KiB, six per wave, issued one at a
// time between the taps' MFMAs; a three-stage ring (two chunks in flight), one workgroup barrier per chunk; wave w owns tile
// rows RW w .. RW w + RW - 1 (RW = 4 or 2).  The epilogue works from the accumulators (bias, + res1, + bilinear x2 of the
// half-resolution residual read from global memory, fp32 pixel-major store, optional fp16 chunk-planar copy / hi | lo planes).
// DMA completion is hand-counted exactly as in conv3x3_ring.hip (the rules sit next to the waits).
#include "common.h"

namespace {

constexpr int R4_THREADS = 512, R4_IW = 34, R4_PPW = 6, R4_NS = 3;
constexpr int R4_ACT_MAX = 37 * 1024, R4_WGT = 8 * 1024, R4_STAGE = R4_ACT_MAX + R4_WGT;      // 46,080 bytes per stage
constexpr int R4_DUMP = R4_NS * R4_STAGE, R4_BIAS = R4_DUMP + 1024, R4_LDS = R4_BIAS + 4096;  // 143,360

typedef _Float16 r4_f16x8 __attribute__((ext_vector_type(8)));
typedef int r4_i32x4 __attribute__((ext_vector_type(4)));

struct r4_args {
  const void* src; int nc;          // fp16 chunk-planar [B][nc][H][W][16]
  const void* w; int CoutP; int w_bytes;      // fp16 [nc][4][2][CoutP][8]: the chunk's four active taps, ascending
  const unsigned* tap_mask;
  const float* bias; int Cout;
  float* out; int ldo;
  const float* res1; int ldr1;
  const float* res_up2; int ldru;
  void* out2_cp16; int out2_lo;
  int B, H, W;
  int rows_big;                     // image rows covered by 32-row tiles (a multiple of 32), the rest by 16-row tiles
};

__device__ __forceinline__ void r4_dma1(unsigned voff, r4_i32x4 rsrc, unsigned soff, unsigned lds) {
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

__device__ __forceinline__ r4_i32x4 r4_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  r4_i32x4 r;
  r[0] = (int)(unsigned)p;
  r[1] = (int)(unsigned)(p >> 32);
  r[2] = (int)bytes;
  r[3] = 0x00020000;
  return r;
}

// All units of one tile height assigned to this workgroup: RW rows per wave, tile = 8 RW rows x 32 pixels.
// unit u (0 .. nunits-1) = (image, tile, 64-channel output block), block fastest; tiles cover image rows [row0, row0 + 8 RW tiles_y).
template <int RW>
__device__ __forceinline__ void r4_run(const r4_args& a, unsigned char* smem, int row0, int tiles_y, int first, int stride, int nunits) {
  constexpr int TH = 8 * RW, NPIX = (TH + 2) * R4_IW, ACT = (NPIX * 32 + 1023) / 1024;      // 37 pieces (RW 4) / 20 (RW 2)
  constexpr int NPIECE = ACT + 8;
  static_assert(NPIECE <= 8 * R4_PPW && ACT * 1024 <= R4_ACT_MAX, "piece budget");
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, nc = a.nc;
  const int tiles_x = (W + 31) >> 5, tiles = tiles_x * tiles_y, nco = a.CoutP >> 6;
  const int my_units = nunits > first ? (nunits - first + stride - 1) / stride : 0;
  if (my_units == 0) return;
  const unsigned lds0 = (unsigned)(unsigned long long)(smem);

  // ---- per-lane DMA descriptors: piece q = R4_PPW wave + j; q < ACT: activation slots 64 q + lane -> pixel p = s >> 1 of the
  // (TH + 2) x 34 halo, k-half (s & 1) ^ ((p >> 3) & 1); ACT <= q < ACT + 8: weight slab row q - ACT (tap*2 + k-half), lane =
  // output channel position; the rest: padding pieces into the dump kilobyte (every wave issues exactly R4_PPW per chunk)
  int d_iy[R4_PPW], d_ix[R4_PPW], d_rel[R4_PPW], d_kind[R4_PPW];      // kind 0 activation, 1 weight, 2 padding
  unsigned dst_off[R4_PPW];
#pragma unroll
  for (int j = 0; j < R4_PPW; ++j) {
    const int q = wave * R4_PPW + j;
    if (q < ACT) {
      const int s = q * 64 + lane, p = s >> 1, half = (s & 1) ^ ((p >> 3) & 1);
      const int iy = p / R4_IW, ix = p - iy * R4_IW;
      d_kind[j] = 0;
      d_iy[j] = p < NPIX ? iy : 1 << 20;
      d_ix[j] = ix;
      d_rel[j] = (iy * W + ix) * 32 + half * 16;
      dst_off[j] = (unsigned)q * 1024u;
    } else if (q < NPIECE) {
      const int wq = q - ACT, m = lane & 31;
      // LDS position `lane` of a slab row = MFMA row m of 32-channel block (lane >> 5); it holds output channel
      // (m>>4)*16 + ((m>>2)&1)*8 + ((m>>3)&1)*4 + (m&3): a lane's accumulators are then 8 consecutive channels of a 16-channel chunk
      const int chan = (lane & 32) + ((m >> 4) & 1) * 16 + ((m >> 2) & 1) * 8 + ((m >> 3) & 1) * 4 + (m & 3);
      d_kind[j] = 1;
      d_iy[j] = 0; d_ix[j] = 0;
      d_rel[j] = (wq * a.CoutP + chan) * 16;
      dst_off[j] = (unsigned)(R4_ACT_MAX + wq * 1024);
    } else {
      d_kind[j] = 2; d_iy[j] = 1 << 20; d_ix[j] = 0; d_rel[j] = 0;
      dst_off[j] = 0xffffffffu;
    }
  }
  const unsigned plane = (unsigned)(H * W) * 32u;                 // bytes of one 16-channel plane
  const unsigned wchunk = (unsigned)(4 * 2 * a.CoutP) * 16u;      // bytes of one chunk's weight slab (all output blocks)
  const unsigned img_bytes = plane * (unsigned)nc;
  const r4_i32x4 rsrc_w = r4_rsrc(a.w, (unsigned)a.w_bytes);

  const int w_off = R4_ACT_MAX + (h * 64 + r) * 16;

  auto unit_coords = [&](int ord, int& b, int& oy0, int& ox0, int& n0) {
    const int uidx = first + ord * stride;
    const int nb = uidx % nco, t = uidx / nco;
    const int tile = t % tiles;
    b = t / tiles;
    const int ty = tile / tiles_x;
    oy0 = row0 + ty * TH; ox0 = (tile - ty * tiles_x) * 32; n0 = nb * 64;
  };

  // ---- issue cursor: (unit, chunk) of the next DMA batch; a batch is issued piece by piece between the MFMAs
  int iu = 0, ic = 0, gi = 0;
  unsigned voff[R4_PPW];
  r4_i32x4 rsrc_a = rsrc_w;
  unsigned soff_a = 0, soff_w = 0, stage_i = 0;
  auto issue_begin = [&]() {
    if (ic == 0) {                    // descriptors of unit `iu`
      int b, oy0, ox0, n0;
      unit_coords(iu, b, oy0, ox0, n0);
      rsrc_a = r4_rsrc(static_cast<const unsigned char*>(a.src) + (unsigned long long)b * img_bytes, img_bytes);
      const int base = ((oy0 - 1) * W + (ox0 - 1)) * 32;
#pragma unroll
      for (int j = 0; j < R4_PPW; ++j) {
        if (d_kind[j] == 0) {
          const int gy = oy0 - 1 + d_iy[j], gx = ox0 - 1 + d_ix[j];
          const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
          voff[j] = ok ? (unsigned)(base + d_rel[j]) : 0x80000000u;      // out of range => the DMA writes zeros
        } else {
          voff[j] = d_kind[j] == 1 ? (unsigned)(d_rel[j] + n0 * 16) : 0x80000000u;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) rsrc_a[k] = __builtin_amdgcn_readfirstlane(rsrc_a[k]);
    soff_a = __builtin_amdgcn_readfirstlane((unsigned)ic * plane);
    soff_w = __builtin_amdgcn_readfirstlane((unsigned)ic * wchunk);
    stage_i = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((gi % R4_NS) * R4_STAGE));
  };
  auto issue_piece = [&](int j) {     // j is a compile-time constant at every call site; the piece's kind is wave-uniform
    const int q = wave * R4_PPW + j;
    const bool is_w = q >= ACT;
    const unsigned dst = __builtin_amdgcn_readfirstlane(q < NPIECE ? stage_i + dst_off[j] : lds0 + (unsigned)R4_DUMP);
    r4_i32x4 rs;
#pragma unroll
    for (int k = 0; k < 4; ++k) rs[k] = __builtin_amdgcn_readfirstlane(is_w ? rsrc_w[k] : rsrc_a[k]);
    r4_dma1(voff[j], rs, __builtin_amdgcn_readfirstlane(is_w ? soff_w : soff_a), dst);
  };
  auto issue_end = [&]() {
    ++gi;
    if (++ic == nc) { ic = 0; ++iu; }
  };
  const int total = my_units * nc;    // chunk batches of this workgroup
#pragma unroll
  for (int k = 0; k < R4_NS - 1; ++k)
    if (k < total) {
      issue_begin();
#pragma unroll
      for (int j = 0; j < R4_PPW; ++j) issue_piece(j);
      issue_end();
    }

  int g = 0;                          // batch being consumed
  for (int ord = 0; ord < my_units; ++ord) {
    // transposed product: M = output channels (weights = A operand), N = pixels -> acc[ni][mi][8jj + q] is channel
    // ni*32 + jj*16 + h*8 + q of pixel r in tile row RW w + mi
    f32x16 acc[2][RW];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < RW; ++mi)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[ni][mi][q] = 0.f;
    int b, oy0, ox0, n0;
    unit_coords(ord, b, oy0, ox0, n0);

    for (int c = 0; c < nc; ++c, ++g) {
      // (1) my pieces of batch g have landed: the younger DMA batch is R4_PPW instructions and DMA pieces retire in issue order
      //     among themselves; other vector-memory operations (epilogue stores / loads) are NOT assumed ordered against them, so
      //     "at most R4_PPW outstanding" is the bound that implies batch g has landed whatever else is in flight.
      // (2) my fragment reads of batch g-1 have RETURNED (its stage is overwritten after the barrier).  Then the barrier:
      //     everyone's pieces of g are in LDS, nobody still reads the stage of g-1.
      if (g + R4_NS - 2 >= total) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      static_assert((R4_NS - 2) * R4_PPW == 6, "counted wait immediate");
      __builtin_amdgcn_s_barrier();
      const bool do_issue = g + R4_NS - 1 < total;     // batch g+2 -> the stage batch g-1 lived in
      if (do_issue) issue_begin();
      const unsigned char* st = smem + (g % R4_NS) * R4_STAGE;
      const unsigned char* sW = st + w_off;
      // the chunk's four taps are a 2x2 window of the 3x3 stencil (checked by the host wrapper): top-left (y0, x0)
      const unsigned tm = a.tap_mask[c];
      const int win = ((tm & 0x7u) ? 0 : 2) + ((tm & 0x49u) ? 0 : 1);
      // (no switch over the four windows: with four copies of the tap loop hipcc spills hundreds of accumulator registers; the
      // window origin enters through ten fragment addresses computed per chunk -- ~40 vector instructions against 32 MFMAs)
      const int y0 = win >> 1, x0 = win & 1;
      int f_off[RW + 1][2];
#pragma unroll
      for (int rr = 0; rr < RW + 1; ++rr)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int p = (wave * RW + rr + y0) * R4_IW + x0 + dx + r;
          f_off[rr][dx] = (2 * p + (h ^ ((p >> 3) & 1))) * 16;
        }
      {
        r4_f16x8 fa[2][RW], fb[2][2];                  // [parity][tile]: fragments are read one tap ahead
        auto load_frags = [&](int j, int par) {        // j = dy*2 + dx inside the window = slab slot
#pragma unroll
          for (int mi = 0; mi < RW; ++mi) fa[par][mi] = *reinterpret_cast<const r4_f16x8*>(st + f_off[mi + (j >> 1)][j & 1]);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) fb[par][ni] = *reinterpret_cast<const r4_f16x8*>(sW + (j * 2 * 64 + ni * 32) * 16);
        };
        load_frags(0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < 3) load_frags(j + 1, (j & 1) ^ 1);
          // (scheduling fences: keep the next tap's fragment reads in FRONT of this tap's MFMAs, as in conv3x3_ring.hip)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < RW; ++mi)
              acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j & 1][ni], fa[j & 1][mi], acc[ni][mi], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (do_issue) {                              // six pieces over four taps: 2, 2, 1, 1
            if (j == 0) { issue_piece(0); issue_piece(1); }
            if (j == 1) { issue_piece(2); issue_piece(3); }
            if (j == 2) issue_piece(4);
            if (j == 3) issue_piece(5);
          }
        }
      }
      if (do_issue) issue_end();
    }

    // ---- epilogue straight from the accumulators: + bias, + res1, + bilinear x2 of res_up2, stores (32 contiguous bytes per
    // lane and channel group; the fp16 chunk-planar copy: 16 bytes per lane, 1 KiB per wave).  Plain global loads / stores: they
    // run behind the next unit's first chunks (whose DMA batches are already in flight) and are drained by the counted waits.
    const int X = ox0 + r;
#pragma unroll
    for (int mi = 0; mi < RW; ++mi) {
      const int oy = oy0 + wave * RW + mi;
      const bool px_ok = oy < H && X < W;
      const long long pix = (long long)(b * H + oy) * W + X;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int n = n0 + ni * 32 + jj * 16 + h * 8;
          // (fence: without it hipcc hoists the residual loads of all 16 iterations to the top -- 160 live 16-byte registers,
          // hundreds of spills -- instead of keeping one iteration's ten in flight)
          __builtin_amdgcn_sched_barrier(0);
          if (!px_ok || n >= a.Cout) continue;
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(smem + R4_BIAS + n * 4);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(smem + R4_BIAS + n * 4 + 16);
          f32x4 v0, v1;
#pragma unroll
          for (int k = 0; k < 4; ++k) { v0[k] = acc[ni][mi][8 * jj + k] + b0[k]; v1[k] = acc[ni][mi][8 * jj + 4 + k] + b1[k]; }
          if (a.res1) {
            const float* p1 = a.res1 + pix * a.ldr1 + n;
            v0 += *reinterpret_cast<const f32x4*>(p1);
            v1 += *reinterpret_cast<const f32x4*>(p1 + 4);
          }
          if (a.res_up2) {      // + bilinear x2 of a half-resolution tensor: taps (Q-1, Q) x (P-1, P), clamped
            const float ly = (oy & 1) ? 0.25f : 0.75f, lx = (X & 1) ? 0.25f : 0.75f;
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
          *reinterpret_cast<f32x4*>(a.out + pix * a.ldo + n) = v0;
          *reinterpret_cast<f32x4*>(a.out + pix * a.ldo + n + 4) = v1;
          if (a.out2_cp16) { // chunk-planar fp16 copy: record (image, chunk n/16, pixel), halves h*8 .. h*8+7
            r4_f16x8 hv;
#pragma unroll
            for (int k = 0; k < 4; ++k) { hv[k] = (_Float16)v0[k]; hv[4 + k] = (_Float16)v1[k]; }
            const int npl = a.out2_lo ? (a.Cout >> 3) : (a.Cout >> 4);      // planes per image: hi | lo, or hi only
            _Float16* o2 = static_cast<_Float16*>(a.out2_cp16) + (((long long)b * npl + (n >> 4)) * H * W + (long long)oy * W + X) * 16 + h * 8;
            *reinterpret_cast<r4_f16x8*>(o2) = hv;
            if (a.out2_lo) {   // + the remainders: the pair is the source of a split-fp16 (hi + lo activations) convolution
              r4_f16x8 lv;
#pragma unroll
              for (int k = 0; k < 4; ++k) { lv[k] = (_Float16)(v0[k] - (float)hv[k]); lv[4 + k] = (_Float16)(v1[k] - (float)hv[4 + k]); }
              *reinterpret_cast<r4_f16x8*>(o2 + (long long)(a.Cout >> 4) * H * W * 16) = lv;
            }
          }
        }
    }
  }
  // every DMA piece of this run has been waited for (the last chunks wait vmcnt(0)); the stages may be re-primed by the next run
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

__global__ __launch_bounds__(R4_THREADS) void conv3x3_ring4_tall_kernel(r4_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  for (int i = tid; i < a.CoutP; i += R4_THREADS)
    reinterpret_cast<float*>(smem + R4_BIAS)[i] = (a.bias && i < a.Cout) ? a.bias[i] : 0.f;
  // (made visible to the other waves by the barriers of the chunk loop, long before the first epilogue)
  const int tiles_x = (a.W + 31) >> 5, nco = a.CoutP >> 6;
  const int ty_big = a.rows_big >> 5, ty_small = (a.H - a.rows_big + 15) >> 4;
  const int n_big = a.B * ty_big * tiles_x * nco, n_small = a.B * ty_small * tiles_x * nco;
  // each XCD (private L2) takes a contiguous band of the units of each kind; its workgroups stride through the band.  The 16-row
  // units of a band go first to the workgroups that got one 32-row unit fewer.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int band_b = (n_big + 7) >> 3, b0 = xcd * band_b, nb = min(band_b, n_big - b0);
  const int band_s = (n_small + 7) >> 3, s0 = xcd * band_s, ns = min(band_s, n_small - s0);
  if (nb > 0) r4_run<4>(a, smem, 0, ty_big, b0 + slot, nslots, b0 + nb);
  if (ns > 0) {
    const int rem = nb > 0 ? nb % nslots : 0;           // slots >= rem received one 32-row unit fewer (when rem > 0)
    const int sslot = (slot - rem + nslots) % nslots;
    r4_run<2>(a, smem, a.rows_big, ty_small, s0 + sslot, nslots, s0 + ns);
  }
}

}  // namespace

// Called by cdfo_conv3x3_ring (conv3x3_ring.hip) for its four-tap form.  Returns 1 when it launched, 0 when this form does not
// apply (the caller runs the 16 x 32 kernel), 2 + hipError_t when the launch failed.
int cdfo_conv3x3_ring4_tall(const cdfo_conv_args& a, hipStream_t st) {
  if (!a.tap_mask || a.res2 || a.act != CDFO_ACT_NONE || a.out_f16 || a.src_plane_wrap || a.H < 32) return 0;
  const int nc = a.Cin / 16;
  const long long w_bytes = (long long)nc * 4 * 2 * a.CoutP * 16;
  const int cus = cdfo_num_cus();
  if (cus < 8) return 0;
  static CdfoAttrOnce once;
  const hipError_t err = cdfo_set_max_lds(once, reinterpret_cast<const void*>(conv3x3_ring4_tall_kernel), R4_LDS);
  if (err != hipSuccess) return 2 + (int)err;
  r4_args r;
  r.src = a.src[0]; r.nc = nc; r.w = a.w; r.CoutP = a.CoutP; r.w_bytes = (int)w_bytes; r.tap_mask = reinterpret_cast<const unsigned*>(a.tap_mask);
  r.bias = a.bias; r.Cout = a.Cout; r.out = static_cast<float*>(a.out); r.ldo = a.ldo; r.res1 = a.res1; r.ldr1 = a.ldr1;
  r.res_up2 = a.res_up2; r.ldru = a.ldru; r.out2_cp16 = a.out2_cp16; r.out2_lo = a.out2_lo;
  r.B = a.B; r.H = a.H; r.W = a.W;
  // 32-row tiles while they fit; at most 16 rows are left to 16-row tiles (17-31 leftover rows take one more 32-row tile)
  int rows_big = a.H / 32 * 32;
  if (a.H - rows_big > 16) rows_big += 32;
  r.rows_big = rows_big > a.H ? (a.H + 31) / 32 * 32 : rows_big;
  hipLaunchKernelGGL(conv3x3_ring4_tall_kernel, dim3(cus / 8 * 8), dim3(R4_THREADS), R4_LDS, st, r);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 1 : 2 + (int)e;
}
