// 3x3 stride-1 convolution on the 16-bit matrix cores (v_mfma_f32_32x32x16_{bf16,f16}) with fp32 accumulation.
//
//   PREC_BF16X3 : split-bf16, three passes.  Every fp32 operand x is held as hi = bf16(x), lo = bf16(x - hi); a product is
//                 a_hi*w_hi + a_lo*w_hi + a_hi*w_lo (the lo*lo term, 2^-16 relative, is dropped).  Whole-forward max-abs
//                 error vs the fp32 reference ~1e-5: fp32-grade, at 16/3 of the exact-fp32 MFMA rate.
//   PREC_FP16X2 : activations split into fp16 hi + fp16 lo (22 significant bits), weights rounded ONCE to fp16 (11 bits):
//                 two passes a_hi*w + a_lo*w.  Whole-forward max-abs error 3-5e-4 (inside the 1e-3 parity bound with a
//                 2x margin), 2/3 of the MFMAs of bf16x3, half the weight bytes in LDS -> three workgroups per CU.
//   PREC_BF16   : plain bf16 operands, one pass -- BASELINE's "bf16" configuration; ~6e-3 max-abs on the forward.
//
// Structure (one 256-thread workgroup = 8 rows x 32 columns of output pixels x 64 output channels; a 32-pixel MFMA M tile
// is ONE image row, which with the 80-byte pixel record makes every ds_read_b128 lane group hit 16 distinct 16-byte
// slots):
//   per 16-input-channel chunk: the 10x34x16 input halo tile is fetched as fp32 float4s into REGISTERS while the
//   previous chunk's MFMAs run (register-staged double buffering), then converted to 16-bit hi/lo and written to LDS
//   (80-byte pixel records: 32 B hi | 32 B lo | 16 B pad); the pre-packed 16-bit weight slab [tap][k-half][cout][8] is
//   copied the same way.  Operand fragments are read one tap ahead of the MFMAs that use them.  The epilogue goes
//   through a wave-private LDS transpose (conv_epilogue.h).  Workgroups are ordered so that the output-channel blocks of
//   one input tile share an XCD's L2.  Optional tap_mask skips (chunk, tap) pairs whose weights are all zero (the
//   stride-2-composed convolution of Block_'s double-resolution branch).  No im2col buffer, no HBM intermediates.
//
// Replaces F.conv2d for the 3x3 convolutions with >= 64 output channels on the CVSR_V8 path
// (arch/SIDECVSR_our.py:383-387 Block_.body -- 89 % of the forward's FLOPs --, :435, :1447, :261-262, :4382).
//
// Measured and rejected this round (all correct, none faster; numbers in profiles/r01_pmc_conv3x3.txt): LDS double
// buffering with one workgroup per CU and the next chunk's staging interleaved into the MFMA stream; the same as a
// strip-persistent (tile, chunk) stream with two-chunk-deep prefetch (compiler waits, then inline-asm loads with
// hand-counted vmcnt); a wave-specialised producer/consumer workgroup.  The ablation of the last one isolates the cause:
// the bare ds_read + MFMA stream of one wave per SIMD already takes 1.9x its 2.4 GHz issue time (the chip holds a lower
// clock under the matrix load, MI355X_MICROARCH.md "DVFS give-back"), and the 4.3 GB of epilogue stores add their
// full HBM time unless another workgroup's MFMAs cover them.
#include "common.h"
#include "conv_epilogue.h"

constexpr int PIXB_F32SRC = 80;   // LDS bytes per staged pixel when converting from fp32: 32 B hi | 32 B lo | 16 B pad

namespace {

constexpr int TH = 8, TW = 32, IW = 34, NPIX = 34 * 10;
constexpr int W_HALF = 9 * 2 * 64 * 16;          // 18,432 B: [tap][h][64 cout][8 x 16-bit]

enum { M_BF16X3 = CDFO_PREC_BF16X3, M_BF16 = CDFO_PREC_BF16, M_FP16X2 = CDFO_PREC_FP16X2, M_FP16IN = CDFO_PREC_FP16,
       M_FP16X1 = CDFO_PREC_FP16X1 };

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE> struct Fmt;
template <> struct Fmt<M_BF16X3> { using frag = bf16x8_t; static constexpr bool ALO = true, WLO = true; };
template <> struct Fmt<M_BF16> { using frag = bf16x8_t; static constexpr bool ALO = false, WLO = false; };
template <> struct Fmt<M_FP16X2> { using frag = f16x8_t; static constexpr bool ALO = true, WLO = false; };
// M_FP16IN: the source tensor already holds fp16 values -> one pass, staging is a plain copy, 48-byte pixel records
template <> struct Fmt<M_FP16IN> { using frag = f16x8_t; static constexpr bool ALO = false, WLO = false; };
// M_FP16X1: fp32 source rounded once to fp16 while staging, one pass
template <> struct Fmt<M_FP16X1> { using frag = f16x8_t; static constexpr bool ALO = false, WLO = false; };

__device__ __forceinline__ f32x16 mma(bf16x8_t a, bf16x8_t b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma(f16x8_t a, f16x8_t b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <bool F16>
__device__ __forceinline__ unsigned pack16(float a, float b) {
  if (F16) {
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
  }
  const __bf16 ha = (__bf16)a, hb = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
}
template <bool F16>
__device__ __forceinline__ float round16(float a) { return F16 ? (float)(_Float16)a : (float)(__bf16)a; }

// CDFO_STORE_OFFMASK: the epilogue of MVDualAttAlignment's conv_offset[2] = the module's offset / mask assembly (arch.py:3336-3350),
// straight from the accumulators into the DCN operator's NCHW planes.  A lane of the 32x32 accumulator holds ONE output channel
// and four runs of four consecutive pixels of an image row: each run is one 16-byte access to that channel's plane, no transpose.
// tanh / sigmoid through v_exp_f32 + v_rcp_f32 (absolute error ~1e-7, i.e. 1e-6 px on a 10 px offset).
__device__ __forceinline__ float fast_tanh(float x) {     // 1 - 2 / (1 + e^(2x)); saturates to +-1 through inf / 0
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * 2.885390081777927f));
}
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
template <int NT>
struct OffmaskLane {             // what a lane of the 32x32 accumulators owns in the offset / mask planes
  float* plane[NT];
  const float* prior[NT];        // what a run starts from: the flipped motion field (first head) or the plane itself (second head)
  float bias[NT];
  bool live[NT], is_off[NT];
  f32x4 pv[NT][2][4];            // the prior values of the lane's 16 runs
};

// Requests every read of the epilogue.  Called ahead of the stores (they may alias the loads as far as the compiler knows: a
// load -> tanh -> store loop would pay one memory latency per run) -- in the single-pass mode ahead of the LAST chunk's MFMAs,
// which cover the latency of the second head's 1.8 GB read-modify-write.
template <int NT>
__device__ __forceinline__ void offmask_prefetch(const cdfo_conv_args& a, OffmaskLane<NT>& L, int lane, int b, int oyb, int ox0, int n0) {
  const int h = lane >> 5, r = lane & 31;
  const int third = a.Cout / 3;
  const long long P = (long long)a.Ho * a.Wo;
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    const int c = n0 + ni * 32 + r;
    L.live[ni] = c < a.Cout;
    L.is_off[ni] = c < 2 * third;
    L.bias[ni] = (a.bias && L.live[ni]) ? a.bias[c] : 0.f;
    L.plane[ni] = L.is_off[ni] ? a.out + ((long long)b * 2 * third + c) * P : a.mask_out + ((long long)b * third + (c - 2 * third)) * P;
    L.prior[ni] = a.off_accumulate ? L.plane[ni]
                                   : (L.is_off[ni] ? a.flow + b * a.flow_bstride + (long long)(1 - (c & 1)) * P : nullptr);   // flip(1)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int oy = oyb + mi, x = ox0 + 8 * j + 4 * h;
        L.pv[ni][mi][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (L.live[ni] && L.prior[ni] && oy < a.Ho && x < a.Wo) L.pv[ni][mi][j] = *reinterpret_cast<const f32x4*>(L.prior[ni] + oy * a.Wo + x);
      }
  }
}

template <int NT>
__device__ __forceinline__ void offmask_finish(const cdfo_conv_args& a, const OffmaskLane<NT>& L, const f32x16 (*acc)[NT], int lane,
                                               int oyb, int ox0) {
  const int h = lane >> 5;
#pragma unroll
  for (int ni = 0; ni < NT; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int oy = oyb + mi, x = ox0 + 8 * j + 4 * h;      // Wo % 4 == 0: a run of four pixels is inside the row or outside it
        if (!L.live[ni] || oy >= a.Ho || x >= a.Wo) continue;
        f32x4 res = L.pv[ni][mi][j];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v = acc[mi][ni][4 * j + k] + L.bias[ni];
          if (L.is_off[ni]) res[k] += a.off_mag * fast_tanh(v);
          else res[k] = a.off_accumulate ? fast_sigmoid(res[k] + v) : v;
        }
        *reinterpret_cast<f32x4*>(L.plane[ni] + oy * a.Wo + x) = res;
      }
}

// DBG (developer ablations, never used by the product path): 1 = skip the MFMAs, 2 = skip the per-chunk global loads,
// 4 = skip the per-chunk convert + LDS writes, 8 = skip the epilogue (store one value), 16 = skip the barriers
// NT: 32-channel output tiles per workgroup (2 = 64 output channels; 1 for convolutions with <= 32 output channels, e.g.
// UDSA's 64 -> 16: half the MFMAs and half the weight-fragment reads of the padded 64-wide tile)
template <int MODE, int WAVES_PER_SIMD, int DBG, int NT = 2>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void conv3x3_mma16_kernel(cdfo_conv_args a) {
  using F = Fmt<MODE>;
  using frag_t = typename F::frag;
  constexpr bool F16 = MODE == M_FP16X2 || MODE == M_FP16IN || MODE == M_FP16X1;
  constexpr bool SRC16 = MODE == M_FP16IN;
  constexpr int PIXB = (SRC16 || MODE == M_FP16X1) ? 48 : ::PIXB_F32SRC;      // LDS bytes per staged pixel (odd multiple of 16: conflict-free)
  constexpr int A_BYTES = NPIX * PIXB;
  constexpr int NA = SRC16 ? (NPIX * 2 + 255) / 256 : (NPIX * 4 + 255) / 256;   // 16-byte loads per thread per chunk
  constexpr int LDS_W = F::WLO ? 2 * W_HALF : W_HALF;
  constexpr int NWU = LDS_W / 16;                 // 16-byte units in the weight slab
  constexpr int NWS = (NWU + 255) / 256;          // per-thread weight loads per chunk (9 or 5)
  constexpr int EPI = ConvEpi<NT>::BLOCK_BYTES;
  constexpr int SMEM = (A_BYTES + LDS_W) > EPI ? (A_BYTES + LDS_W) : EPI;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  unsigned char* sA = smem;
  unsigned char* sW = smem + A_BYTES;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, r = lane & 31;
  // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs (private L2 each), so the nco
  // output-channel blocks that read the SAME input tile are given consecutive slots on ONE XCD (speed only).
  const int tiles_x = (a.Wo + TW - 1) / TW, tiles = tiles_x * ((a.Ho + TH - 1) / TH);
  const int nco = a.CoutP / 64;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pt = (slot / nco) * 8 + xcd;            // pixel tile (over all images)
  if (pt >= tiles * a.B) return;                    // padding blocks of the last group
  const int b = pt / tiles, tile = pt - b * tiles;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int oy0 = ty * TH, ox0 = tx * TW, n0 = (slot % nco) * 64;
  const unsigned short* wq = reinterpret_cast<const unsigned short*>(a.w);
  const long long lo_off = (long long)(a.Cin / 16) * 18 * a.CoutP * 8;  // elements from the hi block to the lo block

  // ---- per-thread staging descriptors (chunk independent)
  int a_pix[NA];   // global pixel index or -1 (outside the image => conv zero padding)
  int a_lds[NA];   // LDS byte offset of the hi half
#pragma unroll
  for (int s = 0; s < NA; ++s) {
    const int idx = tid + 256 * s;
    constexpr int PER = SRC16 ? 2 : 4;             // 16-byte loads per (pixel, chunk)
    const int p = idx / PER, q = idx % PER;
    const int iy = p / IW, ix = p - iy * IW;
    const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
    const bool ok = idx < NPIX * PER && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    a_pix[s] = ok ? (b * a.H + gy) * a.W + gx : -1;
    a_lds[s] = idx < NPIX * PER ? p * PIXB + q * (SRC16 ? 16 : 8) : -1;
  }

  f32x16 acc[2][NT];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  int a_off[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) a_off[mi] = ((wave * 2 + mi) * IW + r) * PIXB + h * 16;
  const int b_off = (h * 64 + r) * 16;

  f32x4 ra[NA];
  u32x4 rw[NWS];
  const int nchunks = a.Cin / 16;
  int s_idx = 0, s_base = 0;

  auto issue_loads = [&](int c) {
    const int ch0 = c * 16;
    while (ch0 >= s_base + a.cs[s_idx]) { s_base += a.cs[s_idx]; ++s_idx; }
    const int ld = a.ld[s_idx];
    if (SRC16) {       // fp16 source: 16 channels = 32 B = two 16-byte pieces per pixel
      const _Float16* src = reinterpret_cast<const _Float16*>(a.src[s_idx]) + (ch0 - s_base);
#pragma unroll
      for (int s = 0; s < NA; ++s) {
        const int q = (tid + 256 * s) & 1;
        ra[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a_pix[s] >= 0) ra[s] = *reinterpret_cast<const f32x4*>(src + (long long)a_pix[s] * ld + q * 8);
      }
    } else {
      const float* src = a.src[s_idx] + (ch0 - s_base);
#pragma unroll
      for (int s = 0; s < NA; ++s) {
        const int q = (tid + 256 * s) & 3;
        ra[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a_pix[s] >= 0) ra[s] = *reinterpret_cast<const f32x4*>(src + (long long)a_pix[s] * ld + q * 4);
      }
    }
#pragma unroll
    for (int s = 0; s < NWS; ++s) {
      const int idx = tid + 256 * s;          // 16-byte unit inside the [hi rows | lo rows] slab
      const int row = idx >> 6, n = idx & 63;  // row = (tap*2+h) (+18 for lo)
      if (idx < NWU) {
        const int rr = row >= 18 ? row - 18 : row;
        const unsigned short* g = wq + (row >= 18 ? lo_off : 0) + ((long long)(c * 18 + rr) * a.CoutP + n0 + n) * 8;
        rw[s] = *reinterpret_cast<const u32x4*>(g);
      }
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int s = 0; s < NA; ++s) {
      if (a_lds[s] < 0) continue;
      const f32x4 v = ra[s];
      if (SRC16) {     // already fp16: the 16 loaded bytes ARE the LDS image
        *reinterpret_cast<f32x4*>(sA + a_lds[s]) = v;
        continue;
      }
      u32x2 hi, lo;
      hi[0] = pack16<F16>(v[0], v[1]);
      hi[1] = pack16<F16>(v[2], v[3]);
      *reinterpret_cast<u32x2*>(sA + a_lds[s]) = hi;
      if (F::ALO) {
        lo[0] = pack16<F16>(v[0] - round16<F16>(v[0]), v[1] - round16<F16>(v[1]));
        lo[1] = pack16<F16>(v[2] - round16<F16>(v[2]), v[3] - round16<F16>(v[3]));
        *reinterpret_cast<u32x2*>(sA + a_lds[s] + 32) = lo;
      }
    }
#pragma unroll
    for (int s = 0; s < NWS; ++s) {
      const int idx = tid + 256 * s;
      if (idx < NWU) *reinterpret_cast<u32x4*>(sW + idx * 16) = rw[s];
    }
  };

  frag_t fah[2][2], fal[2][2], fbh[2][2], fbl[2][2];     // [parity][tile]: fragments are read one tap ahead
  auto load_frags = [&](int t, int par) {
    const int dy = t / 3, dx = t - dy * 3;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      fah[par][mi] = *reinterpret_cast<const frag_t*>(sA + a_off[mi] + (dy * IW + dx) * PIXB);
      if (F::ALO) fal[par][mi] = *reinterpret_cast<const frag_t*>(sA + a_off[mi] + (dy * IW + dx) * PIXB + 32);
    }
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      fbh[par][ni] = *reinterpret_cast<const frag_t*>(sW + (t * 2 * 64 + ni * 32) * 16 + b_off);
      if (F::WLO) fbl[par][ni] = *reinterpret_cast<const frag_t*>(sW + W_HALF + (t * 2 * 64 + ni * 32) * 16 + b_off);
    }
  };
  auto mma_tap = [&](int par) {
    if (DBG & 1) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni][0] += (float)fah[par][mi][0] * (float)fbh[par][ni][0];
      return;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        if (F::ALO) acc[mi][ni] = mma(fal[par][mi], fbh[par][ni], acc[mi][ni]);
        if (F::WLO) acc[mi][ni] = mma(fah[par][mi], fbl[par][ni], acc[mi][ni]);
        acc[mi][ni] = mma(fah[par][mi], fbh[par][ni], acc[mi][ni]);
      }
  };

  constexpr bool OFFMASK_OK = (MODE == M_BF16X3 || MODE == M_FP16X2 || MODE == M_FP16X1) && NT == 2 && DBG == 0;
  constexpr bool OFFMASK_EARLY = OFFMASK_OK && MODE == M_FP16X1;     // the mode with the registers for it
  const bool offmask = OFFMASK_OK && a.store_mode == CDFO_STORE_OFFMASK;
  OffmaskLane<NT> om;

  issue_loads(0);
  write_lds();
  __syncthreads();
  // the single-pass offset head runs its last chunk outside the loop (below): the staging registers are dead there, which is
  // what lets the epilogue's reads be requested ahead of that chunk's MFMAs without spilling
  const int nmain = (OFFMASK_EARLY && offmask) ? nchunks - 1 : nchunks;
  for (int c = 0; c < nmain; ++c) {
    // two waves per SIMD: next chunk's operands travel in registers while the MFMAs below run; with three waves per
    // SIMD the registers are not there (168 budget) and the other workgroups cover the latency instead
    if (WAVES_PER_SIMD <= 2 && c + 1 < nchunks && !(DBG & 2)) issue_loads(c + 1);
    if (a.tap_mask) {
      // sparse taps (stride-2-composed conv in space-to-depth form: 4 of the 9 taps carry weights per chunk)
      const unsigned tm = a.tap_mask[c];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (!((tm >> t) & 1u)) continue;
        load_frags(t, 0);
        mma_tap(0);
      }
    } else {
      if (WAVES_PER_SIMD <= 2) {   // two waves per SIMD: read the fragments one tap ahead
        load_frags(0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if (t < 8) load_frags(t + 1, (t & 1) ^ 1);
          mma_tap(t & 1);
        }
      } else {                     // three waves per SIMD: the other waves hide the LDS latency.  The tap loop is
#pragma unroll 1                   // kept rolled so that hipcc cannot hoist nine taps of fragments (216 VGPRs) at once
        for (int t = 0; t < 9; ++t) {
          load_frags(t, 0);
          mma_tap(0);
        }
      }
    }
    if (!(DBG & 16)) __syncthreads();  // everyone is done reading this chunk's LDS image
    if (c + 1 < nchunks) {
      if (WAVES_PER_SIMD > 2) issue_loads(c + 1);
      if (!(DBG & 4)) write_lds();
      if (!(DBG & 16)) __syncthreads();
    }
  }
  if (OFFMASK_EARLY && offmask) {
    offmask_prefetch<NT>(a, om, lane, b, oy0 + wave * 2, ox0, n0);
    load_frags(0, 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t < 8) load_frags(t + 1, (t & 1) ^ 1);
      mma_tap(t & 1);
    }
    __syncthreads();
  }

  // ---- epilogue (identical contract to conv_igemm_f32) through a wave-private LDS transpose; the loop above ends
  // with a __syncthreads(), so the staged operands are dead
  if (DBG & 8) {
    float t = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[mi][ni][e];
    if (t == 123.456f) a.out[0] = t;
    return;
  }
  if (offmask) {
    if (!OFFMASK_EARLY) offmask_prefetch<NT>(a, om, lane, b, oy0 + wave * 2, ox0, n0);
    offmask_finish<NT>(a, om, acc, lane, oy0 + wave * 2, ox0);
    return;
  }
  float* wl = reinterpret_cast<float*>(smem) + wave * ConvEpi<NT>::WAVE_FLOATS;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) conv_tile_epilogue_row32<NT>(a, wl, acc[mi], lane, b, oy0 + wave * 2 + mi, ox0, n0);
}

// OIHW fp32 -> 16-bit weights [Cin/16][9][2][CoutP][8]  (k = 16*chunk + 8*h + j):
//   bf16: [hi block | lo block];  fp16: one block (single rounding)
template <bool F16>
__global__ void pack16_kernel(const float* __restrict__ w, unsigned short* __restrict__ p, int Cout, int Cin, int CoutP) {
  const long long half = (long long)(Cin / 16) * 18 * CoutP * 8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < half;
       i += (long long)gridDim.x * blockDim.x) {
    const int j = i & 7;
    long long rest = i >> 3;
    const int n = rest % CoutP; rest /= CoutP;
    const int hh = rest & 1; rest >>= 1;
    const int t = rest % 9;
    const int c = rest / 9;
    const int cin = c * 16 + hh * 8 + j;
    float v = 0.f;
    if (n < Cout) v = w[((long long)n * Cin + cin) * 9 + t];
    if (F16) {
      p[i] = __builtin_bit_cast(unsigned short, (_Float16)v);
    } else {
      const __bf16 hi = (__bf16)v;
      p[i] = __builtin_bit_cast(unsigned short, hi);
      p[half + i] = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)hi));
    }
  }
}

}  // namespace

extern "C" int cdfo_conv3x3_bf16(const cdfo_conv_args* pa, void* stream) {
  const cdfo_conv_args& a = *pa;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a.nsrc < 1 || a.nsrc > CDFO_MAXSRC || a.B <= 0 || a.ks != 3 || a.stride != 1 || a.pad != 1) return CDFO_EINVAL;
  if (a.act == CDFO_ACT_SIGMOID) return CDFO_EINVAL;   // conv epilogues support none / LeakyReLU / ReLU
  if (a.out2_cp16 || a.res_up2 || a.src_plane_wrap || a.res2_pixscale) return CDFO_EINVAL;
  int csum = 0;
  for (int s = 0; s < a.nsrc; ++s) {
    if (a.cs[s] <= 0 || a.cs[s] % 16 || a.ld[s] % 4 || a.ld[s] < a.cs[s]) return CDFO_EINVAL;
    if (!aligned16(a.src[s])) return CDFO_EALIGN;
    csum += a.cs[s];
  }
  if (csum != a.Cin || a.CoutP % 64 || a.CoutP < a.Cout || a.Cout <= 0 || a.Ho != a.H || a.Wo != a.W) return CDFO_EINVAL;
  const bool offmask = a.store_mode == CDFO_STORE_OFFMASK;
  if (offmask) {
    const int pr = a.prec & 255;
    if ((pr != CDFO_PREC_BF16X3 && pr != CDFO_PREC_FP16X2 && pr != CDFO_PREC_FP16X1) || (a.prec >> 8) || a.Cout % 3 || a.Cout <= 32 || a.Wo % 4 || a.act != CDFO_ACT_NONE ||
        a.res1 || a.res2 || a.out_f16 || a.tap_mask || !a.mask_out || !a.flow || a.flow_bstride % 4 ||
        (long long)a.Ho * a.Wo >= (1ll << 31))
      return CDFO_EINVAL;
    if (!aligned16(a.mask_out) || !aligned16(a.flow)) return CDFO_EALIGN;
  }
  if ((a.store_mode != CDFO_STORE_PLAIN && a.store_mode != CDFO_STORE_S2D && !offmask) || a.w_bstride != 0) return CDFO_EINVAL;
  if (a.store_mode == CDFO_STORE_S2D && ((a.Ho | a.Wo) & 1 || a.res1 || a.res2)) return CDFO_EINVAL;
  if (!aligned16(a.w) || a.Cout % 4 || a.ldo % 4 || !aligned16(a.out) || (a.bias && !aligned16(a.bias))) return CDFO_EALIGN;
  if ((a.res1 && (a.ldr1 % 4 || !aligned16(a.res1))) || (a.res2 && (a.ldr2 % 4 || !aligned16(a.res2)))) return CDFO_EALIGN;
  if ((long long)a.B * a.H * a.W >= (1ll << 31)) return CDFO_EINVAL;
  if (a.src_f16 != ((a.prec & 255) == CDFO_PREC_FP16) || (a.src_f16 && (a.nsrc != 1 || a.ld[0] % 8))) return CDFO_EINVAL;
  if (a.out_f16 && (a.res1 || a.res2 || a.store_mode == CDFO_STORE_SHUFFLE2)) return CDFO_EINVAL;
  dim3 grid(cdiv(cdiv(a.Wo, TW) * cdiv(a.Ho, TH) * a.B, 8) * 8 * (a.CoutP / 64));
  const double px = (double)a.B * a.Ho * a.Wo;
  CdfoProfScope prof(st, KID_CONV3_WIDE, 2.0 * px * a.Cout * a.Cin * 9,
                     4.0 * (px * a.Cout + px * a.Cin + 9.0 * a.Cin * a.Cout));
  const int dbg = a.prec >> 8, prec = a.prec & 255;
  if (dbg) {   // developer ablations of the split-bf16 kernel (tools/bench_conv.py)
    if (prec != CDFO_PREC_BF16X3) return CDFO_EINVAL;
#define CDFO_DBG_CASE(D) case D: hipLaunchKernelGGL((conv3x3_mma16_kernel<M_BF16X3, 2, D>), grid, dim3(256), 0, st, a); break;
    switch (dbg) {
      CDFO_DBG_CASE(1) CDFO_DBG_CASE(2) CDFO_DBG_CASE(3) CDFO_DBG_CASE(4) CDFO_DBG_CASE(7) CDFO_DBG_CASE(8)
      CDFO_DBG_CASE(15) CDFO_DBG_CASE(16) CDFO_DBG_CASE(31)
      default: return CDFO_EINVAL;
    }
#undef CDFO_DBG_CASE
  } else if (prec == CDFO_PREC_BF16X3 && a.Cout <= 32 && a.CoutP == 64) {
    hipLaunchKernelGGL((conv3x3_mma16_kernel<M_BF16X3, 2, 0, 1>), grid, dim3(256), 0, st, a);
  } else if (prec == CDFO_PREC_FP16X2 && a.Cout <= 32 && a.CoutP == 64) {
    hipLaunchKernelGGL((conv3x3_mma16_kernel<M_FP16X2, 2, 0, 1>), grid, dim3(256), 0, st, a);
  } else if (prec == CDFO_PREC_BF16X3) {
    hipLaunchKernelGGL((conv3x3_mma16_kernel<M_BF16X3, 2, 0>), grid, dim3(256), 0, st, a);
  } else if (prec == CDFO_PREC_BF16) {
    hipLaunchKernelGGL((conv3x3_mma16_kernel<M_BF16, 2, 0>), grid, dim3(256), 0, st, a);
  } else if (prec == CDFO_PREC_FP16X2) {
    hipLaunchKernelGGL((conv3x3_mma16_kernel<M_FP16X2, 2, 0>), grid, dim3(256), 0, st, a);
  } else if (prec == CDFO_PREC_FP16) {
    hipLaunchKernelGGL((conv3x3_mma16_kernel<M_FP16IN, 2, 0>), grid, dim3(256), 0, st, a);
  } else if (prec == CDFO_PREC_FP16X1) {
    hipLaunchKernelGGL((conv3x3_mma16_kernel<M_FP16X1, 2, 0>), grid, dim3(256), 0, st, a);
  } else {
    return CDFO_EINVAL;
  }
  CDFO_LAUNCH_CHECK();
  return 0;
}

static int pack16_launch(const float* w, void* packed, int Cout, int Cin, bool f16, void* stream) {
  if (Cin % 16 || Cout <= 0) return CDFO_EINVAL;
  const int CoutP = (Cout + 63) / 64 * 64;
  const long long half = (long long)(Cin / 16) * 18 * CoutP * 8;
  const int blocks = (int)((half + 255) / 256 < 2048 ? (half + 255) / 256 : 2048);
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_PACK, 0, 8.0 * half);
  if (f16)
    hipLaunchKernelGGL(pack16_kernel<true>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                       static_cast<unsigned short*>(packed), Cout, Cin, CoutP);
  else
    hipLaunchKernelGGL(pack16_kernel<false>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                       static_cast<unsigned short*>(packed), Cout, Cin, CoutP);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_pack_conv3x3_bf16(const float* w_oihw, void* packed, int Cout, int Cin, void* stream) {
  return pack16_launch(w_oihw, packed, Cout, Cin, false, stream);
}
extern "C" int cdfo_pack_conv3x3_f16(const float* w_oihw, void* packed, int Cout, int Cin, void* stream) {
  return pack16_launch(w_oihw, packed, Cout, Cin, true, stream);
}
