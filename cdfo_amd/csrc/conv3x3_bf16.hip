// 3x3 stride-1 convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16) with fp32 accumulation.
//
//   PREC_BF16X3 : split-bf16 ("3-pass") arithmetic.  Every fp32 operand x is held as hi = bf16(x) and
//                 lo = bf16(x - hi); a product is a_hi*w_hi + a_lo*w_hi + a_hi*w_lo (the lo*lo term, 2^-16 relative,
//                 is dropped).  Max-abs error of the whole CVSR_V8 forward vs the fp32 reference stays ~1e-5, i.e.
//                 inside the 1e-3 parity bound, at 16/3 of the exact-fp32 MFMA rate.
//   PREC_BF16   : plain bf16 operands (one pass) -- the BASELINE "bf16" configuration; ~6e-3 max-abs on the forward.
//
// Structure (one 256-thread workgroup = 8 rows x 32 columns of output pixels x 64 output channels, 2 workgroups/CU;
// a 32-pixel MFMA M tile is ONE image row, which with the 80-byte pixel record makes every ds_read_b128 lane group hit
// 16 distinct 16-byte slots):
//   per 16-input-channel chunk: the 10x34x16 input halo tile is fetched as fp32 float4s into REGISTERS while the
//   previous chunk's MFMAs run (register-staged double buffering: global latency hides under the matrix pipe), then
//   converted to bf16 hi/lo and written to LDS (80-byte pixel records: 32 B hi | 32 B lo | 16 B pad, conflict-free
//   ds_read_b128); the pre-split bf16 weight slab [tap][k-half][cout][8] is copied the same way.  Each lane feeds
//   12 MFMAs (2x2 register tile x 3 passes) from 8 ds_read_b128 per tap.  No im2col buffer, no HBM intermediates.
//
// Replaces F.conv2d for the 3x3 convolutions with >= 64 output channels on the CVSR_V8 path
// (arch/SIDECVSR_our.py:383-387 Block_.body -- 89 % of the forward's FLOPs --, :435, :1447, :261-262, :4382).
#include "common.h"
#include "conv_epilogue.h"

namespace {

constexpr int TH = 8, TW = 32, IW = 34, NPIX = 34 * 10;   // a 32-pixel M tile = one image row: conflict-free b128 reads
constexpr int PIXB = 80;                         // LDS bytes per staged pixel
constexpr int A_BYTES = NPIX * PIXB;             // 27,200
constexpr int W_HALF = 9 * 2 * 64 * 16;          // 18,432 B: [tap][h][64 cout][8 bf16]
constexpr int NA = (NPIX * 4 + 255) / 256;       // 6 float4 per thread per chunk
constexpr int NW_X3 = 2 * W_HALF / 16 / 256;     // 9 x 16 B per thread (hi + lo)
constexpr int NW_X1 = W_HALF / 16 / 256;         // 4.5 -> handled as 5 with a guard

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const __bf16 ha = (__bf16)a, hb = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
}
__device__ __forceinline__ float bf16_round(float a) { return (float)(__bf16)a; }

// DBG (developer ablations, never used by the product path): 1 = skip the MFMAs, 2 = skip the per-chunk global loads,
// 4 = skip the per-chunk convert + LDS writes, 8 = skip the epilogue (store one value), 16 = skip the barriers
template <bool X3, int DBG = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_kernel(cdfo_conv_args a) {
  constexpr int LDS_W = X3 ? 2 * W_HALF : W_HALF;
  constexpr int NWS = X3 ? NW_X3 : NW_X1 + 1;
  __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES + LDS_W];
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
    const int p = idx >> 2, q = idx & 3;
    const int iy = p / IW, ix = p - iy * IW;
    const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
    const bool ok = idx < NPIX * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    a_pix[s] = ok ? (b * a.H + gy) * a.W + gx : -1;
    a_lds[s] = idx < NPIX * 4 ? p * PIXB + q * 8 : -1;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
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
    const float* src = a.src[s_idx] + (ch0 - s_base);
    const int ld = a.ld[s_idx];
#pragma unroll
    for (int s = 0; s < NA; ++s) {
      const int q = (tid + 256 * s) & 3;
      ra[s] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a_pix[s] >= 0) ra[s] = *reinterpret_cast<const f32x4*>(src + (long long)a_pix[s] * ld + q * 4);
    }
#pragma unroll
    for (int s = 0; s < NWS; ++s) {
      const int idx = tid + 256 * s;          // 16-byte unit inside the [hi rows | lo rows] slab
      const int row = idx >> 6, n = idx & 63;  // row = (tap*2+h) (+18 for lo)
      if (X3 || idx < W_HALF / 16) {
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
      u32x2 hi, lo;
      hi[0] = pack_bf16(v[0], v[1]);
      hi[1] = pack_bf16(v[2], v[3]);
      *reinterpret_cast<u32x2*>(sA + a_lds[s]) = hi;
      if (X3) {
        lo[0] = pack_bf16(v[0] - bf16_round(v[0]), v[1] - bf16_round(v[1]));
        lo[1] = pack_bf16(v[2] - bf16_round(v[2]), v[3] - bf16_round(v[3]));
        *reinterpret_cast<u32x2*>(sA + a_lds[s] + 32) = lo;
      }
    }
#pragma unroll
    for (int s = 0; s < NWS; ++s) {
      const int idx = tid + 256 * s;
      if (X3 || idx < W_HALF / 16) *reinterpret_cast<u32x4*>(sW + idx * 16) = rw[s];
    }
  };

  issue_loads(0);
  write_lds();
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks && !(DBG & 2)) issue_loads(c + 1);  // in flight while the MFMAs below run
    bf16x8_t fah[2][2], fal[2][2], fbh[2][2], fbl[2][2];     // [parity][tile]: fragments are read one tap ahead
    auto load_frags = [&](int t, int par) {
      const int dy = t / 3, dx = t - dy * 3;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        fah[par][mi] = *reinterpret_cast<const bf16x8_t*>(sA + a_off[mi] + (dy * IW + dx) * PIXB);
        if (X3) fal[par][mi] = *reinterpret_cast<const bf16x8_t*>(sA + a_off[mi] + (dy * IW + dx) * PIXB + 32);
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        fbh[par][ni] = *reinterpret_cast<const bf16x8_t*>(sW + (t * 2 * 64 + ni * 32) * 16 + b_off);
        if (X3) fbl[par][ni] = *reinterpret_cast<const bf16x8_t*>(sW + W_HALF + (t * 2 * 64 + ni * 32) * 16 + b_off);
      }
    };
    if (a.tap_mask) {
      // sparse taps (stride-2-composed conv in space-to-depth form: 4 of the 9 taps carry weights per chunk)
      const unsigned tm = a.tap_mask[c];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (!((tm >> t) & 1u)) continue;
        load_frags(t, 0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            if (X3) {
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[0][mi], fbh[0][ni], acc[mi][ni], 0, 0, 0);
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[0][mi], fbl[0][ni], acc[mi][ni], 0, 0, 0);
            }
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[0][mi], fbh[0][ni], acc[mi][ni], 0, 0, 0);
          }
      }
    } else {
    load_frags(0, 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int par = t & 1;
      if (t < 8) load_frags(t + 1, par ^ 1);
      if (DBG & 1) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            acc[mi][ni][0] += (float)fah[par][mi][0] * (float)fbh[par][ni][0];
            if (X3) acc[mi][ni][1] += (float)fal[par][mi][0] * (float)fbl[par][ni][0];
          }
        continue;
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          if (X3) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[par][mi], fbh[par][ni], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[par][mi], fbl[par][ni], acc[mi][ni], 0, 0, 0);
          }
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[par][mi], fbh[par][ni], acc[mi][ni], 0, 0, 0);
        }
    }
    }
    if (!(DBG & 16)) __syncthreads();  // everyone is done reading this chunk's LDS image
    if (c + 1 < nchunks) {
      if (!(DBG & 4)) write_lds();
      if (!(DBG & 16)) __syncthreads();
    }
  }

  // ---- epilogue (identical contract to conv_igemm_f32) through a wave-private LDS transpose; the loop above ends
  // with a __syncthreads(), so the staged operands are dead
  if (DBG & 8) {
    float t = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[mi][ni][e];
    if (t == 123.456f) a.out[0] = t;
    return;
  }
  float* wl = reinterpret_cast<float*>(smem) + wave * ConvEpi<2>::WAVE_FLOATS;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) conv_tile_epilogue_row32<2>(a, wl, acc[mi], lane, b, oy0 + wave * 2 + mi, ox0, n0);
}



// OIHW fp32 -> [hi | lo] bf16, each [Cin/16][9][2][CoutP][8]  (k = 16*chunk + 8*h + j)
__global__ void pack_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ p, int Cout, int Cin,
                                 int CoutP) {
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
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    p[i] = __builtin_bit_cast(unsigned short, hi);
    p[half + i] = __builtin_bit_cast(unsigned short, lo);
  }
}

}  // namespace

extern "C" int cdfo_conv3x3_bf16(const cdfo_conv_args* pa, void* stream) {
  const cdfo_conv_args& a = *pa;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a.nsrc < 1 || a.nsrc > CDFO_MAXSRC || a.B <= 0 || a.ks != 3 || a.stride != 1 || a.pad != 1) return CDFO_EINVAL;
  if (a.act == CDFO_ACT_SIGMOID) return CDFO_EINVAL;   // conv epilogues support none / LeakyReLU / ReLU
  int csum = 0;
  for (int s = 0; s < a.nsrc; ++s) {
    if (a.cs[s] <= 0 || a.cs[s] % 16 || a.ld[s] % 4 || a.ld[s] < a.cs[s]) return CDFO_EINVAL;
    if (!aligned16(a.src[s])) return CDFO_EALIGN;
    csum += a.cs[s];
  }
  if (csum != a.Cin || a.CoutP % 64 || a.CoutP < a.Cout || a.Cout <= 0 || a.Ho != a.H || a.Wo != a.W) return CDFO_EINVAL;
  if ((a.store_mode != CDFO_STORE_PLAIN && a.store_mode != CDFO_STORE_S2D) || a.w_bstride != 0) return CDFO_EINVAL;
  if (a.store_mode == CDFO_STORE_S2D && ((a.Ho | a.Wo) & 1 || a.res1 || a.res2)) return CDFO_EINVAL;
  if (!aligned16(a.w) || a.Cout % 4 || a.ldo % 4 || !aligned16(a.out) || (a.bias && !aligned16(a.bias))) return CDFO_EALIGN;
  if ((a.res1 && (a.ldr1 % 4 || !aligned16(a.res1))) || (a.res2 && (a.ldr2 % 4 || !aligned16(a.res2)))) return CDFO_EALIGN;
  if ((long long)a.B * a.H * a.W >= (1ll << 31)) return CDFO_EINVAL;
  dim3 grid(cdiv(cdiv(a.Wo, TW) * cdiv(a.Ho, TH) * a.B, 8) * 8 * (a.CoutP / 64));
  const double px = (double)a.B * a.Ho * a.Wo;
  CdfoProfScope prof(st, KID_CONV3_WIDE, 2.0 * px * a.Cout * a.Cin * 9,
                     4.0 * (px * a.Cout + px * a.Cin + 9.0 * a.Cin * a.Cout));
  const int dbg = a.prec >> 8;
  if ((a.prec & 255) == CDFO_PREC_BF16X3 && dbg) {
#define CDFO_DBG_CASE(D) case D: hipLaunchKernelGGL((conv3x3_bf16_kernel<true, D>), grid, dim3(256), 0, st, a); break;
    switch (dbg) {
      CDFO_DBG_CASE(1) CDFO_DBG_CASE(2) CDFO_DBG_CASE(3) CDFO_DBG_CASE(4) CDFO_DBG_CASE(7) CDFO_DBG_CASE(8)
      CDFO_DBG_CASE(9) CDFO_DBG_CASE(15) CDFO_DBG_CASE(16) CDFO_DBG_CASE(31) CDFO_DBG_CASE(23) CDFO_DBG_CASE(6)
      default: return CDFO_EINVAL;
    }
#undef CDFO_DBG_CASE
  }
  else if (a.prec == CDFO_PREC_BF16X3)
    hipLaunchKernelGGL(conv3x3_bf16_kernel<true>, grid, dim3(256), 0, st, a);
  else if (a.prec == CDFO_PREC_BF16)
    hipLaunchKernelGGL(conv3x3_bf16_kernel<false>, grid, dim3(256), 0, st, a);
  else
    return CDFO_EINVAL;
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_pack_conv3x3_bf16(const float* w_oihw, void* packed, int Cout, int Cin, void* stream) {
  if (Cin % 16 || Cout <= 0) return CDFO_EINVAL;
  const int CoutP = (Cout + 63) / 64 * 64;
  const long long half = (long long)(Cin / 16) * 18 * CoutP * 8;
  const int blocks = (int)((half + 255) / 256 < 2048 ? (half + 255) / 256 : 2048);
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_PACK, 0, 8.0 * half);
  hipLaunchKernelGGL(pack_bf16_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w_oihw,
                     static_cast<unsigned short*>(packed), Cout, Cin, CoutP);
  CDFO_LAUNCH_CHECK();
  return 0;
}
