// Implicit-GEMM convolution on the gfx950 matrix cores (fp32 path: v_mfma_f32_32x32x2_f32, exact fp32).
//
// GEMM view: M = output pixels (a 2x16 patch per 32-row MFMA tile), N = output channels, K = taps x Cin.
// One 256-thread workgroup owns a TH x 16 pixel tile and BN = 32*NT output channels.  Per 16-channel chunk
// of K it stages the input halo tile (pixel-major, 80-byte pixel pitch => conflict-free ds_read_b128) and
// the matching weight slab ([tap][cin/4][cout][4]) in LDS; every lane then feeds four consecutive MFMA
// k-steps from one 16-byte LDS read of A and one of B.  No im2col buffer ever exists in HBM.
//
// Replaces F.conv2d for every 1x1 / 3x3 convolution with >= 16 input channels on the CVSR_V8 path
// (arch/SIDECVSR_our.py:383-387, 4382, 4386, 4390-4391, ...), torch.cat in front of it (multi-source K loop),
// the bias / LeakyReLU / ReLU / residual adds behind it, and PixelShuffle(2) (arch.py:4473-4474).
#include "common.h"
#include "conv_epilogue.h"

namespace {

constexpr int TW = 16;       // tile width in output pixels

template <int KS, int S, int TH, int NT, int KC>   // KC = input channels per K chunk (16, or 64 for 1x1)
struct Geo {
  static constexpr int AST = KC + 4;  // LDS floats per staged pixel (+16 B: conflict-free ds_read_b128)
  static constexpr int MT = TH / 8;  // 32-pixel M tiles per wave (4 waves)
  static constexpr int BN = 32 * NT;
  static constexpr int IH = (TH - 1) * S + KS;
  static constexpr int IW = (TW - 1) * S + KS;
  static constexpr int NPIX = IH * IW;
  static constexpr int A_FLOATS = ((NPIX * AST + 3) / 4) * 4;
  static constexpr int W_FLOATS = KS * KS * (KC / 4) * BN * 4;
  static constexpr int MAIN_BYTES = (A_FLOATS + W_FLOATS) * 4;
  static constexpr int LDS_BYTES = MAIN_BYTES > ConvEpi<NT>::BLOCK_BYTES ? MAIN_BYTES : ConvEpi<NT>::BLOCK_BYTES;
};

template <int KS, int S, int TH, int NT, int KC>
__global__ __launch_bounds__(256) void conv_igemm_f32(cdfo_conv_args a) {
  using G = Geo<KS, S, TH, NT, KC>;
  constexpr int MT = G::MT, BN = G::BN, IW = G::IW, NPIX = G::NPIX, T = KS * KS, KG = KC / 4, AST = G::AST;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;
  float* sW = smem + G::A_FLOATS;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, r = lane & 31;
  // XCD-aware block order (see conv3x3_bf16.hip): the output-channel blocks of one input tile share an XCD's L2
  const int tiles_x = (a.Wo + TW - 1) / TW, tiles = tiles_x * ((a.Ho + TH - 1) / TH);
  const int nco = a.CoutP / BN;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pt = (slot / nco) * 8 + xcd;
  if (pt >= tiles * a.B) return;
  const int b = pt / tiles, tile = pt - b * tiles;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int oy0 = ty * TH, ox0 = tx * TW, n0 = (slot % nco) * BN;
  const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
  const float* wbase = a.w + (long long)b * a.w_bstride;
  const int cin4 = a.Cin >> 2;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  // LDS read offsets (floats) that do not depend on the chunk
  int a_off[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int m = wave * MT + mi;
    a_off[mi] = (((2 * m + (r >> 4)) * S) * IW + (r & 15) * S) * AST + 4 * h;
  }
  const int b_off = (h * BN + r) * 4;

  const int nchunks = a.Cin / KC;
  int s_idx = 0, s_base = 0;  // current source and its first concatenated channel
  for (int c = 0; c < nchunks; ++c) {
    const int ch0 = c * KC;
    while (ch0 >= s_base + a.cs[s_idx]) { s_base += a.cs[s_idx]; ++s_idx; }
    const float* src = a.src[s_idx];
    const int ld = a.ld[s_idx];
    const int coff = ch0 - s_base;
    __syncthreads();
    // ---- stage the input tile, zero outside the image (= conv padding)
    if constexpr (KS == 1 && KC == 64) {
      // 1x1 streaming form: all of this thread's 16-byte loads are issued back to back (8 in flight), then the
      // optional per-pixel LayerNorm (16 lanes x float4 = one pixel's 64 channels; arch.py:1169-1185), then LDS
      constexpr int NL = NPIX * KG / 256;
      static_assert(NPIX * KG % 256 == 0, "tile must divide evenly");
      f32x4 v[NL];
      bool inside[NL];
#pragma unroll
      for (int s = 0; s < NL; ++s) {
        const int idx = tid + 256 * s;
        const int p = idx / KG, q = idx - p * KG;
        const int iy = p / IW, ix = p - iy * IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        inside[s] = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const long long pix = inside[s] ? ((long long)(b * a.H + gy) * a.W + gx) : 0ll;   // clamped, always loaded
        v[s] = *reinterpret_cast<const f32x4*>(src + pix * ld + coff + q * 4);
      }
      const bool do_ln = a.ln_gamma != nullptr;
#pragma unroll
      for (int s = 0; s < NL; ++s) {
        const int idx = tid + 256 * s;
        const int p = idx / KG, q = idx - p * KG;
        f32x4 t = v[s];
        if (do_ln) {
          float sm = (t[0] + t[1]) + (t[2] + t[3]);
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
          const f32x4 d = t - sm * (1.f / 64.f);
          float sq = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
          const float rstd = 1.f / sqrtf(sq * (1.f / 64.f) + 1e-5f);
          t = d * rstd * *reinterpret_cast<const f32x4*>(a.ln_gamma + q * 4) +
              *reinterpret_cast<const f32x4*>(a.ln_beta + q * 4);
        }
        if (!inside[s]) t = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(sA + p * AST + q * 4) = t;
      }
    } else {
      for (int idx = tid; idx < NPIX * KG; idx += 256) {
        const int p = idx / KG, q = idx - p * KG;
        const int iy = p / IW, ix = p - iy * IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
          v = *reinterpret_cast<const f32x4*>(src + ((long long)(b * a.H + gy) * a.W + gx) * ld + coff + q * 4);
        *reinterpret_cast<f32x4*>(sA + p * AST + q * 4) = v;
      }
    }
    // ---- stage the weight slab [tap][kg][BN][4]
    for (int idx = tid; idx < T * KG * BN; idx += 256) {
      const int n = idx % BN, tk = idx / BN;
      const int kg = tk % KG, t = tk / KG;
      const f32x4 v = *reinterpret_cast<const f32x4*>(
          wbase + ((long long)(t * cin4 + (ch0 >> 2) + kg) * a.CoutP + n0 + n) * 4);
      *reinterpret_cast<f32x4*>(sW + idx * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int dy = t / KS, dx = t - dy * KS;
#pragma unroll
      for (int j = 0; j < KC / 8; ++j) {
        f32x4 av[MT], bv[NT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
          av[mi] = *reinterpret_cast<const f32x4*>(sA + a_off[mi] + (dy * IW + dx) * AST + 8 * j);
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          bv[ni] = *reinterpret_cast<const f32x4*>(sW + ((t * KG + 2 * j) * BN + ni * 32) * 4 + b_off);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][e], bv[ni][e], acc[mi][ni], 0, 0, 0);
      }
    }
  }

  // ---- epilogue through a wave-private LDS transpose (conv_epilogue.h)
  __syncthreads();  // all waves are done with the staged operands
  float* wl = smem + wave * ConvEpi<NT>::WAVE_FLOATS;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
    conv_tile_epilogue<NT>(a, wl, acc[mi], lane, b, oy0 + 2 * (wave * MT + mi), ox0, n0);
}

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int ks,
                                   int CoutP, int shuffle2, int transposed) {
  const int T = ks * ks;
  const long long total = (long long)T * Cin * CoutP;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int e = i & 3;
    long long rest = i >> 2;
    const int n = rest % CoutP; rest /= CoutP;
    const int c4 = rest % (Cin / 4);
    const int t = rest / (Cin / 4);
    const int cin = c4 * 4 + e;
    float v = 0.f;
    if (n < Cout) {
      int o = n;
      if (shuffle2) { const int cq = Cout / 4; o = (n % cq) * 4 + n / cq; }
      if (!transposed) v = w[((long long)o * Cin + cin) * T + t];
      else v = w[((long long)cin * Cout + o) * T + (T - 1 - t)];  // IOHW, taps flipped
    }
    p[i] = v;
  }
}

template <int KS, int S, int TH, int NT, int KC>
int launch(const cdfo_conv_args& a, hipStream_t st) {
  using G = Geo<KS, S, TH, NT, KC>;
  static CdfoAttrOnce once;
  const hipError_t ea = cdfo_set_max_lds(once, reinterpret_cast<const void*>(&conv_igemm_f32<KS, S, TH, NT, KC>), G::LDS_BYTES);
  if (ea != hipSuccess) return (int)ea;
  dim3 grid(cdiv(cdiv(a.Wo, TW) * cdiv(a.Ho, TH) * a.B, 8) * 8 * (a.CoutP / G::BN));
  const int kid = KS == 1 ? KID_CONV1 : (S == 2 ? KID_CONV3_S2 : (NT == 2 ? KID_CONV3_WIDE : KID_CONV3_NARROW));
  const double px = (double)a.B * a.Ho * a.Wo;
  CdfoProfScope prof(st, kid, 2.0 * px * a.Cout * a.Cin * KS * KS,
                     4.0 * (px * a.Cout + (double)a.B * a.H * a.W * a.Cin + (double)KS * KS * a.Cin * a.Cout));
  hipLaunchKernelGGL((conv_igemm_f32<KS, S, TH, NT, KC>), grid, dim3(256), G::LDS_BYTES, st, a);
  CDFO_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int cdfo_conv_igemm(const cdfo_conv_args* pa, void* stream) {
  const cdfo_conv_args& a = *pa;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a.nsrc < 1 || a.nsrc > CDFO_MAXSRC || a.B <= 0 || a.H <= 0 || a.W <= 0) return CDFO_EINVAL;
  if (a.act == CDFO_ACT_SIGMOID || a.out2_cp16 || a.res_up2 || a.src_plane_wrap || a.res2_pixscale) return CDFO_EINVAL;   // conv epilogues support none / LeakyReLU / ReLU
  int csum = 0;
  for (int s = 0; s < a.nsrc; ++s) {
    if (a.cs[s] <= 0 || a.cs[s] % 16 || a.ld[s] % 4 || a.ld[s] < a.cs[s]) return CDFO_EINVAL;
    if (!aligned16(a.src[s])) return CDFO_EALIGN;
    csum += a.cs[s];
  }
  if (csum != a.Cin || a.CoutP % 32 || a.CoutP < a.Cout || a.Cout <= 0) return CDFO_EINVAL;
  if (!aligned16(a.w) || a.w_bstride % 4) return CDFO_EALIGN;
  if (a.Ho != (a.H + 2 * a.pad - a.ks) / a.stride + 1 || a.Wo != (a.W + 2 * a.pad - a.ks) / a.stride + 1)
    return CDFO_EINVAL;
  if (a.store_mode == CDFO_STORE_SHUFFLE2 && (a.Cout % 16 || a.res1 || a.res2)) return CDFO_EINVAL;
  if (a.Cout % 4 || a.ldo % 4 || !aligned16(a.out) || (a.bias && !aligned16(a.bias))) return CDFO_EALIGN;
  if ((a.res1 && (a.ldr1 % 4 || !aligned16(a.res1))) || (a.res2 && (a.ldr2 % 4 || !aligned16(a.res2)))) return CDFO_EALIGN;
  if (a.prec != CDFO_PREC_F32 || a.tap_mask || a.src_f16 || a.out_f16) return CDFO_EINVAL;
  if (a.store_mode == CDFO_STORE_S2D && ((a.Ho | a.Wo) & 1 || a.res1 || a.res2)) return CDFO_EINVAL;
  const bool wide = (a.CoutP % 64) == 0;
  bool k64 = a.ks == 1 && a.stride == 1 && wide;       // 1x1: stage whole 64-channel pixels (HBM-bound streaming)
  for (int s = 0; s < a.nsrc; ++s) k64 = k64 && (a.cs[s] % 64 == 0);
  if (a.ln_gamma && !(k64 && a.nsrc == 1 && a.cs[0] == 64 && a.ln_beta)) return CDFO_EINVAL;
  if (a.ln_gamma && (!aligned16(a.ln_gamma) || !aligned16(a.ln_beta))) return CDFO_EALIGN;
  if (k64) return launch<1, 1, 8, 2, 64>(a, st);
  if (a.ks == 3 && a.stride == 1) return wide ? launch<3, 1, 16, 2, 16>(a, st) : launch<3, 1, 16, 1, 16>(a, st);
  if (a.ks == 1 && a.stride == 1) return wide ? launch<1, 1, 16, 2, 16>(a, st) : launch<1, 1, 16, 1, 16>(a, st);
  if (a.ks == 3 && a.stride == 2) return wide ? launch<3, 2, 8, 2, 16>(a, st) : launch<3, 2, 8, 1, 16>(a, st);
  return CDFO_EINVAL;
}

extern "C" int cdfo_pack_conv_weight(const float* w, float* packed, int Cout, int Cin, int ks, int shuffle2,
                                     int transposed, void* stream) {
  if (Cin % 4 || Cout <= 0 || ks <= 0) return CDFO_EINVAL;
  const int CoutP = (Cout + 31) / 32 * 32;
  const long long total = (long long)ks * ks * Cin * CoutP;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_PACK, 0, 8.0 * total);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w, packed, Cout,
                     Cin, ks, CoutP, shuffle2, transposed);
  CDFO_LAUNCH_CHECK();
  return 0;
}
