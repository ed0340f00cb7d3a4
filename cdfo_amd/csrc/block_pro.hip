// Block_ prologue (arch/SIDECVSR_our.py:378-406): from ONE read of the block input x produce the sources of the two
// resampled branches, as the fp16 chunk-planar tensors cdfo_conv3x3_c64_ws reads:
//     u16 = up2( up.0(x) )      [B][4][2H][2W][16]   (bilinear x2, align_corners=False, of the 1x1 conv `up.0`)
//     d16 = down.0( down2(x) )  [B][4][H/2][W/2][16] (1x1 conv `down.0` of the 2x2 mean)
// The 1x1 convs commute with the (linear) resampling, so both run at the block's own resolution on the matrix cores
// (split-bf16, 3 passes, fp32-grade) and the resampling happens on the LDS tile.  Replaces five launches (2x2 mean,
// down.0, layout change, up.0, bilinear x2) that read x or a 64-channel fp32 intermediate four times.
//
// Structure = qkv_dw.hip's: persistent 512-thread workgroup per CU, weights resident in LDS, output tile 6 x 30 pixels
// whose halo region is 8 rows x 32 pixels = one MFMA M tile per wave, A fragments loaded straight from global memory.
// Halo pixels outside the image are loaded from the CLAMPED coordinate: bilinear interpolation clamps its taps, so the
// halo of y = up.0(x) must be the edge replica.
#include "common.h"

namespace {

constexpr int BP_THREADS = 512;
constexpr int BP_TR = 6, BP_TC = 30;
constexpr int BP_W_BYTES = 2 * 4 * 2 * 128 * 16;          // [hi|lo][k-step][k-half][128 cout: up.0 | down.0][8 bf16] = 32,768
// the resampling tile holds 32 channels at a time (four passes: up.0 | down.0 x two halves): 32 KB, so that TWO workgroups
// fit a CU (65.5 KB each) and one's resampling / store phase overlaps the other's MFMA phase -- with one 103 KB workgroup per
// CU the phases ran back to back at half the HBM rate.  Pitch 32 floats: a pixel's 8 channel quads are 128 contiguous bytes,
// consecutive pixels alternate between the two halves of the 64 banks = conflict-free ds_read_b128
constexpr int BP_YP = 32;
constexpr int BP_Y_BYTES = 8 * 32 * BP_YP * 4;            // 32,768
constexpr int BP_BIAS_OFF = BP_W_BYTES + BP_Y_BYTES;
constexpr int BP_LDS = BP_BIAS_OFF + 128 * 4;             // 66,048 bytes

typedef __bf16 bp_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bp_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 bp_f16x8 __attribute__((ext_vector_type(8)));

struct bp_args {
  const float* x; int ldx;
  int B, H, W;
  const unsigned short* w;      // bf16 [hi|lo][4][2][128][8]: rows 0-63 = up.0, 64-127 = down.0; k = 16 s + 8 h + j
  const float* bias;            // [128]
  _Float16* u16;                // [B][4][2H][2W][16]   (NULL when t16 is given)
  _Float16* t16;                // optional [B][4][H][W][16]: up.0(x) itself at the block's resolution -- the Winograd kernel's
                                // on-the-fly x2 form interpolates it while it builds its transformed inputs (conv3x3_wino.hip, UP)
  _Float16* d16;                // [B][4][H/2][W/2][16]
  _Float16* x16;                // optional [B][4][H][W][16]: fp16 chunk-planar copy of x itself (the 1x branch's source)
};

__global__ __launch_bounds__(BP_THREADS, 4) void block_pro_kernel(bp_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sY = reinterpret_cast<float*>(smem + BP_W_BYTES);
  float* sBias = reinterpret_cast<float*>(smem + BP_BIAS_OFF);
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, Ho = 2 * H, Wo = 2 * W, Hd = H >> 1, Wd = W >> 1;

  for (int i = tid; i < BP_W_BYTES / 16; i += BP_THREADS)
    reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(a.w)[i];
  for (int i = tid; i < 128; i += BP_THREADS) sBias[i] = a.bias[i];
  __syncthreads();

  const int tiles_x = (W + BP_TC - 1) / BP_TC, tiles_y = (H + BP_TR - 1) / BP_TR;
  const int ntiles = a.B * tiles_y * tiles_x;

  f32x4 xr[8];          // halo row `wave`, this lane's pixel r (clamped into the image), channels 16 s + 8 h .. + 7
  auto load_x = [&](int t) {
    const int tx = t % tiles_x, t2 = t / tiles_x;
    const int ty = t2 % tiles_y, b = t2 / tiles_y;
    int gy = ty * BP_TR - 1 + wave, gx = tx * BP_TC - 1 + r;
    gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
    gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
    const float* px = a.x + ((long long)(b * H + gy) * W + gx) * a.ldx + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      xr[2 * s] = *reinterpret_cast<const f32x4*>(px + 16 * s);
      xr[2 * s + 1] = *reinterpret_cast<const f32x4*>(px + 16 * s + 4);
    }
  };

  int t = blockIdx.x;
  if (t < ntiles) load_x(t);
  for (; t < ntiles; t += gridDim.x) {
    const int tx = t % tiles_x, t2 = t / tiles_x;
    const int ty = t2 % tiles_y, b = t2 / tiles_y;
    const int oy0 = ty * BP_TR, ox0 = tx * BP_TC;
    bp_bf16x8 ah[4], al[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = xr[2 * s + (j >> 2)][j & 3];
        ah[s][j] = (__bf16)v;
        al[s][j] = (__bf16)(v - (float)ah[s][j]);
      }
    if (a.x16) {
      // the tile's own pixels (halo rows 1-6, columns 1-30) also leave as the fp16 chunk-planar copy of x: this lane holds
      // channels 8h .. 8h+7 of each 16-channel chunk s = one 16-byte store per chunk
      const int gy = oy0 - 1 + wave, gx = ox0 - 1 + r;
      if (wave >= 1 && wave <= BP_TR && r >= 1 && r <= BP_TC && gy < H && gx < W) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          bp_f16x8 hv;
#pragma unroll
          for (int j = 0; j < 8; ++j) hv[j] = (_Float16)xr[2 * s + (j >> 2)][j & 3];
          *reinterpret_cast<bp_f16x8*>(a.x16 + ((((long long)b * 4 + s) * H + gy) * W + gx) * 16 + 8 * h) = hv;
        }
      }
    }
    const int tn = t + gridDim.x;
    if (tn < ntiles) load_x(tn);

#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {  // (nb, nt): nb 0: y = up.0(x) -> bilinear x2;  1: z = down.0(x) -> 2x2 mean; nt: 32-channel half
      const int nb = pass >> 1, nt = pass & 1;
      {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const int n = nb * 64 + nt * 32 + r;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bp_bf16x8 wh = *reinterpret_cast<const bp_bf16x8*>(smem + ((s * 2 + h) * 128 + n) * 16);
          const bp_bf16x8 wl = *reinterpret_cast<const bp_bf16x8*>(smem + BP_W_BYTES / 2 + ((s * 2 + h) * 128 + n) * 16);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], wh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], wl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], wh, acc, 0, 0, 0);
        }
        const float bn = sBias[n];
#pragma unroll
        for (int e = 0; e < 16; ++e)
          sY[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * BP_YP + r] = acc[e] + bn;
      }
      __syncthreads();
      if (nb == 0 && a.t16) {
        // y = up.0(x) leaves as it is (fp16 chunk-planar, the tile's own 6 x 30 pixels): the consumer interpolates
        for (int i = tid; i < BP_TR * BP_TC * 8; i += BP_THREADS) {
          const int q = i & 7, px = (i >> 3) % BP_TC, py = (i >> 3) / BP_TC;
          const int cq = nt * 8 + q;
          const int gy = oy0 + py, gx = ox0 + px;
          if (gy >= H || gx >= W) continue;
          const f32x4 v = *reinterpret_cast<const f32x4*>(sY + ((1 + py) * 32 + 1 + px) * BP_YP + q * 4);
          bp_f16x4 hv;
#pragma unroll
          for (int k = 0; k < 4; ++k) hv[k] = (_Float16)v[k];
          *reinterpret_cast<bp_f16x4*>(a.t16 + ((((long long)b * 4 + (cq >> 2)) * H + gy) * W + gx) * 16 + (cq & 3) * 4) = hv;
        }
      } else if (nb == 0) {
        // bilinear x2: thread = 4 channels (quad q of this half) of the 2x2 output block (2Q-1..2Q, 2P-1..2P) fed by source
        // rows Q-1..Q, columns P-1..P (halo-local rows qq..qq+1, columns pp..pp+1; the clamped replicas are already in the
        // tile).  7 x 31 blocks cover the tile's 12 x 60 outputs (+ the shared edge with the neighbouring tiles, written by
        // whichever tile owns the output pixel).
        for (int i = tid; i < 7 * 31 * 8; i += BP_THREADS) {
          const int q = i & 7, pp = (i >> 3) % 31, qq = (i >> 3) / 31;
          const int cq = nt * 8 + q;                      // channel quad of the 64
          const f32x4 vaa = *reinterpret_cast<const f32x4*>(sY + (qq * 32 + pp) * BP_YP + q * 4);
          const f32x4 vab = *reinterpret_cast<const f32x4*>(sY + (qq * 32 + pp + 1) * BP_YP + q * 4);
          const f32x4 vba = *reinterpret_cast<const f32x4*>(sY + ((qq + 1) * 32 + pp) * BP_YP + q * 4);
          const f32x4 vbb = *reinterpret_cast<const f32x4*>(sY + ((qq + 1) * 32 + pp + 1) * BP_YP + q * 4);
          const int Q = oy0 + qq, P = ox0 + pp;          // source block rows Q-1..Q, columns P-1..P
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            const int Y = 2 * Q - 1 + dy;
            // this tile owns output rows 2 oy0 .. 2 oy0 + 11 (and columns likewise)
            if (Y < 2 * oy0 || Y >= 2 * oy0 + 2 * BP_TR || Y >= Ho) continue;
            const float ly = dy ? 0.75f : 0.25f;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
              const int X = 2 * P - 1 + dx;
              if (X < 2 * ox0 || X >= 2 * ox0 + 2 * BP_TC || X >= Wo) continue;
              const float lx = dx ? 0.75f : 0.25f;
              const f32x4 v = (1.f - ly) * ((1.f - lx) * vaa + lx * vab) + ly * ((1.f - lx) * vba + lx * vbb);
              bp_f16x4 hv;
#pragma unroll
              for (int k = 0; k < 4; ++k) hv[k] = (_Float16)v[k];
              *reinterpret_cast<bp_f16x4*>(a.u16 + ((((long long)b * 4 + (cq >> 2)) * Ho + Y) * Wo + X) * 16 + (cq & 3) * 4) = hv;
            }
          }
        }
      } else {
        // 2x2 mean: 3 x 15 half-resolution pixels per tile (tile origin is even in both directions)
        for (int i = tid; i < 3 * 15 * 8; i += BP_THREADS) {
          const int q = i & 7, px = (i >> 3) % 15, py = (i >> 3) / 15;
          const int cq = nt * 8 + q;
          const int yd = (oy0 >> 1) + py, xd = (ox0 >> 1) + px;
          if (yd >= Hd || xd >= Wd) continue;
          const float* p0 = sY + ((1 + 2 * py) * 32 + 1 + 2 * px) * BP_YP + q * 4;      // halo offset 1
          const f32x4 v = 0.25f * ((*reinterpret_cast<const f32x4*>(p0) + *reinterpret_cast<const f32x4*>(p0 + BP_YP)) +
                                   (*reinterpret_cast<const f32x4*>(p0 + 32 * BP_YP) +
                                    *reinterpret_cast<const f32x4*>(p0 + 33 * BP_YP)));
          bp_f16x4 hv;
#pragma unroll
          for (int k = 0; k < 4; ++k) hv[k] = (_Float16)v[k];
          *reinterpret_cast<bp_f16x4*>(a.d16 + ((((long long)b * 4 + (cq >> 2)) * Hd + yd) * Wd + xd) * 16 + (cq & 3) * 4) = hv;
        }
      }
      __syncthreads();
    }
  }
}

}  // namespace

extern "C" int cdfo_block_prologue2(const float* x, int ldx, int B, int H, int W, const void* w_bf16, const float* bias128,
                                    void* u16, void* t16, void* d16, void* x16, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || ldx % 4 || ldx < 64 || (!u16 == !t16)) return CDFO_EINVAL;
  if ((long long)B * H * W * 4 >= (1ll << 31)) return CDFO_EINVAL;
  if (!aligned16(x) || !aligned16(w_bf16) || !aligned16(u16) || !aligned16(t16) || !aligned16(d16) || !aligned16(x16) || !bias128) return CDFO_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  static CdfoAttrOnce once;
  const hipError_t e = cdfo_set_max_lds(once, reinterpret_cast<const void*>(block_pro_kernel), BP_LDS);
  if (e != hipSuccess) return (int)e;
  const int cus = cdfo_num_cus();
  if (cus <= 0) return CDFO_EINVAL;
  const long long ntiles = (long long)B * cdiv(H, BP_TR) * cdiv(W, BP_TC);
  const int grid = (int)(ntiles < 2 * cus ? ntiles : 2 * cus);      // two workgroups per CU (LDS 2 x 65.5 KB, 120 VGPRs)
  bp_args a;
  a.x = x; a.ldx = ldx; a.B = B; a.H = H; a.W = W;
  a.w = static_cast<const unsigned short*>(w_bf16); a.bias = bias128;
  a.u16 = static_cast<_Float16*>(u16); a.t16 = static_cast<_Float16*>(t16); a.d16 = static_cast<_Float16*>(d16);
  a.x16 = static_cast<_Float16*>(x16);
  const double px = (double)B * H * W;
  CdfoProfScope prof(st, KID_RESAMPLE, 2.0 * px * 128 * 64, px * (4.0 * 64 + 2.0 * 64 * (u16 ? 4 : 1) + 2.0 * 16));
  hipLaunchKernelGGL(block_pro_kernel, dim3(grid), dim3(BP_THREADS), BP_LDS, st, a);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_block_prologue(const float* x, int ldx, int B, int H, int W, const void* w_bf16, const float* bias128,
                                   void* u16, void* d16, void* x16, void* stream) {
  return cdfo_block_prologue2(x, ldx, B, H, W, w_bf16, bias128, u16, nullptr, d16, x16, stream);
}
