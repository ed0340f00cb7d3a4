// 3x3 stride-1 convolution 64 -> 16 channels, fp32 pixel-major in and out, split-bf16 arithmetic (fp32-grade).
//
// The prior U-net of the feature extractor opens with Conv2d(64, 16, 3, 1, 1) + LeakyReLU (arch/SIDECVSR_our.py:1815-1834, body.0)
// on all 56 frames of a batch, twice per forward.  On the tiled 16-bit kernel (conv3x3_bf16.hip) its 16 output channels ride on a
// 64-wide tile -- three quarters of the matrix work is masked away in the epilogue -- and the launch runs at 1.9 TB/s of its
// algorithmic bytes (1.24 ms for 2.34 GB).  This kernel does the 16 channels only, on v_mfma_f32_16x16x32_bf16:
//   * M = the 16 output channels, N = 16 pixels, K = 32 input channels of one tap per MFMA: every lane of every MFMA is useful;
//     a_hi*w_hi + a_lo*w_hi + a_hi*w_lo (the tiled kernel's "bf16x3" arithmetic), fp32 accumulation;
//   * one 512-thread workgroup per CU is persistent over 8 x 32-pixel tiles.  A tile's 10 x 34-pixel halo is fetched into
//     REGISTERS one tile ahead (11 float4 per thread), split to bf16 hi | lo and written to LDS as 256-byte pixel records
//     [8 hi chunks | 8 lo chunks of 8 channels], the 16-byte chunk index XORed with the pixel index so that the fragment reads
//     (16 consecutive pixels, one chunk) are conflict-free;
//   * the weights (18 K steps x hi | lo x 1 KiB, packed on the host in MFMA lane order) stay in LDS for the whole launch (in
//     registers -- 144 VGPRs beside the 44 of the halo prefetch -- hipcc spills and the launch is 70 % slower: measured);
//   * a wave owns one tile row = two 16-pixel MFMA columns; its accumulator registers are 4 consecutive output channels of one
//     pixel: the epilogue (bias, activation) stores 16 bytes per lane, 1 KiB contiguous per wave-instruction.
#include "common.h"

namespace {

constexpr int N16_TR = 8, N16_TC = 32;                 // output tile
constexpr int N16_HR = N16_TR + 2, N16_HC = N16_TC + 2, N16_NPIX = N16_HR * N16_HC;      // 340 halo pixels
constexpr int N16_THREADS = 512;
constexpr int N16_W_BYTES = 18 * 2 * 1024;             // [K step = tap*2 + half][hi | lo][64 lanes][8 bf16]
constexpr int N16_A_BYTES = N16_NPIX * 256;            // 87,040
constexpr int N16_LDS = N16_W_BYTES + N16_A_BYTES;     // 123,904
constexpr int N16_ITEMS = N16_NPIX * 16;               // float4 items of a halo tile
constexpr int N16_NL = (N16_ITEMS + N16_THREADS - 1) / N16_THREADS;                      // 11 loads per thread

typedef __bf16 n16_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned n16_u32x2 __attribute__((ext_vector_type(2)));

struct n16_args {
  const float* x; int ldx;
  int B, H, W;
  const unsigned short* w;       // packed weights, N16_W_BYTES
  const float* bias;             // [16] or nullptr
  int act;
  float* out; int ldo;
  int tiles_x, tiles_y, ntiles;
};

__device__ __forceinline__ unsigned n16_pack(float a, float b) {
  const __bf16 ha = (__bf16)a, hb = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
}
__device__ __forceinline__ float n16_round(float a) { return (float)(__bf16)a; }

__global__ __launch_bounds__(N16_THREADS) void conv3x3_c64_n16_kernel(n16_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sW = smem;
  unsigned char* const sA = smem + N16_W_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W;

  for (int i = tid; i < N16_W_BYTES / 16; i += N16_THREADS)
    reinterpret_cast<f32x4*>(sW)[i] = reinterpret_cast<const f32x4*>(a.w)[i];
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + 4 * kg);

  // ---- register prefetch of a tile's halo: item = tid + 512 k -> pixel item >> 4 (row-major inside the halo), float4 q = item & 15
  f32x4 pre[N16_NL];
  unsigned pre_ok = 0;          // bit k: item k lies inside the image.  Applied when the tile is STAGED: a select right behind the
                                // load would make the wave wait for the data there and the prefetch would cover nothing
  // tile index t = (b * tiles_x + tx) * tiles_y + ty: a workgroup's contiguous range walks DOWN a 32-pixel column of an image, so
  // the two halo rows a tile shares with its predecessor were fetched by this CU one tile ago (L2 hits instead of HBM re-reads)
  auto fetch = [&](int t) {               // (32-bit tile arithmetic: a 64-bit division by a run-time value is a ~130-instruction loop)
    const int t2 = t / a.tiles_y, ty = t - t2 * a.tiles_y;
    const int b = t2 / a.tiles_x, tx = t2 - b * a.tiles_x;
    const int y0 = ty * N16_TR - 1, x0 = tx * N16_TC - 1;
    pre_ok = 0;
#pragma unroll
    for (int k = 0; k < N16_NL; ++k) {
      const int item = tid + N16_THREADS * k, p = item >> 4, q = item & 15;
      const int iy = p / N16_HC, ix = p - iy * N16_HC;
      const int gy = y0 + iy, gx = x0 + ix;
      const bool ok = item < N16_ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int pix = (b * H + (ok ? gy : 0)) * W + (ok ? gx : 0);
      pre[k] = *reinterpret_cast<const f32x4*>(a.x + (long long)pix * a.ldx + q * 4);             // always a valid address; zeroed when staged
      pre_ok |= ok ? (1u << k) : 0u;
    }
  };

  int t = (int)((long long)a.ntiles * blockIdx.x / gridDim.x);
  const int t_end = (int)((long long)a.ntiles * (blockIdx.x + 1) / gridDim.x);
  if (t < t_end) fetch(t);
  __syncthreads();                                            // the weight image is complete
  for (; t < t_end; ++t) {
    // ---- stage: split to bf16 hi | lo, 8-byte LDS writes into the swizzled pixel records
#pragma unroll
    for (int k = 0; k < N16_NL; ++k) {
      const int item = tid + N16_THREADS * k, p = item >> 4, q = item & 15;
      if (item < N16_ITEMS) {
        const f32x4 v = (pre_ok >> k) & 1u ? pre[k] : f32x4{0.f, 0.f, 0.f, 0.f};
        n16_u32x2 hi, lo;
        hi[0] = n16_pack(v[0], v[1]); hi[1] = n16_pack(v[2], v[3]);
        lo[0] = n16_pack(v[0] - n16_round(v[0]), v[1] - n16_round(v[1]));
        lo[1] = n16_pack(v[2] - n16_round(v[2]), v[3] - n16_round(v[3]));
        const int chunk = q >> 1, sw = p & 15;
        unsigned char* rec = sA + p * 256 + (q & 1) * 8;
        *reinterpret_cast<n16_u32x2*>(rec + ((chunk ^ sw) << 4)) = hi;
        *reinterpret_cast<n16_u32x2*>(rec + (((8 + chunk) ^ sw) << 4)) = lo;
      }
    }
    const int tn = t + 1;
    const int t2 = t / a.tiles_y, ty = t - t2 * a.tiles_y;
    const int b = t2 / a.tiles_x, tx = t2 - b * a.tiles_x;
    __syncthreads();                                          // the tile's records are in LDS
    if (tn < t_end) fetch(tn);                             // next tile's loads fly while this one is multiplied

    // ---- this wave's row: output pixels (wave, 16 c + n), c = 0, 1
    f32x4 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[c] = bias4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int s = tap * 2 + half;
        const n16_bf16x8 whs = *reinterpret_cast<const n16_bf16x8*>(sW + (s * 2 + 0) * 1024 + lane * 16);
        const n16_bf16x8 wls = *reinterpret_cast<const n16_bf16x8*>(sW + (s * 2 + 1) * 1024 + lane * 16);
        const int chunk = 4 * half + kg;                      // channels 32 half + 8 kg .. + 7
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int p = (wave + dy) * N16_HC + 16 * c + n + dx, sw = p & 15;
          const n16_bf16x8 ah = *reinterpret_cast<const n16_bf16x8*>(sA + p * 256 + ((chunk ^ sw) << 4));
          const n16_bf16x8 al = *reinterpret_cast<const n16_bf16x8*>(sA + p * 256 + (((8 + chunk) ^ sw) << 4));
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whs, al, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wls, ah, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whs, ah, acc[c], 0, 0, 0);
        }
      }
    }
    // ---- epilogue: acc[c][e] = channel 4 kg + e of pixel (ty*8 + wave, tx*32 + 16 c + n)
    const int oy = ty * N16_TR + wave;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ox = tx * N16_TC + 16 * c + n;
      if (oy < H && ox < W) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(acc[c][e], a.act);
        *reinterpret_cast<f32x4*>(a.out + (((long long)b * H + oy) * W + ox) * a.ldo + 4 * kg) = v;
      }
    }
    __syncthreads();                                          // every wave is done reading before the next tile is staged
  }
}

}  // namespace

extern "C" int cdfo_conv3x3_c64_n16(const float* x, int ldx, int B, int H, int W, const void* w_packed, const float* bias, int act,
                                    float* out, int ldo, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || ldx < 64 || ldx % 4 || ldo < 16 || ldo % 4 || act == CDFO_ACT_SIGMOID) return CDFO_EINVAL;
  if (!aligned16(x) || !aligned16(w_packed) || !aligned16(out) || (bias && !aligned16(bias))) return CDFO_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  static CdfoAttrOnce once;
  const hipError_t e = cdfo_set_max_lds(once, reinterpret_cast<const void*>(conv3x3_c64_n16_kernel), N16_LDS);
  if (e != hipSuccess) return (int)e;
  const int cus = cdfo_num_cus();
  if (cus <= 0) return CDFO_EINVAL;
  n16_args a;
  a.x = x; a.ldx = ldx; a.B = B; a.H = H; a.W = W;
  a.w = static_cast<const unsigned short*>(w_packed); a.bias = bias; a.act = act;
  a.out = out; a.ldo = ldo;
  a.tiles_x = cdiv(W, N16_TC); a.tiles_y = cdiv(H, N16_TR);
  const long long nt = (long long)B * a.tiles_x * a.tiles_y;
  if (nt >= (1ll << 31) || (long long)B * H * W >= (1ll << 31)) return CDFO_EINVAL;
  a.ntiles = (int)nt;
  const int grid = a.ntiles < cus ? a.ntiles : cus;
  const double px = (double)B * H * W;
  CdfoProfScope prof(st, KID_CONV3_NARROW, 2.0 * px * 9 * 64 * 16, 4.0 * px * (64 + 16) + 9.0 * 64 * 16 * 4);
  hipLaunchKernelGGL(conv3x3_c64_n16_kernel, dim3(grid), dim3(N16_THREADS), N16_LDS, st, a);
  CDFO_LAUNCH_CHECK();
  return 0;
}
