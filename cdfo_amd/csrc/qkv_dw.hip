// MDTA front end in one pass: per-pixel LayerNorm -> 1x1 conv 64 -> 192 (qkv) -> depthwise 3x3 (qkv_dwconv)
// (arch/SIDECVSR_our.py:1169-1198 LayerNorm, :1551-1552 qkv / qkv_dwconv; Restormer's MDTA).
//
// As three kernels this stage writes the 192-channel pre-depthwise tensor (5.6 GB at 56 frames of 272x480), reads it
// back with a 1.8x over-fetch and writes 5.6 GB again: 18 GB of HBM traffic for 7.5 GB of algorithmic bytes.  Fused:
//   * persistent 512-thread workgroup per CU; the split-bf16 (hi | lo) 1x1 weights (49 KB) and the depthwise taps stay
//     in LDS for the whole launch;
//   * output tile = 6 rows x 30 pixels, so that its halo region is 8 rows x 32 pixels = exactly one 32-pixel MFMA M tile
//     per wave: wave w loads halo row w straight into A fragments (a lane holds 32 of its pixel's 64 channels, the
//     other 32 sit in lane ^ 32: one shuffle per LayerNorm statistic), normalises, splits to bf16 hi / lo and multiplies
//     by the weights (3 passes, fp32-grade) -- LayerNorm's gamma is folded into the weights and beta into a bias on
//     the host, so the per-pixel work is (x - mean) * rstd only;
//   * 64 output channels at a time: accumulators -> LDS y tile (zero outside the image = the depthwise conv's padding)
//     -> barrier -> each thread slides a 3x3 window down one (column, 4-channel) strip (24 LDS reads for 6 outputs)
//     -> 16-byte stores;
//   * the next tile's input rows are fetched into registers while the current tile is in its depthwise phase;
//   * optional fused Gram pass (a.gram != NULL): the attention only needs q and k through sum_p q[c] k[c'] (same head),
//     sum q^2 and sum k^2 (arch.py:1561-1566), so blocks 0 and 1 hold [32 q | 32 k] channels of the same four heads, the
//     depthwise threads exchange their k values inside the wave (ds_bpermute), accumulate the per-head 8x8 products in
//     registers over the tile's 6 rows, fold the 4 columns of a wave with shuffles, and the 8 waves' contributions are
//     summed in a fixed order into a 640-float LDS image (the layout of cdfo_gram_partial) that is stored to the
//     workgroup's own slot of gram[b] when the workgroup moves to another image -- no atomics anywhere, so the sums
//     (and the forward) are bit-reproducible: q and k never reach HBM (3.7 GB written + 3.7 GB re-read per round at 56
//     frames of 272x480).
#include "common.h"

namespace {

constexpr int QD_THREADS = 512;
constexpr int QD_TR = 6, QD_TC = 30;                      // output tile; halo region 8 x 32
constexpr int QD_W_BYTES = 2 * 4 * 2 * 192 * 16;          // [hi|lo][k-step][k-half][192 cout][8 bf16] = 49,152
constexpr int QD_YP = 68;                                 // floats per staged pixel (64 + 4: conflict-free b128 reads)
constexpr int QD_Y_BYTES = 8 * 32 * QD_YP * 4;            // 69,632
constexpr int QD_DW_OFF = QD_W_BYTES + QD_Y_BYTES;        // depthwise taps [9][192] floats, then bias [192]
constexpr int QD_G_OFF = QD_DW_OFF + 9 * 192 * 4 + 192 * 4; // Gram image: 640 floats
constexpr int QD_GW_OFF = QD_G_OFF + 640 * 4;              // per-wave Gram contributions of one block: [8][320] floats
constexpr int QD_LDS = QD_GW_OFF + 8 * 320 * 4;            // 139,264 bytes

typedef __bf16 qd_bf16x8 __attribute__((ext_vector_type(8)));

struct qd_args {
  const float* x; int ldx;
  int B, H, W;
  const unsigned short* w;      // bf16 [hi|lo][4][2][192][8]: W[n][c] * gamma[c], c = 16 s + 8 h + j
  const float* bias;            // [192]: W @ beta
  const float* dw;              // [192][9] depthwise taps
  float eps;
  float* out; int ldo;
  float* gram; int nslot;       // optional [B][nslot][640]: fused Gram pass; `out` then receives v only (64 channels)
};

__global__ __launch_bounds__(QD_THREADS) void qkv_dw_kernel(qd_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sY = reinterpret_cast<float*>(smem + QD_W_BYTES);
  float* sDw = reinterpret_cast<float*>(smem + QD_DW_OFF);
  float* sBias = sDw + 9 * 192;
  float* sG = reinterpret_cast<float*>(smem + QD_G_OFF);
  float* sGW = reinterpret_cast<float*>(smem + QD_GW_OFF);
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W;

  for (int i = tid; i < QD_W_BYTES / 16; i += QD_THREADS)
    reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(a.w)[i];
  for (int i = tid; i < 9 * 192; i += QD_THREADS) sDw[i] = a.dw[(i % 192) * 9 + i / 192];     // -> [tap][channel]
  for (int i = tid; i < 192; i += QD_THREADS) sBias[i] = a.bias ? a.bias[i] : 0.f;
  for (int i = tid; i < 640; i += QD_THREADS) sG[i] = 0.f;
  __syncthreads();

  const int tiles_x = (W + QD_TC - 1) / QD_TC, tiles_y = (H + QD_TR - 1) / QD_TR;
  const int ntiles = a.B * tiles_y * tiles_x;

  // halo row `wave` of tile t, this lane's pixel r, channels 16 s + 8 h .. + 7 (s = 0..3)
  f32x4 xr[8];
  bool xin = false;     // pixel inside the image
  auto load_x = [&](int t) {
    const int tx = t % tiles_x, t2 = t / tiles_x;
    const int ty = t2 % tiles_y, b = t2 / tiles_y;
    const int gy = ty * QD_TR - 1 + wave, gx = tx * QD_TC - 1 + r;
    xin = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const float* px = a.x + ((long long)(b * H + (xin ? gy : 0)) * W + (xin ? gx : 0)) * a.ldx + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      xr[2 * s] = *reinterpret_cast<const f32x4*>(px + 16 * s);
      xr[2 * s + 1] = *reinterpret_cast<const f32x4*>(px + 16 * s + 4);
    }
  };

  // contiguous tile range per workgroup (a workgroup then sees at most a few images: few Gram flushes)
  const int per = (ntiles + gridDim.x - 1) / gridDim.x;
  int t = blockIdx.x * per;
  const int t_end = (t + per) < ntiles ? (t + per) : ntiles;
  if (t < t_end) load_x(t);
  for (; t < t_end; ++t) {
    const int tx = t % tiles_x, t2 = t / tiles_x;
    const int ty = t2 % tiles_y, b = t2 / tiles_y;
    const int oy0 = ty * QD_TR, ox0 = tx * QD_TC;

    // ---- LayerNorm statistics (biased variance) and the split-bf16 A fragments of (x - mean) * rstd
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += (xr[i][0] + xr[i][1]) + (xr[i][2] + xr[i][3]);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.f / 64.f);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = xr[i][e] - mean; sq = fmaf(d, d, sq); }
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = rsqrtf(sq * (1.f / 64.f) + a.eps);
    qd_bf16x8 ah[4], al[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = (xr[2 * s + (j >> 2)][j & 3] - mean) * rstd;
        ah[s][j] = (__bf16)v;
        al[s][j] = (__bf16)(v - (float)ah[s][j]);
      }
    const bool pin = xin;                                   // this tile's pixel; xr / xin move on to the next tile
    const int tn = t + 1;
    if (tn < t_end) load_x(tn);

    // the 16 pixels of this lane's accumulator registers: halo column (e&3) + 8 (e>>2) + 4 h of halo row `wave`
    const int gyw = oy0 - 1 + wave;
    const bool row_in = gyw >= 0 && gyw < H;

#pragma unroll 1
    for (int nb = 0; nb < 3; ++nb) {
      // ---- y[halo pixel][64 channels of block nb] = x_hat W'^T + bias, zero outside the image
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        // transposed product (rows = output channels, columns = pixels): a lane's accumulator registers are 4 x 4
        // consecutive channels of ITS pixel (halo column r), so the y tile is written with four 16-byte stores per lane
        // and the inside-the-image test is one per lane
        const int n0 = nb < 2 ? nt * 64 + nb * 32 : 128 + nt * 32;            // blocks 0, 1: [32 q | 32 k] of the same heads
        const int n = n0 + r;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const qd_bf16x8 wh = *reinterpret_cast<const qd_bf16x8*>(smem + ((s * 2 + h) * 192 + n) * 16);
          const qd_bf16x8 wl = *reinterpret_cast<const qd_bf16x8*>(smem + QD_W_BYTES / 2 + ((s * 2 + h) * 192 + n) * 16);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al[s], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah[s], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah[s], acc, 0, 0, 0);
        }
        const int gxr = ox0 - 1 + r;
        const bool in = row_in && gxr >= 0 && gxr < W;
        float* yrow = sY + (wave * 32 + r) * QD_YP + nt * 32 + 4 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                      // channels n0 + 8j + 4h .. + 3
          const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + n0 + 8 * j + 4 * h);
          f32x4 v = {acc[4 * j] + bv[0], acc[4 * j + 1] + bv[1], acc[4 * j + 2] + bv[2], acc[4 * j + 3] + bv[3]};
          if (!in) v = f32x4{0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(yrow + 8 * j) = v;
        }
      }
      __syncthreads();
      // ---- depthwise 3x3: thread = (column x, 4 channels q*4..), sliding down the 6 output rows
      const bool gram_blk = a.gram != nullptr && nb < 2;
      // Gram products of a head (8 q x 8 k channels) are shared by its four lanes: the q lanes L0 / L1 (channels 0-3 /
      // 4-7) take q x k against the k lanes M0 / M1 with the same index, M0 / M1 take k x q against L1 / L0
      float g[4][4], s2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s2[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) g[i][j] = 0.f;
      }
      const int q = tid & 15, x = tid >> 4;
      const int cb = nb < 2 ? ((q & 8) ? 64 : 0) + nb * 32 + (q & 7) * 4 : 128 + q * 4;
      if (tid < QD_TC * 16) {
        f32x4 wt[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wt[k] = *reinterpret_cast<const f32x4*>(sDw + k * 192 + cb);
        f32x4 win[3][3];
#pragma unroll
        for (int ry = 0; ry < 2; ++ry)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            win[ry][dx] = *reinterpret_cast<const f32x4*>(sY + (ry * 32 + x + dx) * QD_YP + q * 4);
        const bool xok = ox0 + x < W;
        const int src0 = (lane & 48) | (q < 8 ? 8 + q : (q & 7) ^ 1);   // partner lane, same column
#pragma unroll
        for (int y = 0; y < QD_TR; ++y) {
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            win[(y + 2) % 3][dx] = *reinterpret_cast<const f32x4*>(sY + ((y + 2) * 32 + x + dx) * QD_YP + q * 4);
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) o += wt[dy * 3 + dx] * win[(y + dy) % 3][dx];
          const bool ok = xok && oy0 + y < H;
          if (gram_blk) {
            if (!ok) o = f32x4{0.f, 0.f, 0.f, 0.f};
            float pv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[e] = __shfl(o[e], src0, 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              s2[i] = fmaf(o[i], o[i], s2[i]);
#pragma unroll
              for (int j = 0; j < 4; ++j) g[i][j] = fmaf(o[i], pv[j], g[i][j]);
            }
          } else if (ok) {
            const int co = a.gram ? q * 4 : cb;                      // Gram mode: the output tensor holds v only
            *reinterpret_cast<f32x4*>(a.out + ((long long)(b * H + oy0 + y) * W + ox0 + x) * a.ldo + co) = o;
          }
        }
      }
      if (gram_blk) {                                                // all 512 threads: idle ones carry zeros
        // fold the wave's 4 columns, then one LDS add per (channel, partner) from the 16 lanes of column 0
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          s2[i] += __shfl_xor(s2[i], 16, 64);
          s2[i] += __shfl_xor(s2[i], 32, 64);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            g[i][j] += __shfl_xor(g[i][j], 16, 64);
            g[i][j] += __shfl_xor(g[i][j], 32, 64);
          }
        }
        if (lane < 16) {
          float* slot = sGW + wave * 320;
          const int m = q & 7;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (q < 8) {                                             // own q channel m*4+i  x  k channels 4*(m&1) .. +3
#pragma unroll
              for (int j = 0; j < 4; ++j) slot[(m * 4 + i) * 9 + 4 * (m & 1) + j] = g[i][j];
              slot[(m * 4 + i) * 9 + 8] = s2[i];
            } else {                                                 // partner's q channel (m^1)*4+j  x  own k channel 4*(m&1)+i
#pragma unroll
              for (int j = 0; j < 4; ++j) slot[((m ^ 1) * 4 + j) * 9 + 4 * (m & 1) + i] = g[i][j];
              slot[288 + m * 4 + i] = s2[i];
            }
          }
        }
      }
      __syncthreads();
      if (gram_blk && tid < 320) {                                   // fixed summation order over the 8 waves
        float sum = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) sum += sGW[wv * 320 + tid];
        const int idx = tid < 288 ? (nb * 32 + tid / 9) * 10 + tid % 9 : (nb * 32 + tid - 288) * 10 + 9;
        sG[idx] += sum;
      }
    }
    (void)pin;
    if (a.gram) {
      const int tn2 = t + 1;
      const int bn = tn2 < t_end ? (tn2 / tiles_x) / tiles_y : -1;
      if (bn != b) {                                                  // leaving image b: store its Gram sums to this
        const int first = (b * tiles_y * tiles_x) / per;              // workgroup's slot (workgroups cover contiguous ranges)
        float* dst = a.gram + ((long long)b * a.nslot + (blockIdx.x - first)) * 640;
        for (int i = tid; i < 640; i += QD_THREADS) {
          dst[i] = sG[i];
          sG[i] = 0.f;
        }
        __syncthreads();
      }
    }
  }
}

}  // namespace

static int qd_cu_count() { return cdfo_num_cus(); }     // of the current device

// Number of per-image slots the fused Gram pass needs for these shapes on this device (>= 1).
extern "C" int cdfo_qkv_dw_gram_slots(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return CDFO_EINVAL;
  const int cus = qd_cu_count();
  if (cus <= 0) return CDFO_EINVAL;
  const long long tpi = (long long)cdiv(H, QD_TR) * cdiv(W, QD_TC), ntiles = tpi * B;
  const long long grid = ntiles < cus ? ntiles : cus;
  const long long per = (ntiles + grid - 1) / grid;
  return (int)((tpi + per - 1) / per + 1);
}

extern "C" int cdfo_qkv_dw(const float* x, int ldx, int B, int H, int W, const void* w_bf16, const float* bias,
                           const float* dw_w, float eps, float* out, int ldo, float* gram, int gram_slots, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || ldx % 4 || ldx < 64 || ldo % 4 || ldo < (gram ? 64 : 192)) return CDFO_EINVAL;
  if ((long long)B * H * W >= (1ll << 31)) return CDFO_EINVAL;
  if (!aligned16(x) || !aligned16(w_bf16) || !aligned16(out) || !dw_w) return CDFO_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  static CdfoAttrOnce once;
  const hipError_t e = cdfo_set_max_lds(once, reinterpret_cast<const void*>(qkv_dw_kernel), QD_LDS);
  if (e != hipSuccess) return (int)e;
  const int cus = qd_cu_count();
  if (cus <= 0) return CDFO_EINVAL;
  const long long ntiles = (long long)B * cdiv(H, QD_TR) * cdiv(W, QD_TC);
  const int grid = (int)(ntiles < cus ? ntiles : cus);
  qd_args a;
  a.x = x; a.ldx = ldx; a.B = B; a.H = H; a.W = W;
  a.w = static_cast<const unsigned short*>(w_bf16); a.bias = bias; a.dw = dw_w; a.eps = eps; a.out = out; a.ldo = ldo; a.gram = gram; a.nslot = gram_slots;
  if (gram) {
    const long long per = (ntiles + grid - 1) / grid, tpi = ntiles / B;
    if (gram_slots < (int)((tpi + per - 1) / per + 1)) return CDFO_EINVAL;
  }
  const double px = (double)B * H * W;
  CdfoProfScope prof(st, KID_DWCONV, 2.0 * px * 192 * (64 + 9) + (gram ? 2.0 * px * 640 : 0.0), 4.0 * px * (64 + (gram ? 64 : 192)));
  hipLaunchKernelGGL(qkv_dw_kernel, dim3(grid), dim3(QD_THREADS), QD_LDS, st, a);
  CDFO_LAUNCH_CHECK();
  return 0;
}
