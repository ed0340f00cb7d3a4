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
  unsigned long long* clk;      // developer timeline (cdfo_qkv_dw_probe): 8 s_memtime stamps per wave of one steady-state step
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

// ---------------------------------------------------------------------------------------------------------------------------
// Round 3: the Gram form as a ROW-STREAMING kernel (qkv_dw2).  The tiled kernel above recomputes the 1x1 convolution on an
// 8 x 32 halo for every 6 x 30 outputs (1.42x), holds 139 KB of LDS (one 8-wave workgroup per CU), walks three 64-channel
// blocks per tile with two barriers each, and spends 66 ds_bpermute per block on the Gram exchange: 2.4 ms at 56 frames of
// 272x480 for 3.7 GB of traffic (1.6 TB/s), 1.49 ms of it the load / normalise / barrier skeleton.  Here:
//   * a persistent 512-thread workgroup owns (image, ~55-row segment, 30-column strip) units and walks DOWN the strip one image
//     row per step: only the two segment-boundary rows are recomputed (1.04x vertically, 32 / 30 horizontally);
//   * per step every wave first PRODUCES its share of the new halo row -- LayerNorm of a 16-pixel tile, split-bf16
//     v_mfma_f32_16x16x32_bf16 against three 16-channel weight tiles that stay in its REGISTERS for the whole launch (48
//     VGPRs; the weights never touch LDS) -- into a four-row ring of y rows in LDS (100 KB), then, after the step's ONE
//     barrier, CONSUMES: the depthwise 3x3 as running column sums (a new input row completes output row r-1, extends r and
//     starts r+1: two partial sums per pixel instead of a 3x3 window), five ds_read_b128 per thread and row for three pixels;
//   * wave w = attention head w; lanes = 6 channel groups (q lo | q hi | k lo | k hi | v lo | v hi of the head) x 10 pixel
//     triples, so one thread keeps ONE group's nine depthwise taps (36 VGPRs) and 20 Gram accumulators summed over all its
//     pixels; the q / k exchange is 12 ds_bpermute per row between lanes of the same wave;
//   * the Gram sums leave through the same per-workgroup slots as before (fixed-order reduction over the ten pixel lanes via
//     LDS when the workgroup changes image: no atomics, bit-reproducible).
constexpr int Q2_THREADS = 512;
constexpr int Q2_TC = 30, Q2_HC = 32;                       // output columns per strip, halo columns
constexpr int Q2_YP = 196;                                  // floats per staged pixel: 192 + 4 (conflict-free b128 column reads)
constexpr int Q2_ROWB = Q2_HC * Q2_YP * 4;                  // 25,088 bytes per y row
constexpr int Q2_RING = 4;
constexpr int Q2_FPX = 288, Q2_FROW = Q2_HC * Q2_FPX;       // x_hat fragments: per pixel 64 bf16 hi | 64 bf16 lo (+ 16 B pad each)
constexpr int Q2_VPX = 68, Q2_VROW = Q2_HC * Q2_VPX * 4;         // v staging: per pixel 64 floats (+ 4 pad), 32 pixels
constexpr int Q2_F_OFF = Q2_RING * Q2_ROWB, Q2_V_OFF = Q2_F_OFF + 2 * Q2_FROW;
constexpr int Q2_LDS = Q2_V_OFF + 2 * Q2_VROW;              // 100,352 + 18,432 + 18,432 (the Gram flush reuses the ring as scratch)

typedef __bf16 q2_bf16x8 __attribute__((ext_vector_type(8)));

struct q2_geom { int strips, nseg, rs, upi; };              // strips per row, segments per image, rows per segment, units per image
__host__ __device__ inline q2_geom q2_geometry(int H, int W) {
  q2_geom g;
  g.strips = (W + Q2_TC - 1) / Q2_TC;
  g.nseg = H > 80 ? (H + 55) / 56 : 1;
  g.rs = (H + g.nseg - 1) / g.nseg;
  g.upi = g.strips * g.nseg;
  return g;
}

// sum over the 16 lanes of a DPP row (all lanes get the total): VALU cross-lane modifiers, no LDS round trip per step
__device__ __forceinline__ float q2_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

template <bool TL = false>      // TL: timeline stamps (developer probe)
__global__ __launch_bounds__(Q2_THREADS) void qkv_dw2_kernel(qd_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const ring = reinterpret_cast<float*>(smem);
  unsigned char* const sF = smem + Q2_F_OFF;                 // [2 rows][32 pixels][Q2_FPX]
  float* const sV = reinterpret_cast<float*>(smem + Q2_V_OFF);     // [2 rows][32 pixels][Q2_VPX]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W;
  const q2_geom G = q2_geometry(H, W);
  const int units = a.B * G.upi;
  const int per = (units + gridDim.x - 1) / gridDim.x;
  int u = blockIdx.x * per;
  const int u_end = (u + per) < units ? (u + per) : units;
  if (u >= u_end) return;

  // ---- producer role: 16-pixel tile (wave & 1) x three 16-channel tiles 3 (wave >> 1) + j; lane = (pixel n, k group kg)
  const int pn = lane & 15, kg = lane >> 4, pxt = wave & 1, ct0 = 3 * (wave >> 1);
  q2_bf16x8 Wh[3][2], Wl[3][2];                              // A operands: W'[16 ct + n][32 s + 8 kg + j], hi | lo (see pack_qkv_dw)
  f32x4 pbias[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int m = 16 * (ct0 + j) + pn;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int s16 = 2 * s2 + (kg >> 1), hh = kg & 1;
      Wh[j][s2] = *reinterpret_cast<const q2_bf16x8*>(a.w + (((0 * 4 + s16) * 2 + hh) * 192 + m) * 8);
      Wl[j][s2] = *reinterpret_cast<const q2_bf16x8*>(a.w + (((1 * 4 + s16) * 2 + hh) * 192 + m) * 8);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) pbias[j][e] = a.bias ? a.bias[16 * (ct0 + j) + 4 * kg + e] : 0.f;
  }
  // ---- consumer role: head = wave; lane = 10 gi + pb: channel group gi (0 q lo, 1 q hi, 2 k lo, 3 k hi, 4 v lo, 5 v hi), pixels
  // 3 pb .. 3 pb + 2 of the strip
  const int gi = lane / 10, pb = lane - gi * 10;
  const bool c_on = lane < 60;
  const int cg = gi < 2 ? 8 * wave + 4 * gi : (gi < 4 ? 64 + 8 * wave + 4 * (gi - 2) : 128 + 8 * wave + 4 * (gi - 4));   // first channel
  f32x4 wt[9];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) wt[k][e] = c_on ? a.dw[(cg + e) * 9 + k] : 0.f;
  const int partner = gi == 0 ? 20 + pb : (gi == 1 ? 30 + pb : (gi == 2 ? 10 + pb : pb));      // L0<-M0, L1<-M1, M0<-L1, M1<-L0
  float gacc[4][4], s2acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s2acc[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) gacc[i][j] = 0.f;
  }

  int b_cur = u / G.upi;
  auto flush = [&](int b) {       // Gram sums of image b -> this workgroup's slot; the ring is idle (callers sit between units)
    __syncthreads();
    float* scr = ring + (wave * 64 + lane) * 20;
    if (lane < 40) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) scr[i * 4 + j] = gacc[i][j];
        scr[16 + i] = s2acc[i];
      }
    }
    __syncthreads();
    const int first = (b * G.upi) / per;
    float* dst = a.gram + ((long long)b * a.nslot + (blockIdx.x - first)) * 640;
    for (int t = lane; t < 80; t += 64) {
      const int g4 = t / 20, e = t - g4 * 20;
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 10; ++q) sum += ring[(wave * 64 + g4 * 10 + q) * 20 + e];      // fixed order over the pixel lanes
      int idx;
      if (e < 16) {
        const int i = e >> 2, j = e & 3;
        idx = g4 == 0 ? (8 * wave + i) * 10 + j : (g4 == 1 ? (8 * wave + 4 + i) * 10 + 4 + j
              : (g4 == 2 ? (8 * wave + 4 + j) * 10 + i : (8 * wave + j) * 10 + 4 + i));
      } else {
        const int i = e - 16;
        idx = g4 == 0 ? (8 * wave + i) * 10 + 8 : (g4 == 1 ? (8 * wave + 4 + i) * 10 + 8
              : (g4 == 2 ? (8 * wave + i) * 10 + 9 : (8 * wave + 4 + i) * 10 + 9));
      }
      dst[idx] = sum;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s2acc[i] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) gacc[i][j] = 0.f;
    }
    __syncthreads();
  };

  for (; u < u_end; ++u) {
    const int b = u / G.upi, r0 = u - b * G.upi;
    const int sg = r0 / G.strips, sx = r0 - sg * G.strips;
    if (b != b_cur) { flush(b_cur); b_cur = b; }
    __syncthreads();            // the previous unit's last consumer reads of the ring are done before this unit's first row lands
    const int ya = sg * G.rs, yb = (ya + G.rs) < H ? (ya + G.rs) : H;       // output rows [ya, yb)
    const int x0 = sx * Q2_TC;
    // ---- per-lane roles of this unit
    // (a) LayerNorm: wave w normalises halo pixels 4 w .. 4 w + 3; lane = (pixel, channel quad): ONE 16-byte load per lane and row,
    //     each pixel of the row is read from memory exactly once per workgroup
    const int lpx = 4 * wave + (lane >> 4), lcq = lane & 15, lgx = x0 - 1 + lpx;
    const bool lcol_in = lgx >= 0 && lgx < W;
    const float* xcol = a.x + ((long long)b * H * W + (lcol_in ? lgx : 0)) * a.ldx + 4 * lcq;
    auto load_x = [&](int gy) -> f32x4 {
      const int yy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
      return *reinterpret_cast<const f32x4*>(xcol + (long long)yy * W * a.ldx);
    };
    unsigned char* const fr_w = sF + lpx * Q2_FPX + lcq * 8;             // this lane's slot of the fragment row (hi; lo at + Q2_FPX / 2)
    // (b) matrix stage: halo pixel 16 pxt + pn; B fragments from the fragment row, results into the y ring
    const int hx = 16 * pxt + pn, gx = x0 - 1 + hx;
    const bool col_in = gx >= 0 && gx < W;
    const unsigned char* const fr_r = sF + hx * Q2_FPX + 16 * kg;
    // (d) consumer: pixels 3 pb .. 3 pb + 2, channel group cg;  (e) store: pixel tid >> 4, 16-byte part tid & 15
    const int ox = x0 + 3 * pb;
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(a.out + (long long)b * H * W * a.ldo, 0, H * W * a.ldo * 4, 0x00020000);
    const int spx = tid >> 4, sq16 = tid & 15;
    f32x4 S1[3], S2[3];                                     // running depthwise sums of output rows r-1 and r (per pixel)
#pragma unroll
    for (int i = 0; i < 3; ++i) S1[i] = S2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the x rows are requested two steps ahead of their LayerNorm (two registers that alternate; one step is ~2 us, about an HBM
    // round trip under load)
    f32x4 xa = load_x(ya - 1), xb = load_x(ya);
    // One step = image row hy through all stages, software-pipelined over the rows with ONE barrier:
    //   (a) LayerNorm of row hy + 1 -> fragment row (hy + 1) & 1        (b) matrix product of row hy (fragment row hy & 1) -> y ring
    //   barrier   (d) depthwise + Gram: row hy completes output row hy - 1; v -> staging row (hy - 1) & 1
    //   (e) coalesced store of v row hy - 2 from staging row (hy - 2) & 1
    // Every stage runs on every step (rows outside [ya - 1, yb] are clamped reads / dropped stores / unused results), so that the
    // instruction stream -- and with it the compiler's count of loads and stores in flight -- is the same on every trip.
    auto step = [&](int hy, f32x4& xr) {      // xr holds row hy + 1's pixels and receives row hy + 3's
      const int slot = (hy - (ya - 2)) & (Q2_RING - 1);
      unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      auto stamp = [&](int k) {
        if (TL) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[k]) : : "memory");
      };
      stamp(0);
      {   // (a)
        const float mean = q2_sum16((xr[0] + xr[1]) + (xr[2] + xr[3])) * (1.f / 64.f);
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xr[e] - mean; sq = fmaf(d, d, sq); }
        const float rstd = rsqrtf(q2_sum16(sq) * (1.f / 64.f) + a.eps);
        typedef __bf16 q2_bf16x4 __attribute__((ext_vector_type(4)));
        q2_bf16x4 vh, vl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = (xr[e] - mean) * rstd;
          vh[e] = (__bf16)v;
          vl[e] = (__bf16)(v - (float)vh[e]);
        }
        xr = load_x(hy + 3);
        unsigned char* dst = fr_w + ((hy + 1) & 1) * Q2_FROW;
        *reinterpret_cast<q2_bf16x4*>(dst) = vh;
        *reinterpret_cast<q2_bf16x4*>(dst + Q2_FPX / 2) = vl;
      }
      stamp(1);
      {   // (b): zero outside the image = the depthwise convolution's padding
        const unsigned char* src = fr_r + (hy & 1) * Q2_FROW;
        q2_bf16x8 bh[2], bl[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bh[s2] = *reinterpret_cast<const q2_bf16x8*>(src + 64 * s2);
          bl[s2] = *reinterpret_cast<const q2_bf16x8*>(src + Q2_FPX / 2 + 64 * s2);
        }
        const bool in = col_in && hy >= 0 && hy < H;
        float* yrow = ring + slot * (Q2_ROWB / 4) + hx * Q2_YP + 4 * kg;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wh[j][s2], bl[s2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wl[j][s2], bh[s2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wh[j][s2], bh[s2], acc, 0, 0, 0);
          }
          f32x4 v = acc + pbias[j];
          if (!in) v = f32x4{0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(yrow + 16 * (ct0 + j)) = v;
        }
      }
      stamp(2);
      __syncthreads();
      stamp(3);
      {   // (d): input row hy completes output row hy - 1
        const float* yr = ring + slot * (Q2_ROWB / 4) + (3 * pb) * Q2_YP + cg;
        f32x4 col[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) col[c] = *reinterpret_cast<const f32x4*>(yr + c * Q2_YP);
        stamp(4);
        const int oy = hy - 1;
        const bool row_ok = oy >= ya && oy < yb;
        f32x4 o[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const f32x4 h0 = wt[0] * col[i] + wt[1] * col[i + 1] + wt[2] * col[i + 2];
          const f32x4 h1 = wt[3] * col[i] + wt[4] * col[i + 1] + wt[5] * col[i + 2];
          const f32x4 h2 = wt[6] * col[i] + wt[7] * col[i + 1] + wt[8] * col[i + 2];
          o[i] = S1[i] + h2;
          S1[i] = S2[i] + h1;
          S2[i] = h0;
          if (!(row_ok && c_on && ox + i < W)) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // v -> staging row (LDS writes under a lane predicate are harmless to the vmcnt bookkeeping, unlike global stores)
        if (gi >= 4 && c_on) {
          float* vs = sV + (oy & 1) * (Q2_VROW / 4) + (3 * pb) * Q2_VPX + (cg - 128);
#pragma unroll
          for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(vs + i * Q2_VPX) = o[i];
        }
        stamp(5);
        // Gram: q lanes take own q x the partner's k, k lanes own k x the partner's q (16 products + 4 squares per lane, summed
        // over the lane's three pixels); v lanes run the exchange with their own values and skip the sums
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          float pv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) pv[e] = __shfl(o[i][e], partner, 64);
          if (gi < 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              s2acc[q] = fmaf(o[i][q], o[i][q], s2acc[q]);
#pragma unroll
              for (int j = 0; j < 4; ++j) gacc[q][j] = fmaf(o[i][q], pv[j], gacc[q][j]);
            }
          }
        }
      }
      stamp(6);
      {   // (e): v row hy - 2, 256 contiguous bytes per pixel (16 lanes), dropped by the hardware where there is nothing to write
        const int oy = hy - 2;
        const f32x4 v = *reinterpret_cast<const f32x4*>(sV + (oy & 1) * (Q2_VROW / 4) + spx * Q2_VPX + 4 * sq16);
        const bool st = spx < Q2_TC && x0 + spx < W && oy >= ya && oy < yb;
        const unsigned vo = st ? (unsigned)(((oy * W + x0 + spx) * a.ldo + 4 * sq16) * 4) : 0x80000000u;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out, (int)vo, 0, 0);
      }
      if (TL) {
        stamp(7);
        if (a.clk && u == blockIdx.x * per && hy == ya + 24 && lane == 0) {
          unsigned long long* c = a.clk + ((long long)blockIdx.x * 8 + wave) * 8;
#pragma unroll
          for (int k = 0; k < 8; ++k) c[k] = ts[k];
        }
      }
    };
    // two rows per trip, so that the two x registers swap roles without copies (a copy would wait for the load it renames); a
    // surplus row at either end lands in the ring and completes no valid output
    for (int hy = ya - 2; hy <= yb + 1; hy += 2) {
      step(hy, xa);
      step(hy + 1, xb);
    }
  }
  flush(b_cur);
}

}  // namespace

static int qd_cu_count() { return cdfo_num_cus(); }     // of the current device

// developer probe: a device buffer of [workgroups][8 waves][8] uint64 for the row-streaming kernel's timeline (tools/qkv_timeline.py);
// NULL switches it off again
static unsigned long long* g_q2_clk = nullptr;
extern "C" void cdfo_qkv_dw_probe(void* clk) { g_q2_clk = static_cast<unsigned long long*>(clk); }

// Number of per-image slots the fused Gram pass needs for these shapes on this device (>= 1): the workgroups of the
// row-streaming kernel take contiguous unit ranges, an image is touched by ceil(units per image / units per workgroup) + 1 of them.
static int q2_grid(long long units, int cus) { return (int)(units < cus ? units : cus); }
extern "C" int cdfo_qkv_dw_gram_slots(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return CDFO_EINVAL;
  const int cus = qd_cu_count();
  if (cus <= 0) return CDFO_EINVAL;
  const q2_geom G = q2_geometry(H, W);
  const long long units = (long long)B * G.upi;
  const long long grid = q2_grid(units, cus), per = (units + grid - 1) / grid;
  return (int)((G.upi + per - 1) / per + 1);
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
  a.clk = nullptr;
  a.x = x; a.ldx = ldx; a.B = B; a.H = H; a.W = W;
  a.w = static_cast<const unsigned short*>(w_bf16); a.bias = bias; a.dw = dw_w; a.eps = eps; a.out = out; a.ldo = ldo; a.gram = gram; a.nslot = gram_slots;
  const double px = (double)B * H * W;
  CdfoProfScope prof(st, KID_DWCONV, 2.0 * px * 192 * (64 + 9) + (gram ? 2.0 * px * 640 : 0.0), 4.0 * px * (64 + (gram ? 64 : 192)));
  if (gram) {       // the row-streaming kernel (qkv_dw2); the tiled one keeps the 192-channel form
    if (gram_slots < cdfo_qkv_dw_gram_slots(B, H, W)) return CDFO_EINVAL;
    if ((long long)H * W * ldo * 4 >= (1ll << 31)) return CDFO_EINVAL;      // per-image buffer descriptor of the v stores
    const q2_geom G = q2_geometry(H, W);
    a.clk = g_q2_clk;
    if (g_q2_clk) {
      static CdfoAttrOnce once3;
      const hipError_t e3 = cdfo_set_max_lds(once3, reinterpret_cast<const void*>(qkv_dw2_kernel<true>), Q2_LDS);
      if (e3 != hipSuccess) return (int)e3;
      hipLaunchKernelGGL(qkv_dw2_kernel<true>, dim3(q2_grid((long long)B * G.upi, cus)), dim3(Q2_THREADS), Q2_LDS, st, a);
      CDFO_LAUNCH_CHECK();
      return 0;
    }
    static CdfoAttrOnce once2;
    const hipError_t e2 = cdfo_set_max_lds(once2, reinterpret_cast<const void*>(qkv_dw2_kernel<false>), Q2_LDS);
    if (e2 != hipSuccess) return (int)e2;
    hipLaunchKernelGGL(qkv_dw2_kernel<false>, dim3(q2_grid((long long)B * G.upi, cus)), dim3(Q2_THREADS), Q2_LDS, st, a);
    CDFO_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(qkv_dw_kernel, dim3(grid), dim3(QD_THREADS), QD_LDS, st, a);
  CDFO_LAUNCH_CHECK();
  return 0;
}
