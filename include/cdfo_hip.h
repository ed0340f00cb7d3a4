/* cdfo_hip.h -- C-ABI of libcdfo_hip.so: the MI355X (gfx950) kernels behind the CDFO CVSR_V8 forward path
 * and the ops/dcn deformable-convolution operator.
 *
 * Plain pointers (device memory), ints and a hipStream_t (passed as void*); no torch types.  Every entry
 * point returns 0 on success, a negative CDFO_E* code for a rejected argument, or the positive hipError_t of
 * a failed launch.  All activation tensors are fp32, "pixel-major" (N,H,W,C) with an explicit channel pitch
 * `ld` (floats between consecutive pixels), so a tensor may be a channel slice of a wider buffer.
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   - every ATen call inside arch/SIDECVSR_our.py:4406-4481 (CVSR_V8.forward) and the modules it reaches;
 *   - ops/dcn/src/deform_conv_cuda.cpp:151-156 (deform_conv_forward_cuda) and :486-492
 *     (modulated_deform_conv_cuda_forward), i.e. the pybind module `deform_conv_cuda`
 *     (ops/dcn/src/deform_conv_cuda.cpp:681-695).
 */
#ifndef CDFO_HIP_H
#define CDFO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CDFO_EINVAL (-1)   /* bad shape / unsupported configuration */
#define CDFO_EALIGN (-2)   /* pointer or pitch not 16-byte aligned  */
#define CDFO_MAXSRC 8

enum { CDFO_ACT_NONE = 0, CDFO_ACT_LRELU = 1, CDFO_ACT_RELU = 2, CDFO_ACT_SIGMOID = 3 };
enum { CDFO_STORE_PLAIN = 0, CDFO_STORE_SHUFFLE2 = 1, CDFO_STORE_S2D = 2, CDFO_STORE_TAPS9 = 3, CDFO_STORE_OFFMASK = 4,
       CDFO_STORE_S2D_HS = 5 /* cdfo_conv3x3_c64_wino[_up2]: CDFO_STORE_S2D with half-split rows, see cdfo_conv_args.src_halfsplit */ };
enum { CDFO_DTYPE_F32 = 0, CDFO_DTYPE_F16 = 1, CDFO_DTYPE_F64 = 2 };   /* element type tag of the *_dt entry points */
enum { CDFO_PREC_F32 = 0, CDFO_PREC_BF16X3 = 1, CDFO_PREC_BF16 = 2, CDFO_PREC_FP16X2 = 3, CDFO_PREC_FP16 = 4, CDFO_PREC_FP16X1 = 5 };

/* ABI version / build info.  */
int cdfo_abi_version(void);
const char* cdfo_build_info(void);
/* The persistent kernels size their grids to the device's CU count; a host thread that runs two schedules beside each other on two
 * streams gives each a share: launches of the CALLING THREAD fill at most n CUs from now on (0 = all).  Returns the previous value.  */
int cdfo_set_cu_limit(int n);

/* Dense convolution as an implicit GEMM on the matrix cores (replaces F.conv2d for 1x1 / 3x3, stride 1|2;
 * arch/SIDECVSR_our.py e.g. :383-387 Block_.body, :4382, :4386, :4390-4391).
 * Input = channel-concatenation of up to CDFO_MAXSRC sources (replaces torch.cat(...,1) in front of a conv).
 * w: packed by cdfo_pack_conv_weight(); optional per-image weights (w_bstride != 0).
 * Epilogue: +bias -> act -> +res1 -> +res2 -> store (plain or 2x pixel-shuffle, arch.py:4473-4474).
 * CDFO_STORE_S2D writes output pixel (y,x), channel n to pixel (y/2,x/2), channel ((y&1)*2+(x&1))*Cout+n of a
 * [B,Ho/2,Wo/2,4*Cout] tensor (space-to-depth), the input format of the stride-2-composed conv of Block_'s 2x branch.  */
typedef struct {
  const float* src[CDFO_MAXSRC]; int ld[CDFO_MAXSRC]; int cs[CDFO_MAXSRC]; int nsrc;
  int B, H, W, Ho, Wo;
  int ks, stride, pad;
  int Cin, Cout, CoutP;
  const float* w; long long w_bstride; const float* bias;
  int act;
  const float* res1; int ldr1;
  const float* res2; int ldr2;
  float* out; int ldo; int store_mode;
  int prec;
  const float* ln_gamma; const float* ln_beta;   /* optional fused per-pixel LayerNorm of a single 64-channel source (1x1 only) */
  const unsigned* tap_mask;   /* optional, cdfo_conv3x3_bf16 only: per 16-channel chunk, bit t set = tap t has non-zero weights */
  int src_f16;                /* cdfo_conv3x3_bf16 only: the single source is an fp16 tensor (ld in halves); implies CDFO_PREC_FP16 */
  int out_f16;                /* cdfo_conv3x3_bf16 only: store the result as fp16 (ldo in halves); no residual inputs */
  void* out2_cp16;            /* optional, cdfo_conv3x3_ring only: second copy of the result as an fp16 chunk-planar tensor
                                 [B][Cout/16][H][W][16] (the next Block_'s body[0] source), Cout % 16 == 0 */
  int src_plane_wrap;         /* optional, cdfo_conv3x3_ring only: chunk c reads source plane c % src_plane_wrap (0 = plane c), so
                                 that a K-expanded product (a_hi | a_lo | a_hi) x (w_hi | w_hi | w_lo) needs no duplicated planes */
  const float* res_up2; int ldru;   /* optional, cdfo_conv3x3_ring only: a HALF-resolution residual [B][H/2][W/2][ldru] that is
                                 added after bilinear x2 up-sampling (align_corners=False), i.e. Block_'s x1/2 branch */
  int out2_lo;                /* cdfo_conv3x3_ring with out2_cp16: the copy is hi | lo planes [B][2*Cout/16][H][W][16] (fp16(v), then
                                 fp16(v - fp16(v))): the source of a split-fp16 convolution on the same kernel */
  /* CDFO_STORE_OFFMASK (cdfo_conv3x3_bf16 only, Cout = 3 * third, Wo % 4 == 0, no activation / residuals): the convolution is
   * MVDualAttAlignment's conv_offset[2] and its epilogue is the module's offset / mask assembly (arch/SIDECVSR_our.py:3336-3350),
   * written straight into the DCN operator's NCHW inputs -- out = offset [B][2*third][Ho][Wo], mask_out = mask [B][third][Ho][Wo]:
   *   off_accumulate == 0 (the first of the two heads):  offset[k] = off_mag * tanh(v[k]) + flow[b][1 - (k & 1)],  mask[k] = v[2*third + k]
   *   off_accumulate != 0 (the second head, in place):    offset[k] += off_mag * tanh(v[k]),  mask[k] = sigmoid(mask[k] + v[2*third + k])
   * flow: the motion field [B][2][Ho][Wo], image pitch flow_bstride floats (read by the first head only).  */
  float* mask_out; const float* flow; long long flow_bstride; float off_mag; int off_accumulate;
  const float* res2_pixscale;  /* optional, cdfo_conv1x1_bf16x3 only (its streaming form): res2 enters the sum as res2[p][c] * res2_pixscale[p],
                                 one factor per pixel [B][H*W] -- a spatial gate applied to the residual without writing the gated tensor */
  int src_halfsplit;           /* cdfo_conv3x3_ring only: the fp16 chunk-planar source's rows are stored as [8-channel half][W][8]
                                 (a row's first halves, then its second halves: what cdfo_conv3x3_c64_wino writes with
                                 CDFO_STORE_S2D_HS, 16 contiguous bytes per lane and pixel) instead of [W][16] */
} cdfo_conv_args;
int cdfo_sizeof_conv_args(void);   /* sizeof(cdfo_conv_args) as the library was built: a binding checks its own mirror against it */
int cdfo_conv_igemm(const cdfo_conv_args* a, void* stream);

/* Pack an OIHW fp32 weight [Cout][Cin][ks][ks] (device) into the layout cdfo_conv_igemm reads:
 * [Cin/16][ks*ks][4][CoutP][4] floats, CoutP = Cout rounded up to 32, zero filled.
 * shuffle2 != 0 permutes output channels o = c*4+dy*2+dx -> (dy*2+dx)*(Cout/4)+c for CDFO_STORE_SHUFFLE2.
 * transposed != 0 reads an IOHW ConvTranspose2d weight and flips the taps (conv-equivalent form).  */
int cdfo_pack_conv_weight(const float* w_oihw, float* packed, int Cout, int Cin, int ks, int shuffle2,
                          int transposed, void* stream);

/* 3x3 / stride 1 / pad 1 convolution on the bf16 matrix cores, fp32 accumulate; same argument block and epilogue as
 * cdfo_conv_igemm.  a->prec = CDFO_PREC_BF16X3 (split-bf16, 3 passes, fp32-grade), CDFO_PREC_BF16 (plain bf16) -- both
 * with a->w packed by cdfo_pack_conv3x3_bf16 ([hi|lo] x [Cin/16][9][2][CoutP][8] bf16, CoutP = Cout up to 64) -- or
 * CDFO_PREC_FP16X2 (fp16 hi+lo activations x fp16 weights, 2 passes; a->w packed by cdfo_pack_conv3x3_f16, one block),
 * or CDFO_PREC_FP16 with a->src_f16 (the source already IS an fp16 tensor -- Block_'s 256-channel body intermediate,
 * whose rounding to fp16 does not move the forward's error -- one pass, staging is a plain 16-byte copy), or
 * CDFO_PREC_FP16X1 (fp32 source rounded once to fp16 while staging, fp16 weights, one pass: used for the convolutions
 * INSIDE Block_, where the oracle emulation shows the forward's error is set by the weight rounding alone).  */
int cdfo_conv3x3_bf16(const cdfo_conv_args* a, void* stream);
int cdfo_pack_conv3x3_bf16(const float* w_oihw, void* packed, int Cout, int Cin, void* stream);
int cdfo_pack_conv3x3_f16(const float* w_oihw, void* packed, int Cout, int Cin, void* stream);

/* Block_.body[0] (arch/SIDECVSR_our.py:383-387, Conv2d(64, Cout, 3, 1, 1) + activation) as a persistent,
 * weights-stationary fp16 MFMA kernel: one workgroup per CU keeps the fp16 weights of 64 output channels in LDS and
 * streams pixel tiles past them.  src_cp16: fp16 "chunk-planar" tensor [B][4][H][W][16] (cdfo_to_cp16, or
 * cdfo_resample2 with out_f16 = 2); w_f16/CoutP: cdfo_pack_conv3x3_f16 packing and its padded channel count;
 * out_cp16: fp16 chunk-planar [B][Cout/16][H][W][16] (CDFO_STORE_PLAIN) or its space-to-depth form
 * [B][4*Cout/16][H/2][W/2][16], chunk = ((y&1)*2+(x&1))*Cout/16 + channel/16 (CDFO_STORE_S2D).
 * H even, Cout % 64 == 0, the source smaller than 2 GiB.  Since round 3 the call runs the ring-fed, wave-specialised form
 * (four producer waves feed two groups of four consumer waves, v_mfma_f32_16x16x32_f16); same operands, same result layout.
 * dbg: 0 (developer ablation flags otherwise: 1 / 2 / 8 skip the MFMAs / the DMA / the epilogue, 4 = MFMA-shape clock
 * experiment with WRONG arithmetic; with dbg 32 clk_probe receives s_memtime stamps of the consumer waves, 256 x 12 x 4 x 8
 * 64-bit words; with dbg 128 -- private-halo form only -- per wave of the grid {shader-clock cycles, start, end in 100 MHz
 * real-time ticks}: 3 x 8 x 256 words; else pass NULL).  */
int cdfo_conv3x3_c64_ws(const void* src_cp16, int B, int H, int W, const void* w_f16, int CoutP, const float* bias,
                        int Cout, int act, void* out_cp16, int store_mode, int dbg, void* clk_probe, void* stream);
/* The same convolution (Block_.body[0], arch/SIDECVSR_our.py:383-387: 3x3, 64 -> Cout, stride 1, pad 1, + bias + activation; fp16
 * chunk-planar source and result exactly as cdfo_conv3x3_c64_ws) as a row-streaming Winograd F(2,3) product along x: 2/3 of the direct
 * form's MFMAs, weights stationary in REGISTERS (csrc/conv3x3_wino.hip).  w_wino: cdfo_pack_conv3x3_wino image (Cout x 64 x 12 fp16:
 * [Cout/16][dy 3][xi 4][K half 2][lane 64][8], element e of lane (kg, i) = U_xi[dy][channel 32 half + 8 kg + e][output channel 16 cb + i]
 * with U0 = g0, U1 = (g0 + g1 + g2)/2, U2 = (g0 - g1 + g2)/2, U3 = g2 over the kernel row g = w[:, :, dy, 0..2], formed in fp32).
 * Cout % 128 == 0, W even (H even for CDFO_STORE_S2D), one image of source / result smaller than 2 GiB; any B.  The transformed
 * inputs are up to 2x the source's magnitude in fp16: |src| must stay below 32752.  */
int cdfo_conv3x3_c64_wino(const void* src_cp16, int B, int H, int W, const void* w_wino, const float* bias, int Cout, int act,
                          void* out_cp16, int store_mode, void* stream);
int cdfo_pack_conv3x3_wino(const float* w_oihw, void* packed, int Cout, void* stream);
/* the same call with developer ablation bits (dbg != 0: WRONG results; 1 no MFMAs, 2 no global loads, 4 no stores, 8 no epilogue
 * arithmetic, 16 no barrier; 512 = timeline probe: clk_probe receives s_memtime stamps [workgroup][wave][8] of one steady-state
 * batch, 64-bit words, else NULL): tools/bench_wino.py, tools/wino_timeline.py */
/* Block_'s double-resolution branch (arch.py:398-404: body(up(x))) without its double-resolution source: src_lr_cp16
 * [B][4][H/2][W/2][16] = up.0(x) at the block's resolution (cdfo_block_prologue2's t16); H x W (multiples of 4) = the size of the x2
 * image the convolution runs on.  The bilinear x2 (align_corners = False, clamped taps; the convolution pads the x2 image with zeros)
 * is folded into the F(2,3) input transform.  Result as cdfo_conv3x3_c64_wino(..., store_mode) with store_mode = CDFO_STORE_S2D
 * ([B][4 Cout/16][H/2][W/2][16]) or CDFO_STORE_S2D_HS (the same planes with half-split rows [H/2][2][W/2][8]).  */
int cdfo_conv3x3_c64_wino_up2(const void* src_lr_cp16, int B, int H, int W, const void* w_wino, const float* bias, int Cout, int act,
                              void* out_cp16, int store_mode, void* stream);
int cdfo_conv3x3_c64_wino_dbg(const void* src_cp16, int B, int H, int W, const void* w_wino, const float* bias, int Cout, int act,
                              void* out_cp16, int store_mode, int dbg, void* clk_probe, void* stream);
/* MVDualAttAlignment's conv_offset[2] (3x3, 64 -> Cout = 27 dg, arch/SIDECVSR_our.py:3285-3289) on the weights-stationary kernel with
 * the module's offset / mask assembly (arch.py:3336-3350) as its epilogue -- the CDFO_STORE_OFFMASK contract of cdfo_conv_args above,
 * single-pass fp16 operands (src fp16 chunk-planar [B][4][H][W][16], weights from cdfo_pack_conv3x3_f16 with CoutP padded channels):
 * offset [B][2 Cout/3][H][W], mask [B][Cout/3][H][W] fp32 NCHW, flow [B][2][H][W] (image pitch flow_bstride floats).
 * accumulate == 0: first head; != 0: second head, in place.  H even; any W.  */
int cdfo_conv3x3_c64_ws_offmask(const void* src_cp16, int B, int H, int W, const void* w_f16, int CoutP, const float* bias, int Cout,
                                float* offset, float* mask, const float* flow, long long flow_bstride, float mag, int accumulate,
                                void* stream);

/* Residual form of cdfo_conv3x3_c64_ws (ResidualBlock_noBN's second convolution, arch.py:261-262):
 * out[B][H][W][ldo] (fp32, pixel-major) = act(conv + bias) + res1 (+ res2), res* fp32 pixel-major; optionally also the
 * fp16 chunk-planar copy out2_cp16 [B][Cout/16][H][W][16] (NULL to skip).  */
int cdfo_conv3x3_c64_ws_res(const void* src_cp16, int B, int H, int W, const void* w_f16, int CoutP, const float* bias,
                            int Cout, int act, float* out, int ldo, const float* res1, int ldr1, const float* res2,
                            int ldr2, void* out2_cp16, void* stream);
/* 3x3 / stride 1 / pad 1 convolution of an fp16 chunk-planar source (a->src[0] = [B][Cin/16][H][W][16], a->src_f16 = 1,
 * a->ld[0] = 16, a->cs[0] = Cin) as a persistent kernel fed by an LDS-DMA ring: Block_.body[2] (256 -> 64) and the
 * composed stride-2 convolution of Block_'s double-resolution branch.  a->w: fp16 [Cin/16][taps][2][CoutP][8] with
 * taps = 9 (cdfo_pack_conv3x3_f16) or, when a->tap_mask is given, taps = 4: only the chunk's four active taps, in
 * ascending tap order.  Epilogue and the other fields as cdfo_conv_igemm (fp32 or fp16 pixel-major result).  */
int cdfo_conv3x3_ring(const cdfo_conv_args* a, void* stream);
/* fp32 pixel-major [B][P][ldi] (C channels, C % 16 == 0) -> fp16 chunk-planar [B][C/16][P][16].  */
int cdfo_to_cp16(const float* in, int ldi, int B, long long P, int C, void* out_cp16, void* stream);

/* 1x1 convolution as an HBM-streaming GEMM in split-bf16 (3-pass, fp32-grade) arithmetic; same argument block and
 * epilogue as cdfo_conv_igemm with ks = 1 (fp32 packing of cdfo_pack_conv_weight, per-image weights, fused LayerNorm,
 * residuals, pixel-shuffle store).  Every source must be a multiple of 64 channels, CoutP a multiple of 64 (<= 256).
 * With a->out2_cp16 != NULL (Cout == 64, plain store): ln_gamma / ln_beta describe a LayerNorm64 of the RESULT, written there as
 * fp16 hi | lo chunk-planar planes [B][8][P][16] (what cdfo_layernorm64_cp16hl would produce from `out`); with out2_cp16 set and
 * ln_gamma == ln_beta == NULL (round 4) the second output is the result itself as fp16 chunk-planar [B][4][P][16] (= cdfo_to_cp16(out)).  */
int cdfo_conv1x1_bf16x3(const cdfo_conv_args* a, void* stream);

/* Upsampler tail without the HR feature map (arch.py:4474-4480).  cdfo_conv1x1_bf16x3 with a->store_mode =
 * CDFO_STORE_TAPS9 (upconv2: Cout = 256, pixel-shuffle weight packing, LeakyReLU; a->res2 = conv_last.weight
 * [1][64][3][3]) stores for every HR pixel the nine per-tap channel sums of conv_last into out[B][2H][2W][ldo >= 9];
 * cdfo_conv_last_taps then forms out[b][y][x] = bias + sum_k taps[(y,x) + delta_k][k] + bilinear_x4(x_center).  */
int cdfo_conv_last_taps(const float* taps, int ldt, const float* bias, const float* xc, long long xc_bstride, int B, int Hh,
                        int Wh, float* out, void* stream);
/* Layout changes at the module boundary (reference tensors are NCHW).  */
int cdfo_nchw_to_nhwc(const float* in, float* out, int B, int C, int H, int W, int ldo, void* stream);
int cdfo_nhwc_to_nchw(const float* in, int ldi, float* out, int B, int C, int H, int W, void* stream);
/* out[n][b] = in[b][n] for blocks of `block` floats (clip-major <-> frame-major feature stacks).  */
int cdfo_swap_outer(const float* in, float* out, int B, int N, long long block, void* stream);
/* Range probe of a dense fp32 buffer (n % 4 == 0): slots2[0] = max |x| over the finite elements (as the bit pattern of a
 * float, combined by atomicMax), slots2[1] |= 1 if any element is NaN / infinite.  Caller zeroes the two words. */
int cdfo_range_probe(const float* x, long long n, void* slots2, void* stream);

/* ---- bandwidth-bound pixel-major kernels (pointwise.hip) ------------------------------------------------- */
/* 3x3 conv 1 -> 64 channels on a single-channel plane [B][H][W] (image pitch img_bstride floats); raw OIHW weight
 * [64][1][3][3]; optional second output out2 = result + add.  arch/SIDECVSR_our.py:4379-4384,4417-4418,4446-4449. */
int cdfo_stem_conv(const float* img, long long img_bstride, const float* w, const float* bias, int B, int H, int W,
                   int act, float* out, int ldo, const float* add, int lda, float* out2, int ldo2, void* stream);
/* two 1 -> 64 3x3 convolutions of the same plane in one pass: outA = conv(img; wA, bA) + add, outB = actB(conv(img; wB, bB))
 * (fea_com = fea_i + conv_expand_rms(rms), arch.py:4446-4449, and relu(conv_du_re.0(conv_expand_rms(rms))), arch.py:2148-2152,
 * 2200, with the 1x1 conv_du_re.0 composed into wB / bB by the caller: `rms_prior` is never written).  s2dB != 0: outB in
 * space-to-depth form [B][H/2 + 1][W/2 + 1][4 * 64] (phase-major channels; last row / column = the caller's zeros), over which
 * the stride-2 conv_du_re.2 is a stride-1 convolution with a 2x2 tap window. */
int cdfo_stem_conv2(const float* img, long long img_bstride, const float* wA, const float* bA, const float* add, int lda,
                    float* outA, int ldoA, const float* wB, const float* bB, int actB, float* outB, int ldoB, int s2dB, int B,
                    int H, int W, void* stream);
/* per-pixel LayerNorm over 64 channels (arch.py:1169-1198). */
int cdfo_layernorm64(const float* in, int ldi, const float* gamma, const float* beta, long long npix, float* out,
                     int ldo, void* stream);
/* the same LayerNorm written as an fp16 hi / lo pair in chunk-planar layout [B][8][P][16]: planes 0-3 = fp16(v) of channels
 * 0-63, planes 4-7 = fp16(v - fp16(v)): the source of a split-fp16 (fp32-grade) 3x3 convolution on cdfo_conv3x3_ring. */
int cdfo_layernorm64_cp16hl(const float* in, int ldi, const float* gamma, const float* beta, int B, long long P, void* out,
                            void* stream);
/* depthwise 3x3, pad 1, no bias; raw weight [C][1][3][3] (arch.py:1552). */
int cdfo_dwconv3x3(const float* in, int ldi, const float* w, int B, int H, int W, int C, float* out, int ldo,
                   void* stream);
/* flow_warp (arch.py:3068-3099); mv = [B][2][H][W] planes (x then y), image pitch mv_bstride floats. */
int cdfo_flow_warp(const float* in, int ldi, const float* mv, long long mv_bstride, int B, int H, int W, int C,
                   float* out, int ldo, void* stream);
/* bilinear x2 (up=1) or x0.5 (up=0), align_corners=False (arch.py:324-333); accumulate: out += result;
 * out_f16 (up only): 1 = `out` is an fp16 pixel-major tensor (ldo in halves) that feeds a single-pass fp16 convolution;
 * 2 = `out` is an fp16 chunk-planar tensor [B][C/16][2H][2W][16] (ldo ignored), the source of cdfo_conv3x3_c64_ws. */
int cdfo_resample2(const float* in, int ldi, int B, int H, int W, int C, float* out, int ldo, int up, int accumulate,
                   int out_f16, void* stream);
/* out = in * gate[b][c] (CALayer, arch.py:2041-2043); out_cp16 (optional, C % 16 == 0): the same values as the fp16
 * chunk-planar tensor [B][C/16][P][16] that cdfo_conv3x3_c64_ws reads. */
int cdfo_scale_channels(const float* in, int ldi, const float* gate, int B, long long P, int C, float* out, int ldo,
                        void* out_cp16, void* stream);
/* conv_last 3x3 64->1 (+bias) + bilinear x4 of the centre LR frame (arch.py:4476-4480); out = [B][Hh][Wh]. */
int cdfo_conv_last(const float* in, int ldi, const float* w, const float* bias, const float* xc, long long xc_bstride,
                   int B, int Hh, int Wh, float* out, void* stream);

/* ---- thin 16-channel layers of the prior U-net (smallconv.hip; arch.py:1815-1834, 2719-2730) --------------- */
int cdfo_small_conv16(const float* in, int ldi, const float* w, const float* bias, int B, int H, int W, int stride,
                      int pad, int out_pad, int transposed, int act, float* out, int ldo, void* stream);
/* the same, result as fp16 hi | lo planes [B][2][Ho*Wo][16] (chunk-planar): the split-fp16 source of cdfo_conv3x3_ring */
int cdfo_small_conv16_hl(const float* in, int ldi, const float* w, const float* bias, int B, int H, int W, int stride,
                         int pad, int out_pad, int transposed, int act, void* out_hl, void* stream);
/* lrelu(body.0(conv_second(img))): the prior U-net's first layer in the feature extractor's first round (arch.py:4420,
 * 1463-1468, 1819-1820) straight from the one-channel prior image -- conv_second's 64-channel result feeds only this
 * layer there and has no activation, so the two 3x3 convolutions are composed (exactly, borders included: a tap of body.0
 * outside the image sees zero, not conv_second's bias).  wc[t][u][o] = sum_c W0[o][c][t] W2[c][u] ([9][9][16]),
 * bt[t][o] = sum_c W0[o][c][t] b2[c] ([9][16]), b0 = body.0's bias.  img element [b][y][x] at img + b*img_bstride + y*W + x. */
int cdfo_udsa_head(const float* img, long long img_bstride, const float* wc, const float* bt, const float* b0, int B, int H,
                   int W, float* out, int ldo, void* stream);
int cdfo_spatial_gate16(const float* in, int ldi, const float* w, const float* bias, int B, int H, int W, float* out,
                        int ldo, void* stream);

/* ---- reductions + per-image weight folding of the channel attentions (stats.hip) --------------------------- */
int cdfo_chan_sum_partial(const float* in, int ldi, int B, long long P, int nchunk, float* partial, void* stream);
int cdfo_gram_partial(const float* q, int ldq, const float* k, int ldk, int B, long long P, int ch_per_head, int nchunk,
                      float* partial, void* stream);
/* MDTA (arch.py:1555-1575): wout[b] = packed 1x1 weights (64->64) = project_out x blockdiag(softmax(...)). */
int cdfo_mdta_fold(const float* partial, int nchunk, const float* temperature, const float* proj_w, int B, float* wout,
                   void* stream);
/* DualAttAlignment (arch.py:3459-3491): wout[b] = packed 1x1 weights (192->64) over cat[warped, pred, x]. */
int cdfo_align_fold(const float* gram_partial, int nchunk_g, const float* sum_warp, const float* sum_pred, int nchunk_s,
                    long long P, const float* temperature, const float* du0_w, const float* du0_b, const float* du2_w,
                    const float* du2_b, const float* proj_w, const float* fusion_w, int B, float* wout, void* stream);
/* out[b] = act2(W2 act1(W1 mean_b + b1) + b2) from channel-sum partials (CALayer / conv_du_re2). */
int cdfo_vec_mlp(const float* sum_partial, int nchunk, long long P, const float* w1, const float* b1, int c1, int act1,
                 const float* w2, const float* b2, int c2, int act2, int B, float* out, void* stream);

/* ---- LLongRangAttention (attention.hip; arch.py:2179-2249) -------------------------------------------------- */
/* noise: the uniform draws of arch.py:2169 as [B][64][H*W]. */
int cdfo_rdab_prep(const float* xq, int ldx, const float* vmax, const float* noise, const float* wW, const float* bW,
                   int B, long long P, float* sq, int lds_, float* vrow, int ldv, float* qwin, int ldw, void* stream);
/* The same with the uniform draws generated inside the kernel (Philox4x32-10 keyed by seed, `draw` = index of the call
 * within the forward): the reference's default behaviour, torch.rand_like at arch.py:2169, without the 256 B/pixel noise
 * tensor.  noise_out: optional [B][64][H*W] copy of the drawn values (parity tests replay them through the oracle). */
int cdfo_rdab_prep_rng(const float* xq, int ldx, const float* vmax, long long seed, int draw, float* noise_out,
                       const float* wW, const float* bW, int B, long long P, float* sq, int lds_, float* vrow, int ldv,
                       float* qwin, int ldw, void* stream);
/* cdfo_rdab_prep_rng with the Philox key read from DEVICE memory (seed_dev: one 64-bit word, 8-byte aligned): a captured
 * HIP graph then draws fresh uniforms on every replay -- the caller rewrites the word between replays -- instead of
 * freezing a launch argument (the reference draws per forward, arch.py:2169). */
int cdfo_rdab_prep_rng_dev(const float* xq, int ldx, const float* vmax, const void* seed_dev, int draw, float* noise_out,
                           const float* wW, const float* bW, int B, long long P, float* sq, int lds_, float* vrow, int ldv,
                           float* qwin, int ldw, void* stream);
int cdfo_colconv9(const float* in, int ldi, const float* wH, const float* bH, int B, int H, int W, float* out, int ldo,
                  void* stream);
/* Evaluation metrics on the device (metric/psnr_ssim.py:278-317 calculate_psnr, :320-399 _ssim / calculate_ssim): fp64
 * partial sums over [N][H][W] fp32 planes, values = plane * scale (optionally clamped to [0,255] / rounded to integers),
 * `crop` border pixels dropped.  metric 0: sum (a-b)^2;  metric 1: sum of the SSIM map (11x11 Gaussian, sigma 1.5,
 * valid positions).  partial receives [N][*nblocks_out] doubles (partial_cap = its capacity in doubles); the caller
 * adds them up in order and divides by the number of positions.  */
int cdfo_metric_partials(const float* a, const float* b, int N, int H, int W, int crop, float scale, int clamp, int round8,
                         int metric, double* partial, int partial_cap, int* nblocks_out, void* stream);
/* Block_ prologue (arch.py:378-406): from one read of the block input x [B][H][W][64] (H, W even) the fp16 chunk-planar
 * sources of its two resampled branches: u16 [B][4][2H][2W][16] = bilinear_x2(up.0(x)) and d16 [B][4][H/2][W/2][16] =
 * down.0(mean2x2(x)).  w_bf16: split-bf16 1x1 weights [hi|lo][4][2][128][8], rows 0-63 = up.0, 64-127 = down.0,
 * element (s,h,n,j) = W[n][16s+8h+j]; bias128 = [up.0 bias | down.0 bias].  x16 (optional): also the fp16 chunk-planar
 * copy [B][4][H][W][16] of x itself, the source of the block's own-resolution branch.  */
int cdfo_block_prologue(const float* x, int ldx, int B, int H, int W, const void* w_bf16, const float* bias128, void* u16,
                        void* d16, void* x16, void* stream);
/* The same with the double-resolution source left to its consumer: exactly one of u16 / t16 is given; t16 [B][4][H][W][16] receives
 * up.0(x) itself (fp16 chunk-planar, NOT resampled), which cdfo_conv3x3_c64_wino_up2 interpolates while it builds its transformed
 * inputs -- the 4x larger u16 is then never written or read.  */
int cdfo_block_prologue2(const float* x, int ldx, int B, int H, int W, const void* w_bf16, const float* bias128, void* u16, void* t16,
                         void* d16, void* x16, void* stream);
/* MDTA front end in one pass (arch.py:1169-1198 LayerNorm, :1551-1552 qkv + qkv_dwconv): out[B][H][W][192] =
 * depthwise3x3(conv1x1(LayerNorm64(x))).  w_bf16: split-bf16 weights [hi|lo][4][2][192][8], element (s,h,n,j) =
 * W[n][16s+8h+j] * gamma[16s+8h+j]; bias[192] = W @ beta (may be NULL); dw_w: raw [192][1][3][3] taps.
 * gram != NULL fuses the attention's Gram pass (arch.py:1561-1566): q and k are not written, `out` receives v only
 * (64 channels, ldo >= 64) and gram[B][gram_slots][640] -- zero-filled by the caller, gram_slots >=
 * cdfo_qkv_dw_gram_slots(B, H, W); per slot the layout of cdfo_gram_partial with 8 channels per head: [c*10 + j] =
 * sum_p q[c] k[head(c)*8 + j], [c*10 + 8] = sum q[c]^2, [c*10 + 9] = sum k[c]^2 -- receives each workgroup's partial
 * sums in its own slot (plain stores, fixed summation order: bit-reproducible); cdfo_mdta_fold(gram, gram_slots, ...)
 * reduces the slots.  */
int cdfo_qkv_dw_gram_slots(int B, int H, int W);
int cdfo_qkv_dw(const float* x, int ldx, int B, int H, int W, const void* w_bf16, const float* bias, const float* dw_w,
                float eps, float* out, int ldo, float* gram, int gram_slots, void* stream);
/* out = softmax(Q Q^T) V per row (mode 0), column (1) or 8x8 window (2) (arch.py:2179-2249), flash style on the matrix
 * cores: both products with fp16 hi + fp16 lo operands, three passes, fp32 accumulate (scores exact to ~1e-6 relative).
 * Modes 10 / 11 / 12: the plain-VALU forms of 0 / 1 / 2 (kept as A/B references for the tests).
 * Modes 20 / 21 / 22: modes 0 / 1 / 2 with the SECOND product (probabilities x values) on single-fp16 operands, one pass: an
 * output's error is bounded by 2^-11 * max|v| (probabilities sum to 1), ~1e-5 * |v| in practice; the scores keep three passes.
 * The forward's default fp16x2 arithmetic uses these (cdfo_amd/cvsr_v8.py); the fp32-grade modes and the training path do not. */
int cdfo_seq_attn(const float* q, int ldq, const float* v, int ldv, float* out, int ldo, int B, int H, int W, int mode,
                  void* stream);

/* ---- deformable convolution forward (dcn.hip) --------------------------------------------------------------
 * Replaces deform_conv_forward_cuda (ops/dcn/src/deform_conv_cuda.cpp:151-156; mask == NULL, bias == NULL) and
 * modulated_deform_conv_cuda_forward (ops/dcn/src/deform_conv_cuda.cpp:486-492) of the pybind module
 * `deform_conv_cuda` (cpp:681-695).  NCHW fp32 contiguous tensors exactly as the reference passes them:
 * in [B,C,H,W], offset [B,2*dg*kh*kw,Ho,Wo], mask [B,dg*kh*kw,Ho,Wo] or NULL, weight [Co,C/groups,kh,kw],
 * bias [Co] or NULL, out [B,Co,Ho,Wo] (written, not accumulated).  The reference's `columns` scratch tensor
 * [C*kh*kw, Ho*Wo] has no counterpart (sampling and contraction are fused); `workspace` is an optional device scratch
 * of at least B*C*H*W*4 bytes (16-byte aligned) in which the kernel keeps a group-planar copy of `in` so that the
 * bilinear corners of a deformable group's channels are single 16-byte gathers; NULL selects the direct NCHW gathers
 * (same results, slower).  With cdfo_dcn_workspace_bytes(...) bytes (non-zero for groups == 1, (C/dg) % 4 == 0,
 * Co % 32 == 0, Co <= 128, kh*kw <= 64 -- the alignment module's shape) the fast kernel of dcn_fast.hip runs: whole-K
 * LDS staging and split-fp16 matrix products (fp32-grade, ~2^-22 relative; |in * mask| < 65504).  */
long long cdfo_dcn_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int groups, int deformable_groups);
int cdfo_dcn_forward(const float* in, const float* offset, const float* mask, const float* weight, const float* bias,
                     float* out, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw,
                     int dh, int dw, int groups, int deformable_groups, void* workspace, long long workspace_bytes,
                     void* stream);

/* ---- deformable convolution backward (dcn_bwd.hip) -----------------------------------------------------------
 * One entry point behind the module's three backward functions: deform_conv_backward_input_cuda
 * (ops/dcn/src/deform_conv_cuda.cpp:260-266: grad_in + grad_offset; mask, grad_mask, grad_weight, grad_bias NULL),
 * deform_conv_backward_parameters_cuda (cpp:373-378: grad_weight only, `scale` multiplies the contribution) and
 * modulated_deform_conv_cuda_backward (cpp:566-573: all five).  Tensors as in cdfo_dcn_forward plus
 * grad_out [B,Co,Ho,Wo].  Any gradient pointer may be NULL (skipped).  grad_in, grad_weight and grad_bias are
 * ACCUMULATED into -- the reference's callers hand over zero-filled tensors (ops/dcn/deform_conv.py:71-72, 85,
 * 154-158) -- grad_offset and grad_mask are assigned.  kh*kw <= 64.  grad_in / grad_weight use fp32 hardware
 * atomics, so their summation order (not their value beyond fp32 rounding) varies between runs, as in the reference. */
int cdfo_dcn_backward(const float* in, const float* offset, const float* mask, const float* weight,
                      const float* grad_out, float* grad_in, float* grad_offset, float* grad_mask, float* grad_weight,
                      float* grad_bias, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph,
                      int pw, int dh, int dw, int groups, int deformable_groups, float scale, void* stream);
/* The same with a workspace of cdfo_dcn_backward_workspace_bytes(...) bytes (16-byte aligned; non-zero for groups == 1,
 * C/dg == 4, kh*kw <= 9, Co <= 64 -- the alignment module's shape; -1 on bad shapes): the weight gradient is then
 * contracted on the matrix cores inside the data-gradient kernel (grad_output x the column values it samples anyway)
 * instead of by a second sampling pass.  NULL / too small a workspace = cdfo_dcn_backward.  */
long long cdfo_dcn_backward_workspace_bytes(int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph,
                                            int pw, int dh, int dw, int groups, int deformable_groups);
int cdfo_dcn_backward_ws(const float* in, const float* offset, const float* mask, const float* weight,
                         const float* grad_out, float* grad_in, float* grad_offset, float* grad_mask, float* grad_weight,
                         float* grad_bias, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph,
                         int pw, int dh, int dw, int groups, int deformable_groups, float scale, void* workspace,
                         long long workspace_bytes, void* stream);

/* ---- the operator's other dtypes (dcn_typed.hip) --------------------------------------------------------------
 * The reference instantiates its kernels for float, double and half (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * ops/dcn/src/deform_conv_cuda_kernel.cu:258,352,450,780,812,845).  Same tensors and conventions as cdfo_dcn_forward /
 * cdfo_dcn_backward with elements of `dtype` (CDFO_DTYPE_*): F32 forwards to those; F16 = fp16 tensors, fp32 arithmetic
 * (operands widened into `workspace`, the fp32 kernels run, results narrowed once; accumulating gradients are added to the
 * fp16 tensors); F64 = fp64 VALU kernels, fp64 atomics (correctness-first).  F32 / F16 backward: the workspace also carries
 * what cdfo_dcn_backward_ws wants.  `workspace` must hold
 * cdfo_dcn_workspace_bytes_dt(...) bytes, 16-byte aligned (`backward` = 0 / 1; F64 needs none; returns -1 on bad shapes). */
long long cdfo_dcn_workspace_bytes_dt(int dtype, int backward, int B, int C, int H, int W, int Co, int kh, int kw, int sh,
                                      int sw, int ph, int pw, int dh, int dw, int groups, int deformable_groups);
int cdfo_dcn_forward_dt(int dtype, const void* in, const void* offset, const void* mask, const void* weight,
                        const void* bias, void* out, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                        int ph, int pw, int dh, int dw, int groups, int deformable_groups, void* workspace,
                        long long workspace_bytes, void* stream);
int cdfo_dcn_backward_dt(int dtype, const void* in, const void* offset, const void* mask, const void* weight,
                         const void* grad_out, void* grad_in, void* grad_offset, void* grad_mask, void* grad_weight,
                         void* grad_bias, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw,
                         int dh, int dw, int groups, int deformable_groups, float scale, void* workspace,
                         long long workspace_bytes, void* stream);

/* ---- backward kernels of the CVSR_V8 training path (train_ops.hip) ------------------------
 * cdfo_amd/autograd.py runs the module under torch autograd (train_LD_37.py:376-381): every Function's forward is one of
 * the exact-fp32 forward entry points above, its backward either the same entry points with adjoint operands (flipped /
 * transposed weights) or one of these.  fp32 pixel-major tensors with pitch ld (floats).
 * cdfo_conv_wgrad:   out[a][b][ky][kx] = sum_{n,y,x} S[n,y,x,a] * L[n, y*stride+ky-pad, x*stride+kx-pad, b]  -- a
 *                    convolution's weight gradient with S = grad_out, L = input (OIHW), a transposed convolution's with
 *                    S = input, L = grad_out (IOHW).  Split-K exact-fp32 MFMA into `slab`
 *                    (cdfo_conv_wgrad_slab_floats(...) floats), slices summed in a fixed order into
 *                    dw[(a*Btot + b_off + b)*ks*ks + tap] (b_off / Btot: channel offset of this source in a concatenated input).
 * cdfo_coldot:       out[img][c] = scale * sum_p a[img][p][c] * (b ? b[img][p][c] : 1)  (bias / gate / pooled gradients);
 *                    part = nimg*nchunk*C floats of scratch.
 * cdfo_ew:           mode 0 a*b, 1 a*(1-b), 2 a+b, 3 a*act'(y=b) (aux = CDFO_ACT_*), 4 out[p][c] = a[p/P][c]*scale.
 * cdfo_layernorm64_bwd, cdfo_dwconv3x3_wgrad, cdfo_spatial_gate16_bwd, cdfo_chanconv9 (9 taps along the channel axis,
 * flip = adjoint), cdfo_corr9 (weight gradient of a 9-tap convolution along the channel axis (0) or the image rows (1)),
 * cdfo_gumbel_mask (the hard mask of arch.py:2168-2195 as a tensor), cdfo_seq_attn_bwd (adjoint of cdfo_seq_attn, modes
 * 0-2, probabilities recomputed), cdfo_flow_warp_bwd (scatter with fp32 atomics into a zero-filled dx),
 * cdfo_resample2_bwd (adjoints of bilinear x2 / x0.5). */
long long cdfo_conv_wgrad_slab_floats(int A, int Bc, int ks, int nsplit);
int cdfo_conv_wgrad(const float* S, int lds_, int A, const float* L, int ldl, int Bc, int N, int Hs, int Ws, int Hl, int Wl,
                    int ks, int stride, int pad, int nsplit, float* slab, float* dw, int Btot, int b_off, void* stream);
/* the same contraction with both operands split into bf16 hi + lo on the fly (three MFMA passes, fp32 accumulate: ~16 operand
 * bits) -- the training path's default since round 3; prec = CDFO_PREC_F32 runs cdfo_conv_wgrad's exact kernel */
int cdfo_conv_wgrad_prec(const float* S, int lds_, int A, const float* L, int ldl, int Bc, int N, int Hs, int Ws, int Hl, int Wl,
                         int ks, int stride, int pad, int nsplit, float* slab, float* dw, int Btot, int b_off, int prec,
                         void* stream);
int cdfo_coldot(const float* a, int lda, const float* b, int ldb, int nimg, long long P, int C, int nchunk, float scale,
                float* part, float* out, void* stream);
int cdfo_ew(const float* a, int lda, const float* b, int ldb, long long rows, int C, int mode, int aux, float scale, long long P,
            float* out, int ldo, void* stream);
int cdfo_layernorm64_bwd(const float* x, int ldx, const float* g, int ldg, const float* gamma, long long npix, float* dx,
                         int ldo, float* part, int nblk, void* stream);
int cdfo_dwconv3x3_wgrad(const float* x, int ldx, const float* g, int ldg, int B, int H, int W, int C, float* part, int nblk,
                         void* stream);
int cdfo_spatial_gate16_bwd(const float* t, int ldt, const float* g, int ldg, const float* w, const float* bias, int B, int H,
                            int W, float* scratch, float* dt, int ldo, float* dw99, void* stream);
int cdfo_gumbel_mask(const float* vmax, const float* noise, long long seed, int draw, float* noise_out, int B, long long P,
                     float* mask, int ldm, void* stream);
int cdfo_chanconv9(const float* in, int ldi, const float* w9, const float* bias, int flip, long long npix, float* out, int ldo,
                   void* stream);
int cdfo_corr9(const float* in, int ldi, const float* g, int ldg, int axis, int B, int H, int W, float* part, int nblk,
               void* stream);
int cdfo_seq_attn_bwd(const float* q, int ldq, const float* v, int ldv, const float* o, int ldo, const float* g, int ldg,
                      float* dq, int lddq, float* dv, int lddv, int B, int H, int W, int mode, void* stream);
int cdfo_flow_warp_bwd(const float* g, int ldg, const float* mv, long long mv_bstride, int B, int H, int W, int C, float* dx,
                       int ldo, void* stream);
int cdfo_resample2_bwd(const float* g, int ldg, int B, int H, int W, int C, int up, float* din, int ldo, void* stream);

/* ---- pixel-local operators of the CVSR_V7 forward (v7_ops.hip) ---------------------------
 * fp32 pixel-major activations [B,H,W,64] with pitch ld (floats).
 * cdfo_chan_pool:     ChannelPool (arch/SIDECVSR_our.py:1883-1885): out[p] = {max_c x[p][c], mean_c x[p][c]}.
 * cdfo_spatial_gate:  SpatialAttention (arch.py:2719-2730): out = x * sigmoid(conv_ksxks(pooled) + bias), zero padding
 *                     (ks-1)/2, w = the module's [1,2,ks,ks] weight; gate_scratch: B*H*W floats (the gate map, written
 *                     by a one-thread-per-pixel pass, then applied by a pure streaming pass).
 * cdfo_rdab_mix:      RDAB.forward's mixing step (arch.py:2836-2845): out = xf * (softmax_c(vmax[b][c] - log(-log(u)))
 *                     + sigmoid(conv3x3(pooled) + b3)); vmax [B,64]; noise u in the reference's NCHW order [B,64,H,W].
 * cdfo_shrink_planes: F.interpolate(scale_factor = 0.5 (level 1) or 0.25 (level 2), bilinear, align_corners=False) of
 *                     `planes` planes per image, divided by 2 resp. 4 (arch.py:4296-4303): dst [B,planes,H>>l,W>>l]
 *                     contiguous; src element [b][pl][y][x] at src[b*src_bstride + (pl*H + y)*W + x].
 * cdfo_lincomb:       out = ca*a + cb*b + cc*c over n floats (b, c may be NULL; out may alias an input). */
int cdfo_chan_pool(const float* x, int ld, long long npix, int C, float* out, void* stream);
/* The SAME SpatialAttention applied round after round to one tensor (PartitionTransformerBlock, arch.py:1350-1368): x_k = x_0 * G_k with
 * G_k = G_{k-1} * sigmoid(conv_ksxks(G_{k-1} * pooled_0) + bias) per pixel (the gates are positive, so pooling commutes with them).
 * pooled = cdfo_chan_pool(x_0) [B][H][W][2]; cum_in = G_{k-1} [B][H][W] or NULL (= 1); cum_out = G_k (must not alias cum_in).  */
int cdfo_gate_map_cumulative(const float* pooled, const float* cum_in, const float* w, const float* bias, int B, int H, int W, int ks,
                             float* cum_out, void* stream);
int cdfo_spatial_gate(const float* x, int ld, const float* pooled, const float* w, const float* bias, int B, int H, int W,
                      int C, int ks, float* gate_scratch, float* out, int ldo, void* stream);
int cdfo_rdab_mix(const float* xf, int ld, const float* pooled, const float* w3, const float* b3, const float* vmax,
                  const float* noise, int B, int H, int W, float* out, int ldo, void* stream);
int cdfo_shrink_planes(const float* src, long long src_bstride, int planes, int B, int H, int W, int level, float* dst,
                       void* stream);
int cdfo_lincomb(float* out, const float* a, float ca, const float* b, float cb, const float* c, float cc, long long n,
                 void* stream);

/* ---- small NCHW operators of the DCN consumer modules DSTA (ops/attentionlayer.py:86-156) and MVDualAttAlignment
 * (arch/SIDECVSR_our.py:3265-3352); contiguous NCHW fp32, one thread per output (nchw_ops.hip) ------------------ */
int cdfo_conv2d_nchw(const float* in, const float* w, const float* bias, int B, int C, int H, int W, int Co, int kh,
                     int kw, int stride, int pad, int act, float* out, void* stream);
int cdfo_maxpool_nchw(const float* in, int BC, int H, int W, int k, int stride, float* out, void* stream);
int cdfo_resize_bilinear_nchw(const float* in, int BC, int H, int W, int Ho, int Wo, int accumulate, float* out,
                              void* stream);
int cdfo_avgpool_nchw(const float* in, int BC, long long P, float* out, void* stream);
/* mode 0: a+b; 1: relu(a); 2: sigmoid(a); 3: x*sigmoid(a)*y[b][c] (P = H*W). */
int cdfo_ew_nchw(const float* a, const float* b, const float* x, const float* y, long long n, long long P, int mode,
                 float* out, void* stream);
/* arch.py:3336-3350: offset = mag*tanh(o1[:2/3]) + mag*tanh(o2[:2/3]) + flow.flip(1) tiled, mask = sigmoid(o1[2/3:]+o2[2/3:]);
 * o1,o2 pixel-major [B,H,W,3*third] (pitch ld), flow [B][2][H*W] planes, outputs NCHW. */
int cdfo_mv_offset_mask(const float* o1, const float* o2, int ld, const float* flow, long long flow_bstride, int B,
                        long long P, int third, float mag, float* offset, float* mask, void* stream);

/* ---- 3x3 / stride 1 / pad 1 convolution 64 -> 16 channels (conv3x3_n16.hip; round 4): the first layer of the feature extractor's prior
 * U-net (arch/SIDECVSR_our.py:1815-1834, body.0: Conv2d(64, 16, 3, 1, 1) + LeakyReLU) as a persistent stream on
 * v_mfma_f32_16x16x32_bf16, split-bf16 three-pass arithmetic (fp32-grade, as cdfo_conv3x3_bf16's "bf16x3").  x fp32 pixel-major
 * [B,H,W,>=64] (pitch ldx), out fp32 pixel-major [B,H,W,16] (pitch ldo).  w_packed: 36,864 bytes = [18 K steps = tap*2 + 32-channel
 * half][hi | lo][64 lanes][8 bf16], lane l = output channel l & 15, input channels 32 half + 8 (l >> 4) .. + 7
 * (cdfo_amd/kernels.py::pack_conv_n16). */
int cdfo_conv3x3_c64_n16(const float* x, int ldx, int B, int H, int W, const void* w_packed, const float* bias, int act, float* out,
                         int ldo, void* stream);

/* ---- backward of the same operators (nchw_bwd.hip): DSTA and MVDualAttAlignment under autograd, like their reference classes
 * (ops/attentionlayer.py:117-156, arch/SIDECVSR_our.py:3303-3352).  Gather kernels, fixed summation order, no atomics.
 * cdfo_conv2d_nchw_bwd: any of gin [B,C,H,W] (needs w), gw [Co,C,kh,kw] (+ optional gbias [Co]; needs in) may be NULL; all ASSIGNED.
 * cdfo_maxpool_nchw_bwd: idx_scratch = BC*Ho*Wo ints (the windows' arg-max, first maximum in scan order).
 * cdfo_ew_nchw_bwd modes: 1 g*(y>0) (y = relu output), 2 g*y*(1-y) (y = sigmoid output), 4 g[bc]/P (adjoint of the plane mean),
 *   5 / 6: the DSTA gate out = x*sigmoid(a)*yv[bc] w.r.t. a / w.r.t. x; cdfo_gate_nchw_bwd_y: w.r.t. yv ([BC]).
 * cdfo_mv_offset_mask_bwd: adjoint of cdfo_mv_offset_mask w.r.t. o1, o2 (g1, g2 pixel-major, pitch ld; the motion field is a
 *   network input and receives no gradient). */
int cdfo_conv2d_nchw_bwd(const float* in, const float* w, const float* gout, int B, int C, int H, int W, int Co, int kh, int kw,
                         int stride, int pad, float* gin, float* gw, float* gbias, void* stream);
int cdfo_maxpool_nchw_bwd(const float* in, const float* gout, int BC, int H, int W, int k, int stride, int* idx_scratch, float* gin,
                          void* stream);
int cdfo_resize_bilinear_nchw_bwd(const float* gout, int BC, int H, int W, int Ho, int Wo, float* gin, void* stream);
int cdfo_ew_nchw_bwd(const float* g, const float* y, const float* a, const float* x, const float* yv, long long n, long long P,
                     int mode, float* out, void* stream);
int cdfo_gate_nchw_bwd_y(const float* g, const float* a, const float* x, int BC, long long P, float* gy, void* stream);
int cdfo_mv_offset_mask_bwd(const float* o1, const float* o2, int ld, const float* goff, const float* gmask, int B, long long P,
                            int third, float mag, float* g1, float* g2, void* stream);

/* ---- backward-side kernels of CVSR_V7's own operators (v7_train.hip; cdfo_amd/cvsr_v7_train.py): fp32 pixel-major rows of 64
 * channels.  cdfo_chan_pool_bwd: adjoint of cdfo_chan_pool (dpooled [npix][2] = d max, d mean; the first maximum in channel order
 * receives d max).  cdfo_mul_plane: out[p][c] = x[p][c] * plane[p]; cdfo_dot_plane: out[p] = sum_c a[p][c] b[p][c].
 * cdfo_gumbel_softmax: r = softmax_c(v[b][c] - log(-log u)) with u the reference's NCHW noise [B][64][P] (arch.py:2813-2822);
 * cdfo_softmax64_bwd: dz = r * (dm - sum_c r dm). */
int cdfo_chan_pool_bwd(const float* x, int ld, const float* dpooled, long long npix, float* dx, int ldo, void* stream);
int cdfo_mul_plane(const float* x, int ld, const float* plane, long long npix, float* out, int ldo, void* stream);
int cdfo_dot_plane(const float* a, int lda, const float* b, int ldb, long long npix, float* out, void* stream);
int cdfo_gumbel_softmax(const float* v, const float* u, int B, long long P, float* out, int ldo, void* stream);
int cdfo_softmax64_bwd(const float* r, int ldr, const float* dm, int ldm, long long npix, float* dz, int ldo, void* stream);

/* ---- optional per-launch HIP-event timing on the launch stream (bench.py's live roofline figures) ----------- */
int cdfo_prof_begin(int max_records);
int cdfo_prof_end(int* launches, double* ms, double* flops, double* bytes, int nkid);
int cdfo_prof_kid_count(void);

#ifdef __cplusplus
}
#endif
#endif
