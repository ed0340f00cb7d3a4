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

enum { CDFO_ACT_NONE = 0, CDFO_ACT_LRELU = 1, CDFO_ACT_RELU = 2, CDFO_ACT_SIGMOID = 3 };
enum { CDFO_STORE_PLAIN = 0, CDFO_STORE_SHUFFLE2 = 1 };
enum { CDFO_PREC_F32 = 0, CDFO_PREC_BF16X3 = 1, CDFO_PREC_BF16 = 2 };

/* ABI version / build info.  */
int cdfo_abi_version(void);
const char* cdfo_build_info(void);

/* Dense convolution as an implicit GEMM on the matrix cores (replaces F.conv2d for 1x1 / 3x3, stride 1|2;
 * arch/SIDECVSR_our.py e.g. :383-387 Block_.body, :4382, :4386, :4390-4391).
 * Input = channel-concatenation of up to three sources (replaces torch.cat(...,1) in front of a conv).
 * w: packed by cdfo_pack_conv_weight(); optional per-image weights (w_bstride != 0).
 * Epilogue: +bias -> act -> +res1 -> +res2 -> store (plain or 2x pixel-shuffle, arch.py:4473-4474).  */
typedef struct {
  const float* src[3]; int ld[3]; int cs[3]; int nsrc;
  int B, H, W, Ho, Wo;
  int ks, stride, pad;
  int Cin, Cout, CoutP;
  const float* w; long long w_bstride; const float* bias;
  int act;
  const float* res1; int ldr1;
  const float* res2; int ldr2;
  float* out; int ldo; int store_mode;
  int prec;
} cdfo_conv_args;
int cdfo_conv_igemm(const cdfo_conv_args* a, void* stream);

/* Pack an OIHW fp32 weight [Cout][Cin][ks][ks] (device) into the layout cdfo_conv_igemm reads:
 * [Cin/16][ks*ks][4][CoutP][4] floats, CoutP = Cout rounded up to 32, zero filled.
 * shuffle2 != 0 permutes output channels o = c*4+dy*2+dx -> (dy*2+dx)*(Cout/4)+c for CDFO_STORE_SHUFFLE2.
 * transposed != 0 reads an IOHW ConvTranspose2d weight and flips the taps (conv-equivalent form).  */
int cdfo_pack_conv_weight(const float* w_oihw, float* packed, int Cout, int Cin, int ks, int shuffle2,
                          int transposed, void* stream);

/* Layout changes at the module boundary (reference tensors are NCHW).  */
int cdfo_nchw_to_nhwc(const float* in, float* out, int B, int C, int H, int W, int ldo, void* stream);
int cdfo_nhwc_to_nchw(const float* in, int ldi, float* out, int B, int C, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif
