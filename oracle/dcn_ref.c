/* ORACLE (test infrastructure, not product code): plain-C restatement of the reference's deformable-convolution
 * FORWARD operators, DCNv1 and modulated DCNv2.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this; the product path (libcdfo_hip.so) never does.
 *
 * Follows, in the reference repo:
 *   ops/dcn/src/deform_conv_cuda_kernel.cu:84-113,466-496   bilinear sampling, zero outside the image
 *   ops/dcn/src/deform_conv_cuda_kernel.cu:189-242          DCNv1 im2col: offset layout (h,w interleaved per tap per
 *                                                           deformable group), validity test h_im > -1 && ... < H
 *   ops/dcn/src/deform_conv_cuda_kernel.cu:569-632          DCNv2 im2col: same, value * mask
 *   ops/dcn/src/deform_conv_cuda.cpp:151-258, 486-564       output size, per-group GEMM weight[g] x columns[g], + bias
 *
 * Pinned by the reference's only test, ops/dcn/simple_check.py:8-22 (DCNv1 known answer 81,99,...,225), in
 * tests/test_dcn_oracle.py.  DCNv2 has no reference-owned vector (parity unpinned at that
 * boundary); it is pinned by derived identities (zero offset + unit mask == conv2d, integer offsets == shifted conv,
 * mask == 1 reduces to DCNv1).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float bilinear(const float* im, int H, int W, float h, float w) {
  int h_low = (int)floorf(h), w_low = (int)floorf(w);
  int h_high = h_low + 1, w_high = w_low + 1;
  float lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = im[h_low * W + w_low];
  if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * W + w_high];
  if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * W + w_low];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * W + w_high];
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

/* mask == NULL -> DCNv1 (no modulation); bias == NULL -> no bias.  Returns 0, or -1 on a bad shape. */
int dcn_forward_ref(const float* in, const float* offset, const float* mask, const float* weight, const float* bias,
                    float* out, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw,
                    int dh, int dw, int groups, int dg) {
  if (C % groups || Co % groups || C % dg) return -1;
  const int Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  const int Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (Ho <= 0 || Wo <= 0) return -1;
  const int T = kh * kw, P = Ho * Wo, Cg = C / groups, Cog = Co / groups, Cdg = C / dg;
  float* col = (float*)malloc(sizeof(float) * (size_t)C * T * P); /* the reference's `columns` buffer */
  if (!col) return -1;
  for (int b = 0; b < B; ++b) {
    for (int c = 0; c < C; ++c) {
      const int d = c / Cdg;
      const float* im = in + ((size_t)b * C + c) * H * W;
      const float* off = offset + ((size_t)b * dg + d) * 2 * T * P;
      const float* msk = mask ? mask + ((size_t)b * dg + d) * T * P : NULL;
      for (int i = 0; i < kh; ++i)
        for (int j = 0; j < kw; ++j) {
          const int t = i * kw + j;
          for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo) {
              const int p = ho * Wo + wo;
              const float h_im = (float)(ho * sh - ph + i * dh) + off[(size_t)(2 * t) * P + p];
              const float w_im = (float)(wo * sw - pw + j * dw) + off[(size_t)(2 * t + 1) * P + p];
              float v = 0.f;
              if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) v = bilinear(im, H, W, h_im, w_im);
              if (msk) v *= msk[(size_t)t * P + p];
              col[((size_t)c * T + t) * P + p] = v;
            }
        }
    }
    for (int g = 0; g < groups; ++g)
      for (int o = 0; o < Cog; ++o) {
        const int oc = g * Cog + o;
        const float* wr = weight + (size_t)oc * Cg * T;
        float* orow = out + ((size_t)b * Co + oc) * P;
        for (int p = 0; p < P; ++p) {
          double s = 0.0;
          for (int k = 0; k < Cg * T; ++k) s += (double)wr[k] * (double)col[((size_t)g * Cg * T + k) * P + p];
          orow[p] = (float)s + (bias ? bias[oc] : 0.f);
        }
      }
  }
  free(col);
  return 0;
}
