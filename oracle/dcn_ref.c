/* ORACLE (test infrastructure, not product code): plain-C restatement of the reference's deformable-convolution
 * operators, DCNv1 and modulated DCNv2, forward (below) and backward (second half).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
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

/* One source, two element types: compiled as is for float (the reference's default dtype) and, through dcn_ref_f64.c,
 * for double (the reference instantiates its kernels for float, double and half: deform_conv_cuda_kernel.cu:258). */
#ifndef REAL
#define REAL float
#define NAME(x) x
#define FLOOR_ floorf
#endif

static REAL NAME(bilinear)(const REAL* im, int H, int W, REAL h, REAL w) {
  int h_low = (int)FLOOR_(h), w_low = (int)FLOOR_(w);
  int h_high = h_low + 1, w_high = w_low + 1;
  REAL lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
  REAL v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = im[h_low * W + w_low];
  if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * W + w_high];
  if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * W + w_low];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * W + w_high];
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

/* The forward runs its two loops (columns, then the per-group product) over a few host threads: the oracle is also the checker at the
 * benchmark's frame size (36 calls per CVSR_V7 forward at up to 272x480).  The arithmetic of every output element is unchanged --
 * the same products accumulated in double in the same order (k ascending); the product loop is merely interchanged so that it
 * streams over `columns` rows instead of striding through them. */
#ifndef DCN_REF_THREADS_DEFINED
#define DCN_REF_THREADS_DEFINED
#include <pthread.h>
#include <unistd.h>
typedef struct { void (*fn)(void*, int); void* ctx; int begin, end; } dcn_ref_task;
static void* dcn_ref_worker(void* a) {
  dcn_ref_task* t = (dcn_ref_task*)a;
  for (int i = t->begin; i < t->end; ++i) t->fn(t->ctx, i);
  return NULL;
}
static int dcn_ref_nthreads(void) {
  const char* e = getenv("ORACLE_DCN_THREADS");
  long n = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
  if (n < 1) n = 1;
  if (n > 32) n = 32;
  return (int)n;
}
static void dcn_ref_parallel_for(int n, void (*fn)(void*, int), void* ctx) {
  int nt = dcn_ref_nthreads();
  if (nt > n) nt = n;
  if (nt <= 1) { for (int i = 0; i < n; ++i) fn(ctx, i); return; }
  pthread_t th[32];
  dcn_ref_task tk[32];
  for (int t = 0; t < nt; ++t) {
    tk[t].fn = fn; tk[t].ctx = ctx; tk[t].begin = (int)((long)n * t / nt); tk[t].end = (int)((long)n * (t + 1) / nt);
    if (t && pthread_create(&th[t], NULL, dcn_ref_worker, &tk[t]) != 0) { dcn_ref_worker(&tk[t]); th[t] = 0; tk[t].begin = tk[t].end; }
  }
  dcn_ref_worker(&tk[0]);
  for (int t = 1; t < nt; ++t) if (tk[t].begin != tk[t].end || th[t]) { if (th[t]) pthread_join(th[t], NULL); }
}
#endif

typedef struct {
  const REAL *in, *offset, *mask, *weight, *bias;
  REAL *out, *col;
  int b, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg, Ho, Wo;
} NAME(dcn_fwd_ctx);

static void NAME(dcn_fwd_columns)(void* vc, int c) {          /* one input channel of image b -> its kh*kw rows of `columns` */
  const NAME(dcn_fwd_ctx)* q = (const NAME(dcn_fwd_ctx)*)vc;
  const int H = q->H, W = q->W, kh = q->kh, kw = q->kw, T = kh * kw, P = q->Ho * q->Wo, Cdg = q->C / q->dg, d = c / Cdg;
  const REAL* im = q->in + ((size_t)q->b * q->C + c) * H * W;
  const REAL* off = q->offset + ((size_t)q->b * q->dg + d) * 2 * T * P;
  const REAL* msk = q->mask ? q->mask + ((size_t)q->b * q->dg + d) * T * P : NULL;
  for (int i = 0; i < kh; ++i)
    for (int j = 0; j < kw; ++j) {
      const int t = i * kw + j;
      for (int ho = 0; ho < q->Ho; ++ho)
        for (int wo = 0; wo < q->Wo; ++wo) {
          const int p = ho * q->Wo + wo;
          const REAL h_im = (REAL)(ho * q->sh - q->ph + i * q->dh) + off[(size_t)(2 * t) * P + p];
          const REAL w_im = (REAL)(wo * q->sw - q->pw + j * q->dw) + off[(size_t)(2 * t + 1) * P + p];
          REAL v = 0.f;
          if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) v = NAME(bilinear)(im, H, W, h_im, w_im);
          if (msk) v *= msk[(size_t)t * P + p];
          q->col[((size_t)c * T + t) * P + p] = v;
        }
    }
}

static void NAME(dcn_fwd_product)(void* vc, int oc) {        /* one output channel of image b: weight row x columns[group] + bias */
  const NAME(dcn_fwd_ctx)* q = (const NAME(dcn_fwd_ctx)*)vc;
  const int T = q->kh * q->kw, P = q->Ho * q->Wo, Cg = q->C / q->groups, Cog = q->Co / q->groups, g = oc / Cog;
  const REAL* wr = q->weight + (size_t)oc * Cg * T;
  REAL* orow = q->out + ((size_t)q->b * q->Co + oc) * P;
  double* s = (double*)calloc((size_t)P, sizeof(double));
  if (!s) return;
  for (int k = 0; k < Cg * T; ++k) {
    const double w = (double)wr[k];
    const REAL* cr = q->col + ((size_t)g * Cg * T + k) * P;
    for (int p = 0; p < P; ++p) s[p] += w * (double)cr[p];
  }
  for (int p = 0; p < P; ++p) orow[p] = (REAL)s[p] + (q->bias ? q->bias[oc] : 0.f);
  free(s);
}

/* mask == NULL -> DCNv1 (no modulation); bias == NULL -> no bias.  Returns 0, or -1 on a bad shape. */
int NAME(dcn_forward_ref)(const REAL* in, const REAL* offset, const REAL* mask, const REAL* weight, const REAL* bias,
                    REAL* out, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw,
                    int dh, int dw, int groups, int dg) {
  if (C % groups || Co % groups || C % dg) return -1;
  const int Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  const int Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (Ho <= 0 || Wo <= 0) return -1;
  const int T = kh * kw, P = Ho * Wo;
  REAL* col = (REAL*)malloc(sizeof(REAL) * (size_t)C * T * P); /* the reference's `columns` buffer */
  if (!col) return -1;
  NAME(dcn_fwd_ctx) q = {in, offset, mask, weight, bias, out, col, 0, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg, Ho, Wo};
  for (int b = 0; b < B; ++b) {
    q.b = b;
    dcn_ref_parallel_for(C, NAME(dcn_fwd_columns), &q);
    dcn_ref_parallel_for(Co, NAME(dcn_fwd_product), &q);
  }
  free(col);
  return 0;
}

/* ---------------------------------------------------------------------------------------------------------------
 * BACKWARD restatement.  Follows, in the reference repo:
 *   ops/dcn/src/deform_conv_cuda.cpp:566-680   modulated backward: per image  columns = W^T x grad_output,
 *                                              col2im_coord -> grad_offset / grad_mask (assigned),
 *                                              col2im -> grad_input (accumulated), im2col again,
 *                                              grad_weight += grad_output x columns^T, grad_bias += rowsum
 *   ops/dcn/src/deform_conv_cuda.cpp:260-371, 373-484  DCNv1: the same split over two entry points
 *                                              (backward_input, backward_parameters with `scale`)
 *   ops/dcn/src/deform_conv_cuda_kernel.cu:498-523 / 115-141   gradient weight of a corner (== its bilinear weight;
 *                                              0 when the sample lies at or beyond -1 / H / W)
 *   ops/dcn/src/deform_conv_cuda_kernel.cu:525-567 / 143-187   coordinate weight (d sample / d h, d sample / d w)
 *   ops/dcn/src/deform_conv_cuda_kernel.cu:634-766             col2im and col2im_coord loops
 * The reference holds no vector for the backward; tests/test_dcn_oracle.py pins this restatement against REAL64
 * autograd through an independent gather-based statement of the forward.
 * Any of gin / goff / gmask / gw / gbias may be NULL (skipped).  gin, gw, gbias are ACCUMULATED into (the reference's
 * callers zero them first, deform_conv.py:71-72,85,154-158); goff, gmask are assigned. */
int NAME(dcn_backward_ref)(const REAL* in, const REAL* offset, const REAL* mask, const REAL* weight, const REAL* gout,
                     REAL* gin, REAL* goff, REAL* gmask, REAL* gw, REAL* gbias, int B, int C, int H, int W, int Co,
                     int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int groups, int dg, REAL scale) {
  if (C % groups || Co % groups || C % dg) return -1;
  const int Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  const int Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (Ho <= 0 || Wo <= 0) return -1;
  const int T = kh * kw, P = Ho * Wo, Cg = C / groups, Cog = Co / groups, Cdg = C / dg;
  REAL* col = (REAL*)malloc(sizeof(REAL) * (size_t)C * T * P);
  if (!col) return -1;
  for (int b = 0; b < B; ++b) {
    /* columns[g] = weight[g]^T x grad_output[b][g]   (cpp:617-620) */
    for (int c = 0; c < C; ++c) {
      const int g = c / Cg, cl = c - g * Cg;
      for (int t = 0; t < T; ++t)
        for (int p = 0; p < P; ++p) {
          double s = 0.0;
          for (int o = 0; o < Cog; ++o)
            s += (double)weight[((size_t)(g * Cog + o) * Cg + cl) * T + t] * (double)gout[((size_t)b * Co + g * Cog + o) * P + p];
          col[((size_t)c * T + t) * P + p] = (REAL)s;
        }
    }
    for (int d = 0; d < dg; ++d) {
      const REAL* off = offset + ((size_t)b * dg + d) * 2 * T * P;
      const REAL* msk = mask ? mask + ((size_t)b * dg + d) * T * P : NULL;
      for (int i = 0; i < kh; ++i)
        for (int j = 0; j < kw; ++j) {
          const int t = i * kw + j;
          for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo) {
              const int p = ho * Wo + wo;
              const REAL h_im = (REAL)(ho * sh - ph + i * dh) + off[(size_t)(2 * t) * P + p];
              const REAL w_im = (REAL)(wo * sw - pw + j * dw) + off[(size_t)(2 * t + 1) * P + p];
              const REAL m = msk ? msk[(size_t)t * P + p] : 1.f;
              const int valid = !(h_im <= -1 || w_im <= -1 || h_im >= H || w_im >= W);
              REAL vh = 0.f, vw = 0.f, mv = 0.f;
              if (valid) {
                const int hl = (int)FLOOR_(h_im), wl = (int)FLOOR_(w_im), hhi = hl + 1, whi = wl + 1;
                const REAL lh = h_im - hl, lw = w_im - wl, hh = 1 - lh, hw = 1 - lw;
                const int c1 = hl >= 0 && wl >= 0, c2 = hl >= 0 && whi <= W - 1, c3 = hhi <= H - 1 && wl >= 0,
                          c4 = hhi <= H - 1 && whi <= W - 1;
                for (int cc = 0; cc < Cdg; ++cc) {
                  const int c = d * Cdg + cc;
                  const REAL* im = in + ((size_t)b * C + c) * H * W;
                  const REAL cg = col[((size_t)c * T + t) * P + p];
                  const REAL v1 = c1 ? im[hl * W + wl] : 0.f, v2 = c2 ? im[hl * W + whi] : 0.f,
                              v3 = c3 ? im[hhi * W + wl] : 0.f, v4 = c4 ? im[hhi * W + whi] : 0.f;
                  mv += cg * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);       /* cu:733-736 */
                  vh += (-hw * v1 - lw * v2 + hw * v3 + lw * v4) * cg * m;                       /* cu:543-553 */
                  vw += (-hh * v1 + hh * v2 - lh * v3 + lh * v4) * cg * m;                       /* cu:554-564 */
                  if (gin) {                                                                     /* cu:667-683 */
                    REAL* gi = gin + ((size_t)b * C + c) * H * W;
                    const REAL tg = cg * m;
                    if (c1) gi[hl * W + wl] += hh * hw * tg;
                    if (c2) gi[hl * W + whi] += hh * lw * tg;
                    if (c3) gi[hhi * W + wl] += lh * hw * tg;
                    if (c4) gi[hhi * W + whi] += lh * lw * tg;
                  }
                }
              }
              if (goff) {
                goff[(((size_t)b * dg + d) * 2 * T + 2 * t) * P + p] = vh;
                goff[(((size_t)b * dg + d) * 2 * T + 2 * t + 1) * P + p] = vw;
              }
              if (gmask) gmask[(((size_t)b * dg + d) * T + t) * P + p] = mv;
            }
        }
    }
    if (gw || gbias) {
      /* im2col again (cpp:637-641), then grad_weight[g] += grad_output[b][g] x columns[g]^T (cpp:650-655) */
      for (int c = 0; c < C; ++c) {
        const int d = c / Cdg;
        const REAL* im = in + ((size_t)b * C + c) * H * W;
        const REAL* off = offset + ((size_t)b * dg + d) * 2 * T * P;
        const REAL* msk = mask ? mask + ((size_t)b * dg + d) * T * P : NULL;
        for (int i = 0; i < kh; ++i)
          for (int j = 0; j < kw; ++j) {
            const int t = i * kw + j;
            for (int ho = 0; ho < Ho; ++ho)
              for (int wo = 0; wo < Wo; ++wo) {
                const int p = ho * Wo + wo;
                const REAL h_im = (REAL)(ho * sh - ph + i * dh) + off[(size_t)(2 * t) * P + p];
                const REAL w_im = (REAL)(wo * sw - pw + j * dw) + off[(size_t)(2 * t + 1) * P + p];
                REAL v = 0.f;
                if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) v = NAME(bilinear)(im, H, W, h_im, w_im);
                if (msk) v *= msk[(size_t)t * P + p];
                col[((size_t)c * T + t) * P + p] = v;
              }
          }
      }
      for (int oc = 0; oc < Co; ++oc) {
        const int g = oc / Cog;
        const REAL* go = gout + ((size_t)b * Co + oc) * P;
        if (gw)
          for (int k = 0; k < Cg * T; ++k) {
            double s = 0.0;
            const REAL* cr = col + ((size_t)g * Cg * T + k) * P;
            for (int p = 0; p < P; ++p) s += (double)go[p] * (double)cr[p];
            gw[(size_t)oc * Cg * T + k] += scale * (REAL)s;
          }
        if (gbias) {
          double s = 0.0;
          for (int p = 0; p < P; ++p) s += go[p];
          gbias[oc] += (REAL)s;
        }
      }
    }
  }
  free(col);
  return 0;
}
