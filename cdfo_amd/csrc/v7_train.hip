// Backward-side kernels of the operators only CVSR_V7's forward has (v7_ops.hip), for the module's autograd path
// (cdfo_amd/cvsr_v7_train.py; the reference class is trainable: arch/SIDECVSR_our.py:4215-4367).  fp32 pixel-major rows of 64
// channels, 16 lanes per pixel with one float4 each like the forward kernels; every reduction has a fixed order (no atomics).
//
//   cdfo_chan_pool_bwd    adjoint of ChannelPool (arch.py:1883-1885): dx[c] = dmean / 64 + [c == argmax] * dmax
//   cdfo_mul_plane        out[p][c] = x[p][c] * plane[p]          (SpatialAttention's product, arch.py:2729; also its adjoint w.r.t. x)
//   cdfo_dot_plane        out[p] = sum_c a[p][c] * b[p][c]        (adjoint w.r.t. the plane)
//   cdfo_gumbel_softmax   r[p][c] = softmax_c(v[b][c] - log(-log u[b][c][p]))   (RDAB.gumbel_softmax, arch.py:2813-2822; u NCHW)
//   cdfo_softmax64_bwd    dz[p][c] = r[p][c] * (dm[p][c] - sum_c' r[p][c'] dm[p][c'])
#include "common.h"

namespace {

__device__ __forceinline__ float row16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void chan_pool_bwd_kernel(const float* __restrict__ x, int ld, const float* __restrict__ dpool,
                                                            long long npix, float* __restrict__ dx, int ldo) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long p = gid >> 4;
  const int l = (int)(gid & 15);
  if (p >= npix) return;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * ld + l * 4);
  float mx = v[0];
  int am = l * 4;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (v[k] > mx) { mx = v[k]; am = l * 4 + k; }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {            // first maximum in channel order, like torch.max(dim)
    const float om = __shfl_xor(mx, o, 64);
    const int oa = __shfl_xor(am, o, 64);
    if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
  }
  const float dmax = dpool[p * 2], dmean = dpool[p * 2 + 1] * (1.f / 64.f);
  f32x4 g;
#pragma unroll
  for (int k = 0; k < 4; ++k) g[k] = dmean + (l * 4 + k == am ? dmax : 0.f);
  *reinterpret_cast<f32x4*>(dx + p * ldo + l * 4) = g;
}

__global__ __launch_bounds__(256) void mul_plane_kernel(const float* __restrict__ x, int ld, const float* __restrict__ plane,
                                                        long long npix, float* __restrict__ out, int ldo) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long p = gid >> 4;
  const int l = (int)(gid & 15);
  if (p >= npix) return;
  const float g = plane[p];
  f32x4 v = *reinterpret_cast<const f32x4*>(x + p * ld + l * 4);
  v[0] *= g; v[1] *= g; v[2] *= g; v[3] *= g;
  *reinterpret_cast<f32x4*>(out + p * ldo + l * 4) = v;
}

__global__ __launch_bounds__(256) void dot_plane_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                        long long npix, float* __restrict__ out) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long p = gid >> 4;
  const int l = (int)(gid & 15);
  if (p >= npix) return;
  const f32x4 va = *reinterpret_cast<const f32x4*>(a + p * lda + l * 4);
  const f32x4 vb = *reinterpret_cast<const f32x4*>(b + p * ldb + l * 4);
  const float s = row16_sum((va[0] * vb[0] + va[1] * vb[1]) + (va[2] * vb[2] + va[3] * vb[3]));
  if (l == 0) out[p] = s;
}

// one wave = 64 consecutive pixels of one image: the noise is read along pixels (coalesced in its NCHW tensor), transposed
// through LDS, the softmax rows are written 16 lanes per pixel (coalesced)
__global__ __launch_bounds__(64) void gumbel_softmax_kernel(const float* __restrict__ v, const float* __restrict__ u, int B,
                                                            long long P, float* __restrict__ out, int ldo) {
  __shared__ float z[64][65];
  const int lane = threadIdx.x;
  const long long tiles = (P + 63) / 64;
  const long long b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * 64;
  const long long p = p0 + lane;
  const float* vb = v + b * 64;
  if (p < P) {
    const float* ub = u + b * 64 * P + p;
    float m = -INFINITY;
    for (int c = 0; c < 64; ++c) {
      const float t = vb[c] - logf(-logf(ub[(long long)c * P]));
      z[c][lane] = t;
      m = fmaxf(m, t);
    }
    float s = 0.f;
    for (int c = 0; c < 64; ++c) {
      const float e = expf(z[c][lane] - m);
      z[c][lane] = e;
      s += e;
    }
    const float inv = 1.f / s;
    for (int c = 0; c < 64; ++c) z[c][lane] *= inv;
  }
  __syncthreads();
  for (int i = 0; i < 16; ++i) {
    const int idx = i * 64 + lane, px = idx >> 4, c4 = (idx & 15) * 4;
    if (p0 + px >= P) continue;
    const f32x4 t = {z[c4][px], z[c4 + 1][px], z[c4 + 2][px], z[c4 + 3][px]};
    *reinterpret_cast<f32x4*>(out + (b * P + p0 + px) * ldo + c4) = t;
  }
}

__global__ __launch_bounds__(256) void softmax64_bwd_kernel(const float* __restrict__ r, int ldr, const float* __restrict__ dm,
                                                            int ldm, long long npix, float* __restrict__ dz, int ldo) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long p = gid >> 4;
  const int l = (int)(gid & 15);
  if (p >= npix) return;
  const f32x4 vr = *reinterpret_cast<const f32x4*>(r + p * ldr + l * 4);
  const f32x4 vg = *reinterpret_cast<const f32x4*>(dm + p * ldm + l * 4);
  const float s = row16_sum((vr[0] * vg[0] + vr[1] * vg[1]) + (vr[2] * vg[2] + vr[3] * vg[3]));
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = vr[k] * (vg[k] - s);
  *reinterpret_cast<f32x4*>(dz + p * ldo + l * 4) = o;
}

inline bool rows_ok(const void* p, int ld) { return p && ld >= 64 && ld % 4 == 0 && aligned16(p); }

}  // namespace

extern "C" int cdfo_chan_pool_bwd(const float* x, int ld, const float* dpooled, long long npix, float* dx, int ldo, void* stream) {
  if (npix <= 0 || !rows_ok(x, ld) || !rows_ok(dx, ldo) || !dpooled) return CDFO_EINVAL;
  hipLaunchKernelGGL(chan_pool_bwd_kernel, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, ld, dpooled, npix, dx, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_mul_plane(const float* x, int ld, const float* plane, long long npix, float* out, int ldo, void* stream) {
  if (npix <= 0 || !rows_ok(x, ld) || !rows_ok(out, ldo) || !plane) return CDFO_EINVAL;
  hipLaunchKernelGGL(mul_plane_kernel, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     ld, plane, npix, out, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_dot_plane(const float* a, int lda, const float* b, int ldb, long long npix, float* out, void* stream) {
  if (npix <= 0 || !rows_ok(a, lda) || !rows_ok(b, ldb) || !out) return CDFO_EINVAL;
  hipLaunchKernelGGL(dot_plane_kernel, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                     lda, b, ldb, npix, out);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_gumbel_softmax(const float* v, const float* u, int B, long long P, float* out, int ldo, void* stream) {
  if (B <= 0 || P <= 0 || !v || !u || !rows_ok(out, ldo)) return CDFO_EINVAL;
  const long long blocks = (long long)B * ((P + 63) / 64);
  if (blocks > 0x7fffffffLL) return CDFO_EINVAL;
  hipLaunchKernelGGL(gumbel_softmax_kernel, dim3((unsigned)blocks), dim3(64), 0, static_cast<hipStream_t>(stream), v, u, B, P, out, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_softmax64_bwd(const float* r, int ldr, const float* dm, int ldm, long long npix, float* dz, int ldo, void* stream) {
  if (npix <= 0 || !rows_ok(r, ldr) || !rows_ok(dm, ldm) || !rows_ok(dz, ldo)) return CDFO_EINVAL;
  hipLaunchKernelGGL(softmax64_bwd_kernel, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     r, ldr, dm, ldm, npix, dz, ldo);
  CDFO_LAUNCH_CHECK();
  return 0;
}
