// NCHW <-> pixel-major (NHWC) transposes at the module boundary.  The reference's tensors are NCHW
// (arch/SIDECVSR_our.py:4409); everything inside the HIP path is pixel-major so that a pixel's channels are
// one contiguous 256-byte run.  LDS-tiled 32x32 transpose, coalesced on both sides.
#include "common.h"

namespace {

// in: [B][R][Cc] with row pitch ldi  ->  out: [B][Cc][R] with row pitch ldo   (generic 2-D transpose per batch)
__global__ void transpose2d(const float* __restrict__ in, long long in_bstride, int ldi, float* __restrict__ out,
                            long long out_bstride, int ldo, int R, int Cc) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* ip = in + b * in_bstride;
  float* op = out + b * out_bstride;
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < R && c < Cc) ? ip[(long long)r * ldi + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < Cc && r < R) op[(long long)c * ldo + r] = tile[tx][k];
  }
}


__global__ void swap_outer_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int B, int N, long long blk4) {
  const long long total = (long long)B * N * blk4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long e = i % blk4;
    const long long bn = i / blk4;
    const int n = bn % N, b = bn / N;
    out[((long long)n * B + b) * blk4 + e] = in[i];
  }
}

// slots[0] = max over the finite |x| (bit pattern of a non-negative float: integer max == float max), slots[1] |= 1 if any
// element is NaN or infinite.  The caller zeroes the slots; several probes may share one buffer (different slot pairs).
__global__ __launch_bounds__(256) void range_probe_kernel(const f32x4* __restrict__ x, long long n4, unsigned* __restrict__ slots) {
  float m = 0.f;
  bool bad = false;
  // four independent 16-byte loads in flight per thread (one per trip left the kernel latency-bound: 267 MB in 0.137 ms)
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a = fabsf(v[u][k]);
        if (a < 3.0e38f) m = fmaxf(m, a); else bad = true;      // NaN compares false: counted as bad
      }
  }
  for (; i < n4; i += stride) {
    const f32x4 v = x[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = fabsf(v[k]);
      if (a < 3.0e38f) m = fmaxf(m, a); else bad = true;
    }
  }
  m = wave_max(m);
  const bool any_bad = __any(bad);
  if ((threadIdx.x & 63) == 0) {
    if (m > 0.f) atomicMax(slots, __float_as_uint(m));
    if (any_bad) atomicOr(slots + 1, 1u);
  }
}

}  // namespace

// Range probe of a dense fp32 buffer (n % 4 == 0): see range_probe_kernel.  Used by CVSR_V8's fp16 range guard.
extern "C" int cdfo_range_probe(const float* x, long long n, void* slots2, void* stream) {
  if (!x || n <= 0 || n % 4 || !slots2) return CDFO_EINVAL;
  if (!aligned16(x)) return CDFO_EALIGN;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_LAYOUT, 0, 4.0 * (double)n);
  hipLaunchKernelGGL(range_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const f32x4*>(x), n / 4, static_cast<unsigned*>(slots2));
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_nchw_to_nhwc(const float* in, float* out, int B, int C, int H, int W, int ldo, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || ldo < C) return CDFO_EINVAL;
  const int P = H * W;
  dim3 grid(cdiv(P, 32), cdiv(C, 32), B);  // in rows = channels (R=C), cols = pixels (Cc=P)
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_LAYOUT, 0, 8.0*(double)B*C*H*W);
  hipLaunchKernelGGL(transpose2d, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in, (long long)C * P, P, out,
                     (long long)P * ldo, ldo, C, P);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_nhwc_to_nchw(const float* in, int ldi, float* out, int B, int C, int H, int W, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || ldi < C) return CDFO_EINVAL;
  const int P = H * W;
  dim3 grid(cdiv(C, 32), cdiv(P, 32), B);  // in rows = pixels (R=P), cols = channels (Cc=C)
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_LAYOUT, 0, 8.0*(double)B*C*H*W);
  hipLaunchKernelGGL(transpose2d, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in, (long long)P * ldi, ldi, out,
                     (long long)C * P, P, P, C);
  CDFO_LAUNCH_CHECK();
  return 0;
}

extern "C" int cdfo_swap_outer(const float* in, float* out, int B, int N, long long block, void* stream) {
  if (B <= 0 || N <= 0 || block <= 0 || block % 4) return CDFO_EINVAL;
  if (!aligned16(in) || !aligned16(out)) return CDFO_EALIGN;
  long long blocks = ((long long)B * N * (block / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  CdfoProfScope prof(static_cast<hipStream_t>(stream), KID_LAYOUT, 0, 8.0*(double)B*N*block);
  hipLaunchKernelGGL(swap_outer_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const f32x4*>(in), reinterpret_cast<f32x4*>(out), B, N, block / 4);
  CDFO_LAUNCH_CHECK();
  return 0;
}
