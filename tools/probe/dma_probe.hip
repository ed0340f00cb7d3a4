// Developer probe (GPU box only): buffer_load_dwordx4 ... lds on gfx950 -- LDS destinations above 64 KiB, zero fill of
// out-of-range lanes.  Build: hipcc -O3 --offload-arch=gfx950 dma_probe.hip -o dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(64) void dma_test(const float* src, unsigned nbytes, const unsigned* offs,
                                                          float* out, unsigned lds_off, unsigned soff) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 40000; i += 64) reinterpret_cast<float*>(smem)[i] = -7.f;
  __syncthreads();
  i32x4 rsrc;
  const unsigned long long p = reinterpret_cast<unsigned long long>(src);
  rsrc[0] = (int)(unsigned)p; rsrc[1] = (int)(unsigned)(p >> 32); rsrc[2] = (int)nbytes; rsrc[3] = 0x00020000;
  const unsigned voff = offs[lane];
  const unsigned ldsaddr = (unsigned)(unsigned long long)(smem) + lds_off;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(ldsaddr), "s"(soff) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 v = *reinterpret_cast<f32x4*>(smem + lds_off + lane * 16);
  *reinterpret_cast<f32x4*>(out + lane * 4) = v;
}
int main() {
  const int N = 1 << 16;
  std::vector<float> h(N);
  for (int i = 0; i < N; ++i) h[i] = (float)i;
  float *d, *o; unsigned* offs;
  hipMalloc(&d, N * 4); hipMalloc(&o, 256 * 4); hipMalloc(&offs, 64 * 4);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> ho(64);
  for (int l = 0; l < 64; ++l) ho[l] = (l % 5 == 4) ? 0x80000000u : (unsigned)((63 - l) * 32);   // permuted + OOB lanes
  hipMemcpy(offs, ho.data(), 64 * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)dma_test, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
  const unsigned lds_offs[3] = {0, 70000 / 16 * 16, 150000 / 16 * 16};
  for (int t = 0; t < 3; ++t) for (unsigned soff : {0u, 4096u}) {
    hipLaunchKernelGGL(dma_test, dim3(1), dim3(64), 160000, 0, d, (unsigned)(N * 4), offs, o, lds_offs[t], soff);
    std::vector<float> r(256);
    hipError_t e = hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) {
      const float want = (l % 5 == 4) ? 0.f : (float)((63 - l) * 8 + soff / 4 + k);
      if (r[l * 4 + k] != want) { if (bad < 4) printf("  lane %d k %d got %g want %g\n", l, k, r[l * 4 + k], want); ++bad; }
    }
    printf("lds_off %u soff %u: err=%d mismatches=%d\n", lds_offs[t], soff, (int)e, bad);
  }
  return 0;
}
