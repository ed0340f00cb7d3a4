// Developer probe: throughput of LDS atomics on gfx950 (one 256-thread workgroup per CU x 4 per CU resident).
// usage: lds_atomic_probe   (prints cycles per wave-instruction for f32 / u32 / u64 adds, conflict-free and random addresses)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int KIND, bool RANDOM>
__global__ __launch_bounds__(256) void probe(const int* idx, unsigned long long* cyc, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned long long win[4096];      // 32 KB
  for (int i = threadIdx.x; i < 4096; i += 256) win[i] = 0;
  __syncthreads();
  int a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = RANDOM ? idx[(blockIdx.x * 16 + k) * 256 + threadIdx.x] & 4095 : (threadIdx.x + 64 * k) & 4095;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (KIND == 0) atomicAdd(reinterpret_cast<float*>(win) + a[k], 1.0f + it);
      else if (KIND == 1) atomicAdd(reinterpret_cast<unsigned*>(win) + a[k], 3u + it);
      else if (KIND == 2) atomicAdd(win + a[k], 3ull + it);
      else { float* p = reinterpret_cast<float*>(win) + a[k]; *p = *p + 1.0f; }       // plain RMW (not atomic): reference rate
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float s = 0.f;
  for (int i = threadIdx.x; i < 4096; i += 256) s += (float)win[i];
  if (s == 12345.678f) sink[0] = s;
}

template <int KIND, bool RANDOM>
void run(const char* name, const int* idx, unsigned long long* cyc, float* sink, int blocks) {
  const int iters = 200;
  probe<KIND, RANDOM><<<blocks, 256>>>(idx, cyc, sink, 10);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<KIND, RANDOM><<<blocks, 256>>>(idx, cyc, sink, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
  // per CU: blocks/256 workgroups x 4 waves x iters x 16 instructions
  const double per_cu_instr = (double)blocks / 256.0 * 4 * iters * 16;
  printf("%-28s %8.3f ms  block cycles %10.0f  -> %6.1f shader cycles per wave-instruction per CU (wall %.1f ns)\n", name, ms, mean,
         mean / (4.0 * iters * 16) / (blocks / 256.0 < 1 ? 1 : 1), ms * 1e6 / per_cu_instr);
}

int main() {
  const int blocks = 1024;    // 4 workgroups per CU
  int* idx; unsigned long long* cyc; float* sink;
  std::vector<int> h(blocks * 16 * 256);
  srand(7);
  for (auto& v : h) v = rand();
  hipMalloc(&idx, h.size() * 4); hipMalloc(&cyc, blocks * 8); hipMalloc(&sink, 4);
  hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0, false>("ds_add_f32 lane-linear", idx, cyc, sink, blocks);
  run<0, true>("ds_add_f32 random", idx, cyc, sink, blocks);
  run<1, false>("ds_add_u32 lane-linear", idx, cyc, sink, blocks);
  run<1, true>("ds_add_u32 random", idx, cyc, sink, blocks);
  run<2, false>("ds_add_u64 lane-linear", idx, cyc, sink, blocks);
  run<2, true>("ds_add_u64 random", idx, cyc, sink, blocks);
  run<3, false>("plain rmw f32 lane-linear", idx, cyc, sink, blocks);
  run<3, true>("plain rmw f32 random", idx, cyc, sink, blocks);
  return 0;
}
