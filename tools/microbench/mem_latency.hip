// Micro-benchmark: latency seen by ONE wave per CU for a dependent reload from (a) scratch, (b) LDS, with a per-wave
// footprint like the step kernel's spill area.  gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mem_latency.hip -o /tmp/mem_latency && /tmp/mem_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SLOTS>
__global__ __launch_bounds__(64) void k_scratch(unsigned long long* out, double* sink, int iters, int stride) {
  double priv[SLOTS];
  for (int k = 0; k < SLOTS; k++) priv[k] = (double)((k * stride) % SLOTS);       // a permutation walk when gcd(stride, SLOTS) = 1
  asm volatile("" ::: "memory");
  int idx = threadIdx.x % 7 == 99 ? 1 : 0;                                           // runtime value the compiler cannot fold
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) idx = (int)priv[idx];                              // dependent chain of scratch loads (dynamic index)
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = idx;
}

template <int SLOTS>
__global__ __launch_bounds__(64) void k_lds(unsigned long long* out, double* sink, int iters, int stride) {
  __shared__ double sh[SLOTS][64];
  for (int k = 0; k < SLOTS; k++) sh[k][threadIdx.x] = (double)((k * stride) % SLOTS);
  __syncthreads();
  int idx = threadIdx.x % 7 == 99 ? 1 : 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) idx = (int)sh[idx][threadIdx.x];
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = idx;
}

template <class K>
void run(const char* nm, K k, int slots, int wgs) {
  unsigned long long* d; double* sink; const int iters = 2000;
  if (hipMalloc(&d, sizeof(unsigned long long) * wgs) != hipSuccess || hipMalloc(&sink, sizeof(double) * wgs * 64) != hipSuccess) { printf("alloc failed\n"); return; }
  hipLaunchKernelGGL(k, dim3(wgs), dim3(64), 0, 0, d, sink, 16, 7);
  hipLaunchKernelGGL(k, dim3(wgs), dim3(64), 0, 0, d, sink, iters, 7);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: kernel failed\n", nm); return; }
  std::vector<unsigned long long> h(wgs);
  (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  printf("%-44s slots %4d (%3d KB/wave), WGs %4d: %7.1f clk per dependent load (incl. ~12 clk of cvt/address VALU)\n", nm, slots, slots * 8 * 64 / 1024, wgs, s / wgs / iters);
  (void)hipFree(d); (void)hipFree(sink);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  for (int wgs : {128, 256}) {
    run("scratch (private array, dynamic index)", k_scratch<32>, 32, wgs);
    run("scratch (private array, dynamic index)", k_scratch<128>, 128, wgs);
    run("scratch (private array, dynamic index)", k_scratch<256>, 256, wgs);
    run("scratch (private array, dynamic index)", k_scratch<512>, 512, wgs);
    run("LDS column per lane", k_lds<32>, 32, wgs);
    run("LDS column per lane", k_lds<100>, 100, wgs);
  }
  return 0;
}
