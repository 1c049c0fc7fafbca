// Micro-benchmark: v_mfma_f64_16x16x4_f64 on gfx950 -- operand / result layout (checked against a host product) and issue cost
// (independent accumulators back to back, and one dependent chain), one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_f64.hip -o /tmp/mfma_f64 && /tmp/mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double vd4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* A, const double* B, double* D) {      // A[16][4], B[4][16] row-major; D[16][16]
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + l / 16];          // assumed: lane l holds A[i = l % 16][k = l / 16]
  const double b = B[(l / 16) * 16 + l % 16];         // assumed: lane l holds B[k = l / 16][j = l % 16]
  vd4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; v++) D[l * 4 + v] = c[v];    // raw: lane l, element v
}

__global__ void k_time(unsigned long long* out, double* sink, int iters, int dep) {
  const int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  vd4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (dep) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c5, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (l == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1];
}

// permlane16_swap: the sum of two adjacent 16-lane rows in both of them
__global__ void k_swap(double* out) {
  const int l = threadIdx.x;
  const double x = (double)(l / 16 + 1);               // every lane of row r holds r + 1
  unsigned lo = __double2loint(x), hi = __double2hiint(x);
  auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double a = __hiloint2double((int)r1[0], (int)r0[0]), b = __hiloint2double((int)r1[1], (int)r0[1]);
  out[l] = a + b;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  std::vector<double> A(64), B(64), D(256), R(256);
  for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1 + i + 0.01 * k;
  for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 2 + 0.1 * j - k;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
  double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipDeviceSynchronize();
  hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
  // candidate result layouts
  double e1 = 0, e2 = 0;
  for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
    e1 = fmax(e1, fabs(D[l * 4 + v] - R[(4 * (l / 16) + v) * 16 + l % 16]));      // D[i = 4 (l / 16) + v][j = l % 16]
    e2 = fmax(e2, fabs(D[l * 4 + v] - R[((l / 16) + 4 * v) * 16 + l % 16]));      // D[i = l / 16 + 4 v][j = l % 16]
  }
  printf("layout: A[l%%16][l/16], B[l/16][l%%16] assumed;  result = D[4(l/16)+v][l%%16]: max err %.3g;  result = D[l/16+4v][l%%16]: max err %.3g\n", e1, e2);
  printf("lane 0: %g %g %g %g   R[0][0] %g R[1][0] %g R[4][0] %g\n", D[0], D[1], D[2], D[3], R[0], R[16], R[64]);
  for (int dep = 0; dep < 2; dep++) for (int w : {1, 4}) {
    const int wgs = 256, iters = 2000; unsigned long long* d; double* s;
    hipMalloc(&d, 8 * wgs * w); hipMalloc(&s, 8 * wgs * w * 64);
    hipLaunchKernelGGL(k_time, dim3(wgs), dim3(64 * w), 0, 0, d, s, 10, dep);
    hipLaunchKernelGGL(k_time, dim3(wgs), dim3(64 * w), 0, 0, d, s, iters, dep); hipDeviceSynchronize();
    std::vector<unsigned long long> h(wgs * w); hipMemcpy(h.data(), d, 8 * wgs * w, hipMemcpyDeviceToHost);
    double tot = 0; for (auto v : h) tot += (double)v;
    printf("%s v_mfma_f64_16x16x4_f64, %d wave(s) per workgroup: %.1f clocks per instruction per wave\n", dep ? "dependent  " : "independent", w, tot / h.size() / iters / 6);
    hipFree(d); hipFree(s);
  }
  double* o; hipMalloc(&o, 512); hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, o); hipDeviceSynchronize();
  std::vector<double> ho(64); hipMemcpy(ho.data(), o, 512, hipMemcpyDeviceToHost);
  printf("permlane16_swap row-pair sums (expect 3 3 7 7): %g %g %g %g\n", ho[0], ho[16], ho[32], ho[48]);
  return 0;
}
