// Micro-benchmark: would SEVERAL LANES PER ENVIRONMENT shorten the Reach kernel's post-barrier chain (load H -> L^T D L -> solve)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -Iinclude -Imycobotgym_amd/csrc tools/microbench/lanes_per_env.hip -o /tmp/lpe && /tmp/lpe
//
// Variant A (production, L = 1): one 12x12 system per lane -- 64 systems per wave -- with the kernels' own ldl_factor<PAT_H> / ldl_solve
//   (compile-time sparsity, everything in registers, the matrix read from the lane's LDS column).
// Variant B (prototype, L = 4): one system per QUAD of lanes -- 16 systems per wave -- rows dealt cyclically to the four lanes, pivot and
//   column entries exchanged by DPP quad_perm broadcasts, the substitutions' column sums by DPP quad reductions; dense 12x12.
// Both loop over `iters` factor + solve rounds on LDS-resident matrices and report shader clocks per round per wave.  The question is
// latency: at 8192 environments every wave has a SIMD to itself either way, so the launch takes as long as ONE wave's chain.
// Result (MI355X, profiles/r03/lanes_per_env.log): a wave issues one FP64 instruction per ~4 clocks whether or not it depends on the
// previous one (tools/microbench/issue_rate.hip), so a chain is as long as its INSTRUCTION COUNT -- and dealing the rows to four lanes
// does not shrink the count per wave, it adds the exchanges.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mcg_dynamics.hpp"

using namespace mcg;

constexpr int NT = NB * (NB + 1) / 2;

// ---- A: production path
__global__ __launch_bounds__(64) void k_l1(unsigned long long* out, double* sink, int iters) {
  __shared__ real lds[NT + NB][64];
  const int lane = threadIdx.x;
  const LaneScratch MS(&lds[0][lane]);
  for (int i = 0; i < NB; i++) for (int j = 0; j <= i; j++)
    MS.st(tri(i, j), i == j ? 20.0 + i + 1e-3 * lane : (PAT_E.nz[i][j] ? 1.0 / (2 + i - j) : 0.0));
  for (int i = 0; i < NB; i++) MS.st(NT + i, 1.0 + 0.1 * i);
  __syncthreads();
  real acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    real L[NT], dinv[NB], x[NB];
    static_for<NB>([&](auto I) { constexpr int i = I;
      static_for<i + 1>([&](auto Jj) { constexpr int j = Jj;
        if constexpr (PAT_E.nz[i][j]) L[tri(i, j)] = MS.ld(tri(i, j)); else if constexpr (PAT_H.nz[i][j]) L[tri(i, j)] = 0.0; }); });
    static_for<NB>([&](auto I) { constexpr int i = I; x[i] = MS.ld(NT + i); });
    ldl_factor<PAT_H>(L, dinv);
    ldl_solve<PAT_H>(L, dinv, x);
    static_for<NB>([&](auto I) { constexpr int i = I; acc += x[i]; });
    asm volatile("" : "+v"(acc));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + lane] = acc;
}

// ---- B: four lanes per system
template <int K> __device__ __forceinline__ real quad_bcast(real v) {      // lane K of every quad to its four lanes
  constexpr int C = K | (K << 2) | (K << 4) | (K << 6);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), C, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), C, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int C> __device__ __forceinline__ real dpp64(real v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), C, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), C, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ real quad_sum(real v) { v += dpp64<0xB1>(v); v += dpp64<0x4E>(v); return v; }

__global__ __launch_bounds__(64) void k_l4(unsigned long long* out, double* sink, int iters) {
  __shared__ real lds[16][NB * NB + NB];             // 16 systems per wave, full rows
  const int lane = threadIdx.x, sys = lane >> 2, q = lane & 3;
  for (int e = q; e < NB * NB; e += 4) { const int i = e / NB, j = e % NB; const int a = i > j ? i : j, b = i > j ? j : i;
    lds[sys][e] = i == j ? 20.0 + i + 1e-3 * sys : (PAT_E.nz[a][b] ? 1.0 / (2 + a - b) : 0.0); }
  for (int i = q; i < NB; i += 4) lds[sys][NB * NB + i] = 1.0 + 0.1 * i;
  __syncthreads();
  real acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    real H[3][NB], g[3], dinv[NB];                   // rows q, q + 4, q + 8
    static_for<3>([&](auto R) { constexpr int r = R;
      static_for<NB>([&](auto Jj) { constexpr int j = Jj; H[r][j] = lds[sys][(4 * r + q) * NB + j]; });
      g[r] = lds[sys][NB * NB + 4 * r + q]; });
    // right-looking L D L^T: the pivot and the column below it travel by quad broadcasts
    static_for<NB>([&](auto Kk) { constexpr int k = Kk; constexpr int ok = k & 3, rk = k >> 2;
      dinv[k] = rcp_nr(quad_bcast<ok>(H[rk][k]));
      real l[3];
      static_for<3>([&](auto R) { constexpr int r = R; l[r] = H[r][k] * dinv[k]; });
      static_for<NB - 1 - k>([&](auto Jj) { constexpr int j = k + 1 + Jj; constexpr int oj = j & 3, rj = j >> 2;
        const real hjk = quad_bcast<oj>(H[rj][k]);                 // H[j][k] = L[j][k] D_k
        static_for<3>([&](auto R) { constexpr int r = R; H[r][j] = fma(-l[r], hjk, H[r][j]); }); });
      static_for<3>([&](auto R) { constexpr int r = R; H[r][k] = (4 * r + q > k) ? l[r] : 0.0; }); });
    // y = L^-1 g (row oriented), z = D^-1 y, x = L^-T z (column sums over the quad)
    static_for<NB>([&](auto Kk) { constexpr int k = Kk; constexpr int ok = k & 3, rk = k >> 2;
      const real yk = quad_bcast<ok>(g[rk]);
      static_for<3>([&](auto R) { constexpr int r = R; g[r] = fma(-H[r][k], yk, g[r]); }); });
    real x[3];
    static_for<3>([&](auto R) { constexpr int r = R; x[r] = g[r]; });
    static_for<NB>([&](auto Kk) { constexpr int k = NB - 1 - Kk; constexpr int ok = k & 3, rk = k >> 2;
      real s = 0;
      static_for<3>([&](auto R) { constexpr int r = R; s = fma(H[r][k], x[r], s); });      // rows above k hold zero in column k
      s = quad_sum(s);
      const real xk = g[rk] * dinv[k] - s;
      x[rk] = (q == ok) ? xk : x[rk]; });
    static_for<3>([&](auto R) { constexpr int r = R; acc += x[r]; });
    asm volatile("" : "+v"(acc));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + lane] = quad_sum(acc);
}

template <class K>
double run(const char* nm, K k, int wgs, int systems_per_wave, double* check) {
  unsigned long long* d; double* s; const int iters = 200;
  hipMalloc(&d, sizeof(unsigned long long) * wgs); hipMalloc(&s, sizeof(double) * wgs * 64);
  hipLaunchKernelGGL(k, dim3(wgs), dim3(64), 0, 0, d, s, 2);
  hipLaunchKernelGGL(k, dim3(wgs), dim3(64), 0, 0, d, s, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(wgs); std::vector<double> hs(wgs * 64);
  hipMemcpy(h.data(), d, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost);
  hipMemcpy(hs.data(), s, sizeof(double) * hs.size(), hipMemcpyDeviceToHost);
  double tot = 0; for (auto v : h) tot += (double)v;
  const double per = tot / wgs / iters;
  printf("%-44s %4d waves: %8.0f clocks per round per wave, %7.1f per system  (checksum lane 0: %.12g)\n", nm, wgs, per, per / systems_per_wave, hs[0] / iters);
  *check = hs[0] / iters;
  hipFree(d); hipFree(s);
  return per;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  double ca, cb;
  for (int wgs : {128, 512, 1024, 4096}) {
    const double a = run("A: one system per lane (production code)", k_l1, wgs, 64, &ca);
    const double b = run("B: one system per quad of lanes (DPP)", k_l4, wgs, 16, &cb);
    printf("   -> a wave of B takes %.2fx a wave of A and carries a quarter of the systems; sum of x agrees to %.1e\n", b / a, (ca - cb * 1.0) / ca);
  }
  return 0;
}
