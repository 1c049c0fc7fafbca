// Micro-benchmark: cycles per instruction of ONE wave per SIMD for f64 VALU streams (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate
// Answers: is a single wave limited by VALU issue (4 clk / wave64 f64 op), by dependent-op latency, or by instruction fetch
// (8-byte VOP3 encodings vs 4-byte VOP2)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))

// 8 instructions per group; each kernel body = 256 groups = 2048 instructions, looped ITER times
#define DEP_FMA   "v_fma_f64 v[0:1], v[16:17], v[18:19], v[0:1]\n"
#define IND_FMA   "v_fma_f64 v[0:1], v[16:17], v[18:19], v[0:1]\n v_fma_f64 v[2:3], v[16:17], v[18:19], v[2:3]\n v_fma_f64 v[4:5], v[16:17], v[18:19], v[4:5]\n v_fma_f64 v[6:7], v[16:17], v[18:19], v[6:7]\n v_fma_f64 v[8:9], v[16:17], v[18:19], v[8:9]\n v_fma_f64 v[10:11], v[16:17], v[18:19], v[10:11]\n v_fma_f64 v[12:13], v[16:17], v[18:19], v[12:13]\n v_fma_f64 v[14:15], v[16:17], v[18:19], v[14:15]\n"
#define IND_FMAC  "v_fmac_f64_e32 v[0:1], v[16:17], v[18:19]\n v_fmac_f64_e32 v[2:3], v[16:17], v[18:19]\n v_fmac_f64_e32 v[4:5], v[16:17], v[18:19]\n v_fmac_f64_e32 v[6:7], v[16:17], v[18:19]\n v_fmac_f64_e32 v[8:9], v[16:17], v[18:19]\n v_fmac_f64_e32 v[10:11], v[16:17], v[18:19]\n v_fmac_f64_e32 v[12:13], v[16:17], v[18:19]\n v_fmac_f64_e32 v[14:15], v[16:17], v[18:19]\n"
#define DEP_FMAC  "v_fmac_f64_e32 v[0:1], v[16:17], v[18:19]\n"
#define IND_MUL   "v_mul_f64 v[0:1], v[16:17], v[18:19]\n v_mul_f64 v[2:3], v[16:17], v[18:19]\n v_mul_f64 v[4:5], v[16:17], v[18:19]\n v_mul_f64 v[6:7], v[16:17], v[18:19]\n v_mul_f64 v[8:9], v[16:17], v[18:19]\n v_mul_f64 v[10:11], v[16:17], v[18:19]\n v_mul_f64 v[12:13], v[16:17], v[18:19]\n v_mul_f64 v[14:15], v[16:17], v[18:19]\n"
#define IND_F32   "v_fmac_f32_e32 v0, v16, v18\n v_fmac_f32_e32 v2, v16, v18\n v_fmac_f32_e32 v4, v16, v18\n v_fmac_f32_e32 v6, v16, v18\n v_fmac_f32_e32 v8, v16, v18\n v_fmac_f32_e32 v10, v16, v18\n v_fmac_f32_e32 v12, v16, v18\n v_fmac_f32_e32 v14, v16, v18\n"
#define MIX_SALU  "v_fma_f64 v[0:1], v[16:17], v[18:19], v[0:1]\n s_add_u32 s20, s20, 1\n v_fma_f64 v[2:3], v[16:17], v[18:19], v[2:3]\n s_add_u32 s21, s21, 1\n v_fma_f64 v[4:5], v[16:17], v[18:19], v[4:5]\n s_add_u32 s22, s22, 1\n v_fma_f64 v[6:7], v[16:17], v[18:19], v[6:7]\n s_add_u32 s23, s23, 1\n"

#define CLOBBER "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","s20","s21","s22","s23"

#define KERNEL(name, body, per_group, groups)                                                      \
  __global__ void name(unsigned long long* out, int iters) {                                       \
    asm volatile("v_mov_b32 v16, 0\n v_mov_b32 v17, 0x3ff00000\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0x3ff00000\n" ::: CLOBBER); \
    const unsigned long long t0 = __builtin_readcyclecounter();                                    \
    for (int i = 0; i < iters; i++) asm volatile(groups(body) ::: CLOBBER);                        \
    const unsigned long long t1 = __builtin_readcyclecounter();                                    \
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;  \
  }                                                                                                \
  static const int name##_n = per_group;

KERNEL(k_dep_fma, DEP_FMA, 1, REP256)
KERNEL(k_ind_fma, IND_FMA, 8, REP256)
KERNEL(k_ind_fma_big, IND_FMA, 8, REP4(REP256(IND_FMA)) REP256)       // 5 x 2048 x 8 B = 80 KB of code: beyond the I-cache
KERNEL(k_ind_fmac, IND_FMAC, 8, REP256)
KERNEL(k_dep_fmac, DEP_FMAC, 1, REP256)
KERNEL(k_ind_mul, IND_MUL, 8, REP256)
KERNEL(k_ind_f32, IND_F32, 8, REP256)
KERNEL(k_mix_salu, MIX_SALU, 8, REP256)

template <class K>
void run(const char* nm, K k, int per_group, int groups, int waves_per_wg, int wgs) {
  unsigned long long* d; const int iters = 50;
  hipMalloc(&d, sizeof(unsigned long long) * wgs * waves_per_wg);
  hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * waves_per_wg), 0, 0, d, 2);       // warm the instruction cache
  hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * waves_per_wg), 0, 0, d, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(wgs * waves_per_wg);
  hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  const double n = (double)iters * groups * per_group;
  printf("%-34s waves/WG %d, WGs %4d: %6.2f clk (s_memtime) per instruction per wave\n", nm, waves_per_wg, wgs, s / h.size() / n);
  hipFree(d);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  for (int w : {1, 4, 8, 16}) {      // 1 wave per CU, 1 per SIMD, 2 per SIMD, 4 per SIMD
    run("dependent v_fma_f64 (8 B)", k_dep_fma, k_dep_fma_n, 256, w, 256);
    run("independent v_fma_f64 (8 B)", k_ind_fma, k_ind_fma_n, 256, w, 256);
    run("independent v_fma_f64, 80 KB loop", k_ind_fma_big, k_ind_fma_big_n, 256 * 5, w, 256);
    run("independent v_fmac_f64_e32 (4 B)", k_ind_fmac, k_ind_fmac_n, 256, w, 256);
    run("dependent v_fmac_f64_e32 (4 B)", k_dep_fmac, k_dep_fmac_n, 256, w, 256);
    run("independent v_mul_f64 (8 B)", k_ind_mul, k_ind_mul_n, 256, w, 256);
    run("independent v_fmac_f32_e32 (4 B)", k_ind_f32, k_ind_f32_n, 256, w, 256);
    run("v_fma_f64 / s_add_u32 alternating", k_mix_salu, k_mix_salu_n, 256, w, 256);
  }
  return 0;
}
