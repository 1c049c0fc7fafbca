// Micro-benchmark: what does ONE polytope-box pair of the mesh narrow phase (csrc/mcg_mesh.hpp: mesh_box) cost a wave that is alone on its
// SIMD, as in the PickAndPlace kernels?  One 64-lane workgroup per CU, the default polytope block, a box placed so that the pair touches
// (the full path: B, P and E families, contact) or far away (separated by a box axis).  gfx950.
// Measured (round 4): the exhaustive search 10.2 k clocks per touching pair (profiles/r04x/mesh_pair.log; ~1 600 executed instructions of
// which 660 are FP64, 6 clocks each; stopping early: B 2.2 k, P +1.5 k, E +5.0 k, contact +1.6 k), 2.3 k for a pair that a box axis
// separates; with the face bound (profiles/r04y/mesh_pair.log) a contact in the middle of a box face 3.5 k.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -Iinclude -Imycobotgym_amd/csrc tools/microbench/mesh_pair.hip -o /tmp/mesh_pair && /tmp/mesh_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#include "mcg.h"
#include "mcg_dynamics.hpp"
#include "mcg_cube.hpp"
#include "mcg_mesh.hpp"
#include "polytopes_gen.h"

using namespace mcg;

__global__ __launch_bounds__(64) void k_pairs(const double* __restrict__ poly, const double* __restrict__ poses, int iters, int mesh0, unsigned long long* out,
                                              double* sink, int* hits) {
  const int L = threadIdx.x;
  double acc = 0; int nh = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    const int m = (mesh0 + it) % NMESH;
    const double* P = poses + (size_t)(((blockIdx.x + it) & 63) * NMESH + m) * 27;      // Rm 9, pm 3, Rb 9, pb 3, h 3
    real Rm[9], pm[3], Rb[9], pb[3], h[3];
    for (int k = 0; k < 9; k++) { Rm[k] = P[k]; Rb[k] = P[12 + k]; }
    for (int k = 0; k < 3; k++) { pm[k] = P[9 + k]; pb[k] = P[21 + k]; h[k] = P[24 + k]; }
    const MeshTab T = mesh_tab(poly, m);
    real pos[3], nrm[3], dist = 1.0;
    const bool hit = mesh_box(T, L, Rm, pm, Rb, pb, h, (it & 1) != 0, pos, nrm, dist);
    if (hit) { acc += pos[0] + nrm[1] + dist; nh++; }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (L == 0) { out[blockIdx.x] = t1 - t0; hits[blockIdx.x] = nh; }
  sink[blockIdx.x * 64 + L] = acc;
}

static void rot(double* R, unsigned& s) {      // a random rotation (rows)
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / 16777216.0 * 2 - 1; };
  double q[4], n2 = 0; for (double& x : q) { x = rnd(); n2 += x * x; } n2 = std::sqrt(n2); for (double& x : q) x /= n2;
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double M[9] = {1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y), 2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x), 2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)};
  for (int k = 0; k < 9; k++) R[k] = M[k];
}

int main() {
  const int nwg = 256, iters = 2000;
  double* d_poly; hipMalloc(&d_poly, sizeof(kDefaultPolytopes)); hipMemcpy(d_poly, kDefaultPolytopes, sizeof(kDefaultPolytopes), hipMemcpyHostToDevice);
  unsigned long long* d_out; hipMalloc(&d_out, nwg * 8); double* d_sink; hipMalloc(&d_sink, nwg * 64 * 8); int* d_hits; hipMalloc(&d_hits, nwg * 4);
  double* d_pose; hipMalloc(&d_pose, 64 * NMESH * 27 * 8);
  for (int variant = 0; variant < 4; variant++) {            // 0: touching in the middle of the table-sized box's face, 1: a cube-sized box, 2: far away, 3: touching 1 mm from the rim
    std::vector<double> P(64 * NMESH * 27); unsigned s = 12345u;
    for (int w = 0; w < 64; w++) for (int m = 0; m < NMESH; m++) {
      double* q = &P[(size_t)(w * NMESH + m) * 27];
      rot(q, s); q[9] = q[10] = q[11] = 0;                                            // the mesh at the origin, rotated
      for (int k = 0; k < 9; k++) q[12 + k] = (k % 4 == 0) ? 1.0 : 0.0;                // the box axis-aligned
      // the polytope's lowest point along world z
      const double* meta = kDefaultPolytopes + 8 * m; const int nv = (int)meta[0], off = (int)meta[3], vp = (int)meta[4];
      double lo = 1e30, lx = 0, ly = 0;
      for (int v = 0; v < nv; v++) {
        const double x = kDefaultPolytopes[off + v], y = kDefaultPolytopes[off + vp + v], z = kDefaultPolytopes[off + 2 * vp + v];
        const double wz = q[6]*x + q[7]*y + q[8]*z;
        if (wz < lo) { lo = wz; lx = q[0]*x + q[1]*y + q[2]*z; ly = q[3]*x + q[4]*y + q[5]*z; }
      }
      const double hx = variant == 1 ? 0.02 : 0.4, hz = variant == 1 ? 0.02 : 0.1;
      q[24] = hx; q[25] = hx; q[26] = hz;
      q[21] = lx + (variant == 1 ? 0.013 : variant == 3 ? hx - 0.001 : 0.0); q[22] = ly + (variant == 1 ? 0.011 : 0.0);
      q[23] = lo - hz + (variant == 2 ? -0.5 : 0.002);                                 // 2 mm deep, or half a metre below
    }
    hipMemcpy(d_pose, P.data(), P.size() * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k_pairs, dim3(nwg), dim3(64), 0, 0, d_poly, d_pose, iters, rep, d_out, d_sink, d_hits);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> o(nwg); std::vector<int> hh(nwg);
    hipMemcpy(o.data(), d_out, nwg * 8, hipMemcpyDeviceToHost); hipMemcpy(hh.data(), d_hits, nwg * 4, hipMemcpyDeviceToHost);
    double sum = 0; long nh = 0; for (int w = 0; w < nwg; w++) { sum += (double)o[w]; nh += hh[w]; }
    printf("%-48s %8.0f clocks per pair (%d workgroups x %d pairs, %.1f %% touch)\n",
           variant == 0 ? "table-sized box, 2 mm under the lowest vertex:" : variant == 1 ? "cube-sized box, 2 mm under the lowest vertex:" : variant == 2 ? "box far below (separated by a box axis):" : "table-sized box, the vertex 1 mm from its rim:",
           sum / nwg / iters, nwg, iters, 100.0 * nh / ((double)nwg * iters));
  }
  return 0;
}
