// mcg_dynamics.hpp -- per-environment forward dynamics of the MyCobot-280 arm + gripper, one env per lane.
//
// Replaces, for this one model, the MuJoCo pipeline the reference runs inside
// `mujoco.mj_step(self.model, self.data, nstep=self.frame_skip)` (/root/reference/mycobotgym/envs/mycobot.py:170,193):
// kinematics, composite-rigid-body mass matrix, recursive Newton-Euler bias forces, affine actuators with
// ctrl/force clamps, soft equality + joint-limit constraints (primal Newton), implicit-damping Euler.
//
// It is NOT a port of MuJoCo's data flow.  The model's structure is compiled in:
//   * every hinge axis is a signed coordinate axis of its own body frame and every moving body has an identity
//     rest orientation, so a parent<-child rotation is a 2x2 rotation in one coordinate plane (4 mul + 2 add);
//   * the recursion runs in body-local frames about the joint anchors (classical-acceleration RNEA, CRBA by
//     propagating unit-acceleration wrenches up the chain), so no world transforms are formed per sub-step;
//   * the gripper's two loop closures are written in the link6 frame, where the mechanism is planar (all six
//     gripper axes are +-y of link6) -- legal because a connect's three rows share one isotropic D.
// Numbers (offsets, inertias, gains, solver parameters) come from the mcg_model block in device memory; its
// reads are wave-uniform and become scalar loads.
#pragma once
#ifndef MCG_DUP
#define MCG_DUP 0       // development probes of the four-wave kernel's critical path: a stage executed twice (1 cube wave's solver numbers, 2 M / RNE
#endif                  // waves' arm meshes, 3 robot wave's H_eq assembly) -- a stage off the critical path costs nothing when doubled

#include <type_traits>
#include <hip/hip_runtime.h>
#include "mcg.h"

namespace mcg {

typedef double real;

constexpr int NB = 12;                                                 // moving robot bodies = robot dofs
constexpr int AXK[NB] = {2, 0, 0, 0, 2, 0, 1, 1, 1, 1, 1, 1};         // axis index of joint i in its body frame
constexpr int AXS[NB] = {-1, -1, 1, -1, -1, -1, 1, -1, -1, 1, 1, 1};   // axis sign
constexpr int PAR[NB] = {-1, 0, 1, 2, 3, 4, 5, 6, 5, 8, 5, 5};         // parent body (-1 = static base)
constexpr real MINVAL = 1e-15, MINIMP = 1e-4, MAXIMP = 0.9999;

#define MCG_DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------------ helpers
// Opt-in stage clocks (-DMCG_STAGE_CLOCKS, development builds only; tools/stage_clocks.py): lane 0 of each wave
// accumulates shader-clock deltas per pipeline stage in LDS and adds them to a device-global table at kernel end.
enum { ST_LOAD = 0, ST_CTRL, ST_TRIG, ST_RNE, ST_ACT, ST_CRB, ST_ROWS, ST_G0, ST_NEWTON, ST_EULER, ST_COLLIDE, ST_CUBE,
       ST_COUPLED, ST_CUBE_FIN, ST_POST, ST_N_BUILD, ST_N_FACTOR, ST_N_SOLVE, ST_N_CHECK, ST_E_RHS, ST_R_AX5, ST_R_CONNECT, ST_R_LIMITS, ST_C_MASK, ST_C_ASSEMBLE, ST_C_SCHUR, ST_C_SOLVE, ST_C_CHECK, ST_C_LS, ST_W2_WAIT1, ST_W2_COLLIDE, ST_W2_CUBE, ST_W2_WAIT2, ST_W1_WAIT, ST_A_ENTRY, ST_A_G, ST_A_TWIST, ST_A_LOOP, ST_A_MAP, ST_A_STORE,
       ST_CO_SETUP, ST_CO_ROWS, ST_CO_H0, ST_CO_RESID, ST_CO_ASM, ST_CO_FACTOR, ST_CO_SOLVE, ST_CO_CHECK, ST_CO_LS, ST_CO_OUT, ST_CO_IDLE,
       ST_X_S1C, ST_X_NUMBERS, ST_X_S2, ST_S_S1B, ST_S_MESH, ST_S_S1C, ST_S_NUMBERS, ST_COUNT,
       CN_SUBSTEP = 0, CN_NEWTON_IT, CN_LINESEARCH, CN_CUBE_IT, CN_CUBE_LS, CN_COUPLED, CN_COUPLED_IT, CN_COUPLED_LS, CN_CONTACTS, CN_COOP_ROWS, CN_COOP_LSEVAL, CN_COOP_LONG, CN_COOP_CAP, CN_COOP_12, CN_G_FAILED, CN_G_LIM, CN_G_STAT, CN_G_CUBE, CN_G_MISSING, CN_G_EXTRA, CN_G_EQ1, CN_G_EQ2, CN_G_EQ2N1, CN_MP_PAIRS, CN_MP_HITS, CN_MP_FACE, CN_MP_EXIT_B, CN_MP_EXIT_P, CN_MP_EXIT_E, CN_MP_KIND_E, CN_MP_KIND_P, CN_MP_CUBE, CN_COUNT };
#ifdef MCG_STAGE_CLOCKS
__device__ unsigned long long g_stage_clocks[ST_COUNT + CN_COUNT];      // stage clocks, then event counts (summed over waves)
__device__ unsigned long long g_tick_last[4096 * 4];
__device__ unsigned long long g_wg_stat[4096 * 4];                      // per workgroup: kernel clocks, cooperative solves, cube Newton iterations, cube line searches (robot / cube wave)                    // last time stamp of (workgroup, wave): no LDS is used,
                                                                         // the PickAndPlace kernels need all 160 KB of it
#define MCG_TICK_SLOT_ (g_tick_last[(blockIdx.x & 4095) * 4 + (threadIdx.x >> 6)])
#define MCG_TICK_INIT() do { if ((threadIdx.x & 63) == 0) MCG_TICK_SLOT_ = __builtin_readcyclecounter(); } while (0)
#define MCG_TICK(k) do { if ((threadIdx.x & 63) == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_stage_clocks[k], t_ - MCG_TICK_SLOT_); MCG_TICK_SLOT_ = t_; } } while (0)
#define MCG_TICK2_INIT() MCG_TICK_INIT()
#define MCG_TICK2(k) MCG_TICK(k)
#define MCG_TICK_FLUSH() do {} while (0)
#define MCG_COUNT(k) do { if (threadIdx.x == 0) atomicAdd(&g_stage_clocks[ST_COUNT + (k)], 1ull); } while (0)
#define MCG_COUNTW(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_stage_clocks[ST_COUNT + (k)], (unsigned long long)(v)); } while (0)      // per wave
// a stage's results must exist before its tick: arithmetic is otherwise sunk past the clock read towards its uses
#define MCG_TICK_PIN(arr, n) do { for (int k_ = 0; k_ < (n); k_++) asm volatile("" : "+v"((arr)[k_])); } while (0)
#else
#define MCG_TICK_INIT() do {} while (0)
#define MCG_TICK(k) do {} while (0)
#define MCG_TICK2_INIT() do {} while (0)
#define MCG_TICK2(k) do {} while (0)
#define MCG_TICK_FLUSH() do {} while (0)
#define MCG_COUNT(k) do {} while (0)
#define MCG_COUNTW(k, v) do {} while (0)
#define MCG_TICK_PIN(arr, n) do {} while (0)
#endif
template <int... Is, class F>
MCG_DEV void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
MCG_DEV void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

MCG_DEV void cross(const real* a, const real* b, real* r) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
MCG_DEV void cross_add(const real* a, const real* b, real* r) {   // r += a x b
  r[0] += a[1] * b[2] - a[2] * b[1]; r[1] += a[2] * b[0] - a[0] * b[2]; r[2] += a[0] * b[1] - a[1] * b[0];
}
MCG_DEV real dot3(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// out = Rot(e_K, theta) v   (child -> parent coordinates), c = cos theta, s = sin theta
template <int K>
MCG_DEV void rot_up(real c, real s, const real* v, real* o) {
  constexpr int A = (K + 1) % 3, B = (K + 2) % 3;
  real a = v[A], b = v[B];
  o[K] = v[K]; o[A] = c * a - s * b; o[B] = s * a + c * b;
}
// out = Rot(e_K, theta)^T v (parent -> child coordinates)
template <int K>
MCG_DEV void rot_down(real c, real s, const real* v, real* o) {
  constexpr int A = (K + 1) % 3, B = (K + 2) % 3;
  real a = v[A], b = v[B];
  o[K] = v[K]; o[A] = c * a + s * b; o[B] = -s * a + c * b;
}
// symmetric 3x3 (xx yy zz xy xz yz) times vector
MCG_DEV void sym_mul(const real* I, const real* v, real* o) {
  o[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2];
  o[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2];
  o[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2];
}
// I' = R I R^T for R = Rot(e_K, theta): a plane rotation of the symmetric tensor
template <int K>
MCG_DEV void sym_rot_up(real c, real s, real* I) {
  constexpr int A = (K + 1) % 3, B = (K + 2) % 3;
  // index of the off-diagonal entry (i,j) in (xy=3, xz=4, yz=5)
  constexpr int iAB = (A + B == 1) ? 3 : (A + B == 2) ? 4 : 5;
  constexpr int iKA = (K + A == 1) ? 3 : (K + A == 2) ? 4 : 5;
  constexpr int iKB = (K + B == 1) ? 3 : (K + B == 2) ? 4 : 5;
  real aa = I[A], bb = I[B], ab = I[iAB], ka = I[iKA], kb = I[iKB];
  real cc = c * c, ss = s * s, cs = c * s;
  I[A] = cc * aa - 2 * cs * ab + ss * bb;
  I[B] = ss * aa + 2 * cs * ab + cc * bb;
  I[iAB] = cs * (aa - bb) + (cc - ss) * ab;
  I[iKA] = c * ka - s * kb;
  I[iKB] = s * ka + c * kb;
}

// Uniform-pointer laundering: hides the model pointer from loop-invariant code motion so that the (wave-uniform,
// scalar) loads of model constants stay next to their uses instead of being hoisted out of the sub-step loop and
// spilled -- with one wave per SIMD every spilled SGPR costs two issue slots (v_writelane / v_readlane).
// The pointer is carried in the constant address space (4) so that these reads are emitted as s_load.
typedef const __attribute__((address_space(4))) mcg_model* ModelPtr;
typedef const __attribute__((address_space(4))) real* CRealPtr;
MCG_DEV ModelPtr as_model_ptr(const mcg_model* p) { return (ModelPtr)p; }
MCG_DEV ModelPtr launder(ModelPtr p) { asm volatile("" : "+s"(p)); return p; }
// Scheduling fence between body blocks / stages: without it the machine scheduler clusters the scalar loads of a
// whole unrolled pass at the top of the (several-thousand-instruction) block and spills hundreds of SGPRs.
#define MCG_FENCE() __builtin_amdgcn_sched_barrier(0)
// ... and instruction selection would still sink a block's arithmetic below the following blocks' loads.  Pinning a
// block's results through an empty volatile asm orders "everything they depend on" before the next block's loads.
MCG_DEV void pin(real& a) { asm volatile("" : "+v"(a)); }
MCG_DEV void pin3(real* v) { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])); }
MCG_DEV void pin6(real* v) { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])); }
template <int N> MCG_DEV void ldc(CRealPtr src, real* dst) { for (int k = 0; k < N; k++) dst[k] = src[k]; }
struct BodyC { real r[3], mass, mc[3], inertia[6], armature, damping; };
MCG_DEV BodyC load_body(ModelPtr P, int i) {
  BodyC b;
  ldc<3>(P->body[i].r, b.r); b.mass = P->body[i].mass; ldc<3>(P->body[i].mc, b.mc); ldc<6>(P->body[i].inertia, b.inertia);
  b.armature = P->body[i].armature; b.damping = P->body[i].damping;
  return b;
}

// 1/d to full double precision: v_rcp_f64 seed + two Newton steps (5 issue slots instead of the ~12 of an IEEE fdiv)
MCG_DEV real rcp_nr(real d) {
  real y = __builtin_amdgcn_rcp(d);
  real e = fma(-d, y, 1.0); y = fma(y, e, y);
  e = fma(-d, y, 1.0); y = fma(y, e, y);
  return y;
}

// sin and cos for |x| up to a few turns (joint angles): Cody-Waite reduction by pi/2 in two pieces and the fdlibm
// kernel polynomials on [-pi/4, pi/4]; branch-free, ~30 issue slots, < 1 ulp.
struct TrigC { real t[16]; };
__constant__ const real kTrig[16] = {
    6.36619772367581382433e-01, 1.57079632673412561417e+00, 6.07710050650619224932e-11,          // 2/pi, pio2_1, pio2_1t
    1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06,         // S6 .. S1
    -1.98412698298579493134e-04, 8.33333333332248946124e-03, -1.66666666666666324348e-01,
    -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07,        // C6 .. C1
    2.48015872894767294178e-05, -1.38888888888741095749e-03, 4.16666666666666019037e-02, 0.0};
MCG_DEV TrigC load_trig() {
  TrigC T; CRealPtr p = (CRealPtr)kTrig; asm volatile("" : "+s"(p));
  for (int k = 0; k < 16; k++) T.t[k] = p[k];
  return T;
}
MCG_DEV void sincos_cw(const TrigC& T, real x, real& s, real& c) {
  const real k = rint(x * T.t[0]);
  real r = fma(-k, T.t[1], x);
  r = fma(-k, T.t[2], r);
  const real z = r * r;
  real ps = fma(z, T.t[3], T.t[4]);
  ps = fma(z, ps, T.t[5]); ps = fma(z, ps, T.t[6]);
  ps = fma(z, ps, T.t[7]); ps = fma(z, ps, T.t[8]);
  const real sr = fma(r * z, ps, r);
  real pc = fma(z, T.t[9], T.t[10]);
  pc = fma(z, pc, T.t[11]); pc = fma(z, pc, T.t[12]);
  pc = fma(z, pc, T.t[13]); pc = fma(z, pc, T.t[14]);
  const real cr = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = (int)k;
  const real s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}

// Impedance sigmoid (MuJoCo getimpedance [RECALL]); par = K B d0 dmax width midpoint power 1/width a b
MCG_DEV real impedance(const real* par, real dist) {
  const real d0 = par[2], dmax = par[3], width = par[4], mid = par[5], power = par[6];
  real imp;
  if (d0 == dmax || width <= MINVAL) imp = 0.5 * (d0 + dmax);           // wave-uniform branch
  else {
    const real x = fabs(dist) * par[7];
    real y;
    if (power == 2) {                                                   // wave-uniform; the model's default
      const real u = 1 - x;
      y = (x <= mid) ? par[8] * (x * x) : 1 - par[9] * (u * u);
    } else y = x;       // power == 1; any other power is rejected by mcg_create (keeps pow() out of the hot loop)
    imp = d0 + y * (dmax - d0);
    imp = (x >= 1) ? dmax : imp;
    imp = (x == 0) ? d0 : imp;
  }
  return fmin(fmax(imp, MINIMP), MAXIMP);
}

constexpr int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// ---------------------------------------------------------------------------- sparse L^T D L with fixed patterns
// The joint-space matrices have compile-time sparsity: M follows the kinematic tree (no fill-in when eliminated
// leaves-first, Featherstone); H = M + J^T D J adds the two connect cliques {gear, finger, hinge} and the gear-gear
// coupling.  The symbolic factorisation runs in constexpr; the numeric code below touches only structural non-zeros.
struct Pattern { bool nz[NB][NB]; };

constexpr bool is_ancestor_or_self(int a, int i) {       // a on the path from i to the root
  while (i >= 0) { if (i == a) return true; i = PAR[i]; }
  return false;
}
constexpr Pattern symbolic(bool with_constraints, bool fill = true) {
  Pattern P{};
  for (int i = 0; i < NB; i++) for (int j = 0; j <= i; j++) P.nz[i][j] = is_ancestor_or_self(j, i);
  if (with_constraints) {
    constexpr int grp[2][3] = {{6, 7, 10}, {8, 9, 11}};
    for (int g = 0; g < 2; g++) for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
      int i = grp[g][a], j = grp[g][b];
      if (i >= j) P.nz[i][j] = true;
    }
    for (int g = 0; g < 2; g++) for (int a = 0; a < 3; a++) for (int j = 0; j < 6; j++) P.nz[grp[g][a]][j] = true;
    P.nz[8][6] = true;
  }
  if (!fill) return P;
  for (int k = NB - 1; k >= 0; k--)                      // fill-in of the leaves-first elimination
    for (int i = 0; i < k; i++) if (P.nz[k][i])
      for (int j = 0; j <= i; j++) if (P.nz[k][j]) P.nz[i][j] = true;
  return P;
}
constexpr Pattern symbolic_grasp() {          // H plus the cross terms a cube pinched by both pads induces (Schur complement)
  Pattern P = symbolic(true);
  constexpr int gr[4] = {6, 7, 8, 9};
  for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) if (gr[a] >= gr[b]) P.nz[gr[a]][gr[b]] = true;
  for (int k = NB - 1; k >= 0; k--)
    for (int i = 0; i < k; i++) if (P.nz[k][i])
      for (int j = 0; j <= i; j++) if (P.nz[k][j]) P.nz[i][j] = true;
  return P;
}
constexpr Pattern PAT_M = symbolic(false);
constexpr Pattern PAT_H = symbolic(true);
constexpr Pattern PAT_E = symbolic(true, false);          // M + J^T D J over the equality rows, before fill-in
constexpr Pattern PAT_G = symbolic_grasp();

// in place on the packed lower triangle: A = L^T D L, L unit lower (stored below the diagonal), dinv = 1 / D
template <const Pattern& PT>
MCG_DEV void ldl_factor(real* A, real* dinv) {
  static_for<NB>([&](auto Kk) {
    constexpr int k = NB - 1 - Kk;
    dinv[k] = rcp_nr(A[tri(k, k)]);
    static_for<k>([&](auto Ii) {
      constexpr int i = k - 1 - Ii;
      if constexpr (PT.nz[k][i]) {
        const real l = A[tri(k, i)] * dinv[k];
        static_for<i + 1>([&](auto Jj) {
          constexpr int j = Jj;
          if constexpr (PT.nz[k][j]) A[tri(i, j)] = fma(-l, A[tri(k, j)], A[tri(i, j)]);
        });
        A[tri(k, i)] = l;
      }
    });
  });
}
template <const Pattern& PT>
MCG_DEV void ldl_solve(const real* A, const real* dinv, real* x) {
  static_for<NB>([&](auto Kk) {
    constexpr int k = NB - 1 - Kk;
    static_for<k>([&](auto Ii) { constexpr int i = Ii; if constexpr (PT.nz[k][i]) x[i] = fma(-A[tri(k, i)], x[k], x[i]); });
  });
  static_for<NB>([&](auto Kk) { constexpr int k = Kk; x[k] *= dinv[k]; });
  static_for<NB>([&](auto Kk) {
    constexpr int k = Kk;
    static_for<k>([&](auto Ii) { constexpr int i = Ii; if constexpr (PT.nz[k][i]) x[k] = fma(-A[tri(k, i)], x[i], x[k]); });
  });
}

// dense SPD solve by L D L^T with reciprocals (no square roots, no IEEE divisions): the cube's 6x6 Newton systems
template <int N>
MCG_DEV void spd_factor(real* A, real* dinv) {           // packed lower triangle, in place: L below the diagonal, D on it
  static_for<N>([&](auto J) {
    constexpr int j = J;
    real v[N];
    real d = A[tri(j, j)];
    static_for<j>([&](auto Kk) { constexpr int k = Kk; v[k] = A[tri(j, k)] * A[tri(k, k)]; d = fma(-A[tri(j, k)], v[k], d); });
    A[tri(j, j)] = d;
    dinv[j] = rcp_nr(d);
    static_for<N - 1 - j>([&](auto Ii) {
      constexpr int i = j + 1 + Ii;
      real sacc = A[tri(i, j)];
      static_for<j>([&](auto Kk) { constexpr int k = Kk; sacc = fma(-A[tri(i, k)], v[k], sacc); });
      A[tri(i, j)] = sacc * dinv[j];
    });
  });
}
template <int N>
MCG_DEV void spd_forward(const real* A, real* x) {            // x <- L^-1 x  (unit lower L below the diagonal of A)
  static_for<N>([&](auto I) { constexpr int i = I; static_for<i>([&](auto Kk) { constexpr int k = Kk; x[i] = fma(-A[tri(i, k)], x[k], x[i]); }); });
}
template <int N>
MCG_DEV void spd_backward(const real* A, real* x) {           // x <- L^-T x
  static_for<N>([&](auto I) { constexpr int i = N - 1 - I;
    static_for<N - 1 - i>([&](auto Kk) { constexpr int k = i + 1 + Kk; x[i] = fma(-A[tri(k, i)], x[k], x[i]); }); });
}
template <int N>
MCG_DEV void spd_solve(const real* A, const real* dinv, real* x) {
  static_for<N>([&](auto I) { constexpr int i = I; static_for<i>([&](auto Kk) { constexpr int k = Kk; x[i] = fma(-A[tri(i, k)], x[k], x[i]); }); });
  static_for<N>([&](auto I) { constexpr int i = I; x[i] *= dinv[i]; });
  static_for<N>([&](auto I) { constexpr int i = N - 1 - I;
    static_for<N - 1 - i>([&](auto Kk) { constexpr int k = i + 1 + Kk; x[i] = fma(-A[tri(k, i)], x[k], x[i]); }); });
}

// dense SPD solve for the 6x6 IK system (once per control step; not on the sub-step path)
template <int N>
MCG_DEV void chol_factor(real* A) {
  static_for<N>([&](auto I) {
    constexpr int i = I;
    static_for<i + 1>([&](auto J) {
      constexpr int j = J;
      real s = A[i * (i + 1) / 2 + j];
      static_for<j>([&](auto Kk) { constexpr int k = Kk; s -= A[i * (i + 1) / 2 + k] * A[j * (j + 1) / 2 + k]; });
      if constexpr (i == j) A[i * (i + 1) / 2 + i] = sqrt(s);
      else A[i * (i + 1) / 2 + j] = s / A[j * (j + 1) / 2 + j];
    });
  });
}
template <int N>
MCG_DEV void chol_solve(const real* L, real* x) {
  static_for<N>([&](auto I) {
    constexpr int i = I;
    real s = x[i];
    static_for<i>([&](auto Kk) { constexpr int k = Kk; s -= L[i * (i + 1) / 2 + k] * x[k]; });
    x[i] = s / L[i * (i + 1) / 2 + i];
  });
  static_for<N>([&](auto I) {
    constexpr int i = N - 1 - I;
    real s = x[i];
    static_for<N - 1 - i>([&](auto Kk) { constexpr int k = i + 1 + Kk; s -= L[k * (k + 1) / 2 + i] * x[k]; });
    x[i] = s / L[i * (i + 1) / 2 + i];
  });
}

// ------------------------------------------------------------------------------------- robot sub-step state
struct Robot {
  real q[NB], qd[NB], ctrl[7], warm[NB];
};

// Per-lane scratch in LDS: slot k of this lane lives at base[k * 64] (lane-contiguous rows: conflict-free
// ds_read_b64 / ds_write_b64 with immediate offsets).  The joint-space inertia M is kept here between the stages
// that consume it, so that only one 12x12 system occupies registers at a time.
constexpr int LDS_M = 0;                              // packed lower triangle of M                     (78 slots)
constexpr int LDS_HEQ = NB * (NB + 1) / 2;            // H_eq = M + J^T D J over the equality (and weld) rows  (78 slots)
constexpr int LDS_SLOTS = 2 * (NB * (NB + 1) / 2);
// two-wave variant (SplitA / helper_substep): factor of M + hB, its reciprocal pivots, and q published for the helper wave
constexpr int LDS_FAC = LDS_SLOTS, LDS_FDINV = LDS_FAC + NB * (NB + 1) / 2, LDS_QB = LDS_FDINV + NB, LDS_QDB = LDS_QB + NB;
constexpr int LDS_FS = LDS_QDB + NB, LDS_WARM = LDS_FS + NB, LDS_QLAG = LDS_WARM + NB, LDS_IKT = LDS_QLAG + 6, LDS_SLOTS_SPLIT = LDS_IKT + 8;      // LDS_WARM: qacc_warmstart parked between sub-steps; LDS_QLAG: q of the last forward pass; LDS_IKT: the IK controller's six ctrl increments of a control step, handed from the RNE wave to the main wave (two slots spare)
// The lane's LDS column.  The pointer carries the LDS address space explicitly: passed through structs as a generic
// pointer the accesses degrade to flat_load/flat_store with 64-bit address arithmetic instead of ds_read/ds_write
// with immediate offsets.
// Value selects.  Written `c ? a[k] : b[k]` inside a loop, the two arm loads are merged into ONE load through a selected
// pointer before the loop is unrolled, which pins both arrays in scratch memory; evaluating both operands first (function
// arguments) keeps the select on values and the arrays in registers.
template <class A, class B> MCG_DEV std::common_type_t<A, B> sel(bool c, A a, B b) { return c ? a : b; }
template <class T> MCG_DEV T sel3(int idx, T a, T b, T c) { return idx == 0 ? a : (idx == 1 ? b : c); }

#ifdef MCG_GENERIC_LDS
typedef real* LdsPtr;
#else
typedef __attribute__((address_space(3))) real* LdsPtr;
#endif
template <int STRIDE>
struct LaneScratchT {
  LdsPtr base;
  MCG_DEV explicit LaneScratchT(real* shared_column) : base((LdsPtr)shared_column) {}
  MCG_DEV explicit LaneScratchT(LdsPtr column) : base(column) {}
  MCG_DEV real ld(int k) const { return base[k * STRIDE]; }
  MCG_DEV void st(int k, real v) const { base[k * STRIDE] = v; }
};
typedef LaneScratchT<64> LaneScratch;

// One physics sub-step (mj_step) of the 12-dof robot.  `qlag` receives the positions the forward pass used.
// Coupling hook: PickAndPlace passes an object that, when a finger pad touches the cube, solves the robot and cube
// accelerations together (the Euler step needs no constraint force: M a carries it).  Reach passes NoCoupling (compiled out).
struct NoCoupling { static constexpr bool enabled = false, publishes = false; };
struct NoSideWork { MCG_DEV void operator()(const real*, const real*) const {} };      // what a helper / RNE wave does after its own share
// A hook with `publishes` (and not `enabled`) is handed the Newton system's smooth right-hand side and the limit rows once they are
// complete: the four-wave PickAndPlace kernel parks them in LDS for the cooperative coupled solve (mcg_coop.hpp).

// Mocap weld (mocap controller, mycobot.py:172-189; mocap.xml:15-20): six equality rows pull gripper_tcp to the mocap
// body's pose.  NoWeld compiles the rows out (joint / IK controllers).
struct NoWeld { static constexpr bool enabled = false; };
struct Weld { static constexpr bool enabled = true; real pos[3], quat[4]; };     // mocap pose, quaternion normalised

MCG_DEV void mulquat(const real* a, const real* b, real* r);
MCG_DEV void normalize4(real* q);
MCG_DEV void quat_to_mat(const real* q, real* m) {       // row-major world <- body
  const real q00 = q[0]*q[0], q01 = q[0]*q[1], q02 = q[0]*q[2], q03 = q[0]*q[3];
  const real q11 = q[1]*q[1], q12 = q[1]*q[2], q13 = q[1]*q[3], q22 = q[2]*q[2], q23 = q[2]*q[3], q33 = q[3]*q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2*(q12 - q03); m[2] = 2*(q13 + q02); m[3] = 2*(q12 + q03);
  m[5] = 2*(q23 - q01); m[6] = 2*(q13 - q02); m[7] = 2*(q23 + q01);
}

// World pose of gripper_tcp at arm angles q6: position of the weld point, MuJoCo's xquat (the chain product
// base_quat * prod_i (cos(q_i/2), axis_i sin(q_i/2)), sign included: the weld residual depends on it), Jacobians.
struct TcpPose { real pos[3], quat[4], jacp[3][6], jacr[3][6]; };
MCG_DEV void tcp_forward(ModelPtr P, const real* q6, TcpPose& X, bool want_jac) {
  real R[9], p[3], anchor[6][3], axis[6][3], Q[4];
  const TrigC T = load_trig();
  for (int k = 0; k < 9; k++) R[k] = P->base_mat[k];
  for (int k = 0; k < 3; k++) p[k] = P->base_pos[k];
  for (int k = 0; k < 4; k++) Q[k] = P->base_quat[k];
  static_for<6>([&](auto I) {
    constexpr int i = I; constexpr int K = AXK[i]; constexpr int A = (K + 1) % 3, B = (K + 2) % 3;
    real r[3]; ldc<3>(P->body[i].r, r);
    for (int k = 0; k < 3; k++) p[k] += R[3 * k] * r[0] + R[3 * k + 1] * r[1] + R[3 * k + 2] * r[2];
    for (int k = 0; k < 3; k++) { anchor[i][k] = p[k]; axis[i][k] = AXS[i] * R[3 * k + K]; }
    real s, c; sincos_cw(T, AXS[i] * q6[i], s, c);
    for (int k = 0; k < 3; k++) {
      real ca = R[3 * k + A], cb = R[3 * k + B];
      R[3 * k + A] = c * ca + s * cb; R[3 * k + B] = -s * ca + c * cb;
    }
    real sh, ch; sincos_cw(T, 0.5 * (AXS[i] * q6[i]), sh, ch);
    real ql[4] = {ch, 0, 0, 0}, Qn[4]; ql[1 + K] = sh;
    mulquat(Q, ql, Qn);
    for (int k = 0; k < 4; k++) Q[k] = Qn[k];
  });
  normalize4(Q);
  for (int k = 0; k < 4; k++) X.quat[k] = Q[k];
  real wa[3]; ldc<3>(P->weld_anchor, wa);
  for (int k = 0; k < 3; k++) X.pos[k] = p[k] + R[3 * k] * wa[0] + R[3 * k + 1] * wa[1] + R[3 * k + 2] * wa[2];
  if (want_jac) {
    static_for<6>([&](auto I) {
      constexpr int i = I;
      real d[3] = {X.pos[0] - anchor[i][0], X.pos[1] - anchor[i][1], X.pos[2] - anchor[i][2]}, c3[3];
      cross(axis[i], d, c3);
      for (int k = 0; k < 3; k++) { X.jacp[k][i] = c3[k]; X.jacr[k][i] = axis[i][k]; }
    });
  }
}

// ---- P6 recursive Newton-Euler, q'' = 0: bias = Coriolis + centrifugal + gravity        (mj_rne, flg_acc=0)
// fs <- passive (joint damping) - bias.  Depends on q (through cs / sn), qd and the model only.
MCG_DEV void rne_bias(ModelPtr Pm, const real* cs, const real* sn, const real* qd, real* fs) {
  real F[NB][3], Nn[NB][3];                 // net force / moment about the body origin, body frame
  real w[NB][3], al[NB][3], ac[NB][3];      // angular velocity, angular acceleration, linear acceleration of the origin
  BodyC nxt = load_body(launder(Pm), 0);      // each block starts the s_loads of the next body before it computes (see crb_to_lds)
  static_for<NB>([&](auto I) {
    constexpr int i = I; constexpr int p = PAR[i]; constexpr int K = AXK[i];
    constexpr int A = (K + 1) % 3, B = (K + 2) % 3;
    const BodyC bc = nxt; const BodyC* b = &bc;
    if constexpr (i + 1 < NB) nxt = load_body(launder(Pm), i + 1); else nxt = load_body(launder(Pm), NB - 1);
    const real g = AXS[i] * qd[i];
    if constexpr (p < 0) {
      // static base: w_p = al_p = 0, a_p = -gravity (base frame)
      real gb[3]; ldc<3>(launder(Pm)->gravity_base, gb);
      rot_down<K>(cs[i], sn[i], gb, ac[i]);
      w[i][0] = w[i][1] = w[i][2] = 0; w[i][K] = g;
      al[i][0] = al[i][1] = al[i][2] = 0;
    } else {
      real t[3], accp[3], we[3];
      cross(w[p], b->r, t);
      accp[0] = ac[p][0]; accp[1] = ac[p][1]; accp[2] = ac[p][2];
      cross_add(al[p], b->r, accp); cross_add(w[p], t, accp);
      rot_down<K>(cs[i], sn[i], accp, ac[i]);
      rot_down<K>(cs[i], sn[i], w[p], we);
      rot_down<K>(cs[i], sn[i], al[p], al[i]);
      al[i][A] += we[B] * g; al[i][B] -= we[A] * g;          // + (E w_p) x (g e_K)
      w[i][0] = we[0]; w[i][1] = we[1]; w[i][2] = we[2]; w[i][K] += g;
    }
    real t2[3], Iw[3];
    cross(w[i], b->mc, t2);
    F[i][0] = b->mass * ac[i][0]; F[i][1] = b->mass * ac[i][1]; F[i][2] = b->mass * ac[i][2];
    cross_add(al[i], b->mc, F[i]); cross_add(w[i], t2, F[i]);
    sym_mul(b->inertia, al[i], Nn[i]); sym_mul(b->inertia, w[i], Iw);
    cross_add(w[i], Iw, Nn[i]); cross_add(b->mc, ac[i], Nn[i]);
    pin3(F[i]); pin3(Nn[i]);
    MCG_FENCE();
  });
  static_for<NB>([&](auto I) {
    constexpr int i = NB - 1 - I; constexpr int p = PAR[i]; constexpr int K = AXK[i];
    const BodyC bc = nxt; const BodyC* b = &bc;
    if constexpr (i > 0) nxt = load_body(launder(Pm), i - 1);
    fs[i] = -b->damping * qd[i] - AXS[i] * Nn[i][K];
    if constexpr (p >= 0) {
      real fp[3], np[3];
      rot_up<K>(cs[i], sn[i], F[i], fp); rot_up<K>(cs[i], sn[i], Nn[i], np);
      cross_add(b->r, fp, np);
      for (int k = 0; k < 3; k++) { F[p][k] += fp[k]; Nn[p][k] += np[k]; }
      pin3(F[p]); pin3(Nn[p]);
    }
    pin(fs[i]);
    MCG_FENCE();
  });
}

// ---- P3 composite rigid bodies -> joint-space inertia M (packed lower triangle, in LDS)      (mj_crb)
// Depends on the joint angles (through cs / sn) and the model only.
template <class LS>
MCG_DEV void crb_to_lds(ModelPtr Pm, const real* cs, const real* sn, const LS MS) {
    real cm[NB], cmc[NB][3], cI[NB][6];
    // Scalar loads, batched: the offsets of the six arm bodies (51 of the 57 walk levels below pass through them) come in one batch
    // up front, and each body block starts the loads of the NEXT body's constants before it computes -- a wave alone on its SIMD
    // otherwise stalls ~100 clocks on every one of ~70 dependent s_load round trips, and this wave is on the critical path.
    real rarm[6][3];
    { ModelPtr Q = launder(Pm); static_for<6>([&](auto I) { constexpr int i = I; ldc<3>(Q->body[i].r, rarm[i]); }); }
    BodyC nxt = load_body(launder(Pm), NB - 1);
    static_for<NB>([&](auto I) {
      constexpr int i = NB - 1 - I; constexpr int K = AXK[i]; constexpr int A = (K + 1) % 3, B = (K + 2) % 3;
      constexpr int p = PAR[i];
      constexpr bool leaf = (i == 7 || i == 9 || i == 10 || i == 11);
      const BodyC bc = nxt; const BodyC* b = &bc;
      if constexpr (i > 0) nxt = load_body(launder(Pm), i - 1);
      if constexpr (leaf) {
        cm[i] = b->mass;
        for (int k = 0; k < 3; k++) cmc[i][k] = b->mc[k];
        for (int k = 0; k < 6; k++) cI[i][k] = b->inertia[k];
      } else {          // children (higher index) have already added their composites
        cm[i] += b->mass;
        for (int k = 0; k < 3; k++) cmc[i][k] += b->mc[k];
        for (int k = 0; k < 6; k++) cI[i][k] += b->inertia[k];
      }
      // wrench of a unit acceleration about joint i acting on composite body i (frame i, about origin i)
      constexpr int iKA = (K + A == 1) ? 3 : (K + A == 2) ? 4 : 5;
      constexpr int iKB = (K + B == 1) ? 3 : (K + B == 2) ? 4 : 5;
      const real sg = AXS[i];
      real fj[3], nj[3];
      nj[K] = sg * cI[i][K]; nj[A] = sg * cI[i][iKA]; nj[B] = sg * cI[i][iKB];
      fj[K] = 0; fj[A] = -sg * cmc[i][B]; fj[B] = sg * cmc[i][A];          // (sg e_K) x mc
      MS.st(tri(i, i), cI[i][K] + b->armature);
      // walk to the root: M[i][j] = axis_j . moment about origin j
      auto up = [&](auto self, auto Cur) -> void {
        constexpr int cur = Cur; constexpr int pj = PAR[cur];
        if constexpr (pj >= 0) {
          constexpr int Kc = AXK[cur];
          real rc[3];
          if constexpr (cur < 6) { rc[0] = rarm[cur][0]; rc[1] = rarm[cur][1]; rc[2] = rarm[cur][2]; }
          else if constexpr (cur == i) { rc[0] = b->r[0]; rc[1] = b->r[1]; rc[2] = b->r[2]; }      // this block's own constants
          else ldc<3>(launder(Pm)->body[cur].r, rc);
          real f2[3], n2[3];
          rot_up<Kc>(cs[cur], sn[cur], fj, f2);
          rot_up<Kc>(cs[cur], sn[cur], nj, n2);
          cross_add(rc, f2, n2);
          for (int k = 0; k < 3; k++) { fj[k] = f2[k]; nj[k] = n2[k]; }
          MS.st(tri(i, pj), AXS[pj] * nj[AXK[pj]]);
          self(self, std::integral_constant<int, pj>{});
        }
      };
      up(up, std::integral_constant<int, i>{});
      // add composite i to its parent
      if constexpr (p >= 0) {
        constexpr bool first_child = (i == 11 || i == 7 || i == 9 || i < 6);   // highest-index child of its parent
        real It[6], h3[3];
        for (int k = 0; k < 6; k++) It[k] = cI[i][k];
        sym_rot_up<K>(cs[i], sn[i], It);
        rot_up<K>(cs[i], sn[i], cmc[i], h3);
        const real* r = b->r; const real m = cm[i];
        const real rr = dot3(r, r), rh = dot3(r, h3);
        const real d = m * rr + 2 * rh;
        real add[6];
        add[0] = It[0] + d - (m * r[0] * r[0] + 2 * r[0] * h3[0]);
        add[1] = It[1] + d - (m * r[1] * r[1] + 2 * r[1] * h3[1]);
        add[2] = It[2] + d - (m * r[2] * r[2] + 2 * r[2] * h3[2]);
        add[3] = It[3] - (m * r[0] * r[1] + r[0] * h3[1] + h3[0] * r[1]);
        add[4] = It[4] - (m * r[0] * r[2] + r[0] * h3[2] + h3[0] * r[2]);
        add[5] = It[5] - (m * r[1] * r[2] + r[1] * h3[2] + h3[1] * r[2]);
        if constexpr (first_child) {
          cm[p] = m;
          for (int k = 0; k < 3; k++) cmc[p][k] = h3[k] + m * r[k];
          for (int k = 0; k < 6; k++) cI[p][k] = add[k];
        } else {
          cm[p] += m;
          for (int k = 0; k < 3; k++) cmc[p][k] += h3[k] + m * r[k];
          for (int k = 0; k < 6; k++) cI[p][k] += add[k];
        }
        pin(cm[p]); pin3(cmc[p]); pin6(cI[p]);
      }
      asm volatile("" ::: "memory");      // the M entries of this body are in LDS before the next block starts
      MCG_FENCE();
    });
}

// P10, velocity part: at the minimiser the gradient vanishes, M a = qfrc_smooth + J^T f over ALL rows (equality, limit, weld,
// contact), so the right-hand side of (M + hB) a' = qfrc_smooth + qfrc_constraint is M a: no Jacobian is live after the solve.
// rhs <- (M + hB)^-1 M a from the LDS-resident M.
template <class LS>
MCG_DEV void euler_accel(ModelPtr Pm, real h, const LS MS, const real* a, real* rhs) {
  real Mh[NB * (NB + 1) / 2], dinv[NB];
  static_for<NB>([&](auto I) { constexpr int i = I;
    static_for<i + 1>([&](auto Jj) { constexpr int j = Jj; if constexpr (PAT_M.nz[i][j]) Mh[tri(i, j)] = MS.ld(LDS_M + tri(i, j)); }); });
  static_for<NB>([&](auto I) { constexpr int i = I; real sacc = 0;
    static_for<NB>([&](auto Jj) { constexpr int j = Jj;
      if constexpr (PAT_M.nz[i > j ? i : j][i > j ? j : i]) sacc = fma(Mh[tri(i, j)], a[j], sacc); }); rhs[i] = sacc; });
  { ModelPtr Q = launder(Pm); static_for<NB>([&](auto I) { constexpr int i = I; Mh[tri(i, i)] = fma(h, Q->body[i].damping, Mh[tri(i, i)]); }); }
  ldl_factor<PAT_M>(Mh, dinv);
  ldl_solve<PAT_M>(Mh, dinv, rhs);
}

// Three-wave variant (Reach, grids of at most one workgroup per CU, where 3 of the 4 SIMDs of a CU would idle): the workgroup
// has two more waves over the same 64 environments.  The HELPER wave computes what depends on the joint angles alone -- M by
// the composite rigid body pass, then the L^T D L factor of M + hB for the Euler step; the RNE wave computes the bias forces;
// the main wave does actuation and constraint rows meanwhile, then H_eq, the Newton solve and the Euler step.  Three workgroup
// barriers per sub-step:
//   S1  q(t), qd(t) are published in LDS      (the other waves may read them)
//   S2  M(t) and passive - bias are in LDS    (main wave: g0, H_eq, Newton solve)
//   S3  the factor of M + hB is in LDS        (main wave: a' = a - h (M + hB)^-1 (B a), which equals (M + hB)^-1 M a)
// A split policy says which pieces other waves provide and where the exchange slots are.
// early_heq: the J^T D J part of H_eq is assembled (and the constraint part of g0 formed) BEFORE barrier S2, while the main wave
// would otherwise wait for M; build_H then adds M on the fly.
// warm_lds: qacc_warmstart lives in LDS slots WARM.. between sub-steps instead of in registers.  The closed-form limit solve does not
// read it; only the rare general iteration does -- but in registers it stays live through the factorisation, the kernel's register peak.
// With it the Euler step also re-reads q(t), qd(t) from the slots they were published in for the other waves (QB, QDB) instead of
// carrying them through the solve, and the lagged configuration (q of this forward pass, for observations and IK) goes to slots QLAG.
struct NoSplit { static constexpr bool enabled = false, rne_remote = false, factor_remote = false, early_heq = false, warm_lds = false, mesh_split = false;
                 static constexpr int QB = 0, QDB = 0, FS = 0, WARM = 0, QLAG = 0; };
struct SplitMain { static constexpr bool enabled = true, rne_remote = true, factor_remote = true, early_heq = true, warm_lds = true, mesh_split = false;
                   static constexpr int QB = LDS_QB, QDB = LDS_QDB, FS = LDS_FS, WARM = LDS_WARM, QLAG = LDS_QLAG; };

// COMMIT = false: the new q / qd / qacc_warmstart go to *next and S stays as it was (speculative sub-step of the two-wave
// PickAndPlace kernel: discarded when the helper wave's collision pass finds a pad contact).
// mj_checkPos / mj_checkVel / mj_checkAcc [RECALL]: a coordinate that is not finite or beyond 1e10 makes mj_step call mj_resetData
MCG_DEV bool bad_value(real x) { return !(fabs(x) <= 1e10); }

// Returns true in the lanes whose new state failed MuJoCo's checks and was reset (COMMIT only; the speculative caller checks *next).
template <class LS, class CPL = NoCoupling, class WLD = NoWeld, class SPL = NoSplit, bool COMMIT = true>
MCG_DEV bool robot_substep(ModelPtr Pm, Robot& S, real* qlag6, const LS MS, CPL* CP = nullptr, const WLD* WD = nullptr,
                           Robot* next = nullptr) {
  MCG_COUNT(CN_SUBSTEP);
  if constexpr (SPL::enabled) __syncthreads();                     // S1
  const real h = launder(Pm)->timestep;
  real cs[NB], sn[NB];
  {
    const TrigC T = load_trig();
    static_for<NB>([&](auto I) { constexpr int i = I; sincos_cw(T, AXS[i] * S.q[i], sn[i], cs[i]); });
  }
  if constexpr (!(SPL::warm_lds && !CPL::enabled)) static_for<6>([&](auto I) { constexpr int i = I; qlag6[i] = S.q[i]; });      // (else: to LDS at the end of the sub-step)
  static_for<NB>([&](auto I) { constexpr int i = I; pin(sn[i]); pin(cs[i]); });
  MCG_FENCE();

  MCG_TICK(ST_TRIG);
  real fs[NB];        // becomes qfrc_smooth = passive - bias + actuation
  if constexpr (SPL::rne_remote) { static_for<NB>([&](auto I) { constexpr int i = I; fs[i] = 0; }); }   // a third wave computes it meanwhile
  else rne_bias(Pm, cs, sn, S.qd, fs);
  MCG_TICK(ST_RNE);
  // ---- P7 actuation                                                             (mj_fwdActuation)
  {
    ModelPtr Q = launder(Pm);
    static_for<6>([&](auto I) {
      constexpr int u = I;
      real c = fmin(fmax(S.ctrl[u], Q->act_ctrlrange[u][0]), Q->act_ctrlrange[u][1]);
      real f = Q->act_gain[u] * c + Q->act_bias[u][0] + Q->act_bias[u][1] * S.q[u] + Q->act_bias[u][2] * S.qd[u];
      fs[u] += fmin(fmax(f, Q->act_forcerange[u][0]), Q->act_forcerange[u][1]);
    });
    const real c0 = Q->tendon_coef[0], c1 = Q->tendon_coef[1];
    real len = c0 * S.q[6] + c1 * S.q[8], vel = c0 * S.qd[6] + c1 * S.qd[8];
    real c = fmin(fmax(S.ctrl[6], Q->act_ctrlrange[6][0]), Q->act_ctrlrange[6][1]);
    real f = Q->act_gain[6] * c + Q->act_bias[6][0] + Q->act_bias[6][1] * len + Q->act_bias[6][2] * vel;
    f = fmin(fmax(f, Q->act_forcerange[6][0]), Q->act_forcerange[6][1]);
    fs[6] += c0 * f; fs[8] += c1 * f;
  }
  static_for<NB>([&](auto I) { constexpr int i = I; pin(fs[i]); });
  MCG_FENCE();

  MCG_TICK(ST_ACT);
  if constexpr (!SPL::enabled) crb_to_lds(Pm, cs, sn, MS);        // two-wave variant: the helper wave does it meanwhile
  MCG_TICK(ST_CRB);
  // ---- P5 constraint rows                                                   (mj_makeConstraint)
  // arm joint axes expressed in the link6 frame (for the tiny arm columns of the connect rows)
  real ax5[6][3];
  static_for<6>([&](auto I) {
    constexpr int i = I;
    for (int k = 0; k < 3; k++) ax5[i][k] = 0;
    ax5[i][AXK[i]] = AXS[i];
    static_for<5 - i>([&](auto Jj) {
      constexpr int j = i + 1 + Jj;
      real t[3]; rot_down<AXK[j]>(cs[j], sn[j], ax5[i], t);
      for (int k = 0; k < 3; k++) ax5[i][k] = t[k];
    });
  });
  MCG_TICK_PIN(ax5[0], 3); MCG_TICK_PIN(ax5[1], 3); MCG_TICK_PIN(ax5[2], 3); MCG_TICK_PIN(ax5[3], 3); MCG_TICK_PIN(ax5[4], 3);
  MCG_TICK(ST_R_AX5);
  // two connects; side 0: gear 6 / finger 7 / hinge 10, side 1: gear 8 / finger 9 / hinge 11.
  // Jc[sd][row][col], cols = arm 0..5, gear, finger, hinge; the y row has no gripper entries (planar mechanism).
  real Jc[2][3][9];
  real Dc[2], arefc[2][3];
  static_for<2>([&](auto Sd) {
    constexpr int sd = Sd; constexpr int g = 6 + 2 * sd, fi = 7 + 2 * sd, hg = 10 + sd;
    ModelPtr Q = launder(Pm);
    real rg[3], rh[3], rf[3], an1[3], an2[3], par[10];
    ldc<3>(Q->body[g].r, rg); ldc<3>(Q->body[hg].r, rh); ldc<3>(Q->body[fi].r, rf);
    ldc<3>(Q->eq_anchor1[sd], an1); ldc<3>(Q->eq_anchor2[sd], an2); ldc<10>(Q->eq_par[sd], par);
    real t[3], t2[3], o_f[3], p1[3], p2[3];
    rot_up<1>(cs[g], sn[g], rf, t);
    for (int k = 0; k < 3; k++) o_f[k] = rg[k] + t[k];
    rot_up<1>(cs[fi], sn[fi], an1, t2); rot_up<1>(cs[g], sn[g], t2, t);
    for (int k = 0; k < 3; k++) p1[k] = o_f[k] + t[k];
    rot_up<1>(cs[hg], sn[hg], an2, t);
    for (int k = 0; k < 3; k++) p2[k] = rh[k] + t[k];
    const real pos[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    static_for<6>([&](auto I) { constexpr int i = I; real c3[3]; cross(ax5[i], pos, c3); for (int k = 0; k < 3; k++) Jc[sd][k][i] = c3[k]; });
    // (s e_y) x d = s (d_z, 0, -d_x)
    Jc[sd][0][6] = AXS[g] * (p1[2] - rg[2]);   Jc[sd][2][6] = -AXS[g] * (p1[0] - rg[0]);
    Jc[sd][0][7] = AXS[fi] * (p1[2] - o_f[2]); Jc[sd][2][7] = -AXS[fi] * (p1[0] - o_f[0]);
    Jc[sd][0][8] = -AXS[hg] * (p2[2] - rh[2]); Jc[sd][2][8] = AXS[hg] * (p2[0] - rh[0]);
    Jc[sd][1][6] = Jc[sd][1][7] = Jc[sd][1][8] = 0;
    const real imp = impedance(par, sqrt(dot3(pos, pos)));
    Dc[sd] = imp * rcp_nr(fmax(MINVAL * imp, (1 - imp) * Q->eq_diag[sd]));      // 1 / max(MINVAL, (1-imp) diag / imp)
    static_for<3>([&](auto Kk) {
      constexpr int k = Kk;
      real vel = 0;
      static_for<6>([&](auto I) { constexpr int i = I; vel = fma(Jc[sd][k][i], S.qd[i], vel); });
      if constexpr (k != 1) vel += Jc[sd][k][6] * S.qd[g] + Jc[sd][k][7] * S.qd[fi] + Jc[sd][k][8] * S.qd[hg];
      arefc[sd][k] = -par[1] * vel - par[0] * imp * pos[k];
    });
    pin(Dc[sd]); pin3(arefc[sd]);
    for (int k = 0; k < 3; k++) { pin3(&Jc[sd][k][0]); pin3(&Jc[sd][k][3]); pin3(&Jc[sd][k][6]); }
    MCG_FENCE();
  });
  // joint coupling q6 - q8 = 0
  real Dj, arefj;
  {
    ModelPtr Q = launder(Pm);
    real par[10]; ldc<10>(Q->eq_par[2], par);
    const real pos = S.q[6] - S.q[8];
    const real imp = impedance(par, pos);
    Dj = imp * rcp_nr(fmax(MINVAL * imp, (1 - imp) * Q->eq_diag[2]));
    arefj = -par[1] * (S.qd[6] - S.qd[8]) - par[0] * imp * pos;
  }
  MCG_TICK_PIN(&Dj, 1); MCG_TICK_PIN(&arefj, 1);
  MCG_TICK(ST_R_CONNECT);
  // joint limits on dofs 0..9: a row exists while violated; sign = d(dist)/dq.
  // Control flow is wave-uniform (__any) with per-lane selects: hipcc (ROCm 7.2) places spill stores of values
  // merged after a lane-divergent region BEFORE the exec mask is restored, silently dropping lanes (see DESIGN.md
  // "Compiler hazard"), so no lane-divergent branch is allowed around code that may spill.
  real Dl[10], arefl[10], sgl[10];
  bool any_limit = false;
  // One laundered pointer and ONE batch of scalar loads for the ten ranges: a wave alone on its SIMD pays every s_load round trip
  // (~100+ clocks) in full, and ten dependent ones in a row were a tenth of the sub-step.
  real jr[10][2];
  { ModelPtr Q = launder(Pm); static_for<10>([&](auto I) { constexpr int j = I; jr[j][0] = Q->jnt_range[j][0]; jr[j][1] = Q->jnt_range[j][1]; }); }
  static_for<10>([&](auto I) {
    constexpr int j = I;
    const real lo = S.q[j] - jr[j][0], hi = jr[j][1] - S.q[j];
    real dist = (lo < 0) ? lo : 0.0, sg = (lo < 0) ? 1.0 : 0.0;
    dist = (hi < 0) ? hi : dist; sg = (hi < 0) ? -1.0 : sg;
    sgl[j] = sg; Dl[j] = 0; arefl[j] = 0;
    if (__any(sg != 0)) {
      ModelPtr Q = launder(Pm);
      real par[10]; ldc<10>(Q->limit_par[j], par);
      const real imp = impedance(par, dist);
      const real D = imp * rcp_nr(fmax(MINVAL * imp, (1 - imp) * Q->limit_diag[j]));
      const real ar = -par[1] * (sg * S.qd[j]) - par[0] * imp * dist;
      Dl[j] = (sg != 0) ? D : 0.0; arefl[j] = (sg != 0) ? ar : 0.0;
      any_limit = any_limit || (sg != 0);
    }
  });
  static_for<10>([&](auto I) { constexpr int j = I; pin(Dl[j]); pin(arefl[j]); pin(sgl[j]); });
  MCG_FENCE();
  MCG_TICK(ST_R_LIMITS);

  // mocap weld: rows 0-2 position (mocap point - weld point), rows 3-5 orientation torquescale * imag(neg(q_tcp) q_m relquat);
  // Jacobian = -(jac of the weld point), its rotational part mapped through 0.5 neg(q_tcp) (.) q_m relquat; arm dofs only.
  // [RECALL mj_instantiateEquality mjEQ_WELD; the common row weight is pinned by the reference keyframe, see the oracle]
  real Jw[6][6], Dw[2] = {0, 0}, arefw[6];       // Dw: translational rows 0-2, rotational rows 3-5
  if constexpr (WLD::enabled) {
    ModelPtr Q = launder(Pm);
    TcpPose X; tcp_forward(Q, S.q, X, true);
    real par[10], rq[4], rp[3]; ldc<10>(Q->weld_par, par); ldc<4>(Q->weld_relquat, rq); ldc<3>(Q->weld_relpos, rp);
    const real ts = Q->weld_torquescale;
    real Rm[9]; quat_to_mat(WD->quat, Rm);
    real cpos[6];
    for (int k = 0; k < 3; k++) cpos[k] = (WD->pos[k] + Rm[3 * k] * rp[0] + Rm[3 * k + 1] * rp[1] + Rm[3 * k + 2] * rp[2]) - X.pos[k];
    real quat[4], quat1[4] = {X.quat[0], -X.quat[1], -X.quat[2], -X.quat[3]}, quat2[4];
    mulquat(WD->quat, rq, quat);
    mulquat(quat1, quat, quat2);
    for (int k = 0; k < 3; k++) cpos[3 + k] = ts * quat2[1 + k];
    static_for<6>([&](auto I) {
      constexpr int j = I;
      for (int k = 0; k < 3; k++) Jw[k][j] = -X.jacp[k][j];
      const real ax[4] = {0, -X.jacr[0][j], -X.jacr[1][j], -X.jacr[2][j]};
      real t4[4], q3[4];
      mulquat(quat1, ax, t4); mulquat(t4, quat, q3);
      for (int k = 0; k < 3; k++) Jw[3 + k][j] = 0.5 * q3[1 + k] * ts;
    });
    real ss = 0; for (int k = 0; k < 6; k++) ss = fma(cpos[k], cpos[k], ss);
    const real imp = impedance(par, sqrt(ss));
    Dw[0] = imp * rcp_nr(fmax(MINVAL * imp, (1 - imp) * Q->weld_diag[0]));
    Dw[1] = imp * rcp_nr(fmax(MINVAL * imp, (1 - imp) * Q->weld_diag[1]));
    static_for<6>([&](auto Rr) {
      constexpr int r = Rr;
      real vel = 0;
      static_for<6>([&](auto I) { constexpr int j = I; vel = fma(Jw[r][j], S.qd[j], vel); });
      arefw[r] = -par[1] * vel - par[0] * imp * cpos[r];
    });
    pin(Dw[0]); pin(Dw[1]); pin6(arefw);
    for (int r = 0; r < 6; r++) pin6(Jw[r]);
    MCG_FENCE();
  }

  MCG_TICK(ST_ROWS);
  // H_eq = M + J^T D J over the equality rows (two connects, gear coupling, mocap weld) is assembled ONCE per sub-step,
  // group by group (arm block, then each side's gripper rows: at most 24 accumulators live), and parked in LDS next to M.
  // The connect / weld Jacobians are dead from here on: the Newton iterations and the line search read H_eq back and add the
  // active limit rows' diagonal, the Euler step needs M only.
  auto assemble_heq = [&](auto WithM) {
    constexpr bool WITH_M = WithM;
  {
    real acc[21];
    static_for<6>([&](auto A_) { constexpr int a = A_;
      static_for<a + 1>([&](auto B_) { constexpr int b = B_; acc[tri(a, b)] = WITH_M ? MS.ld(LDS_M + tri(a, b)) : 0.0; }); });
    static_for<2>([&](auto Sd) { constexpr int sd = Sd;
      static_for<3>([&](auto Kk) { constexpr int k = Kk;
        static_for<6>([&](auto A_) { constexpr int a = A_; const real ja = Dc[sd] * Jc[sd][k][a];
          static_for<a + 1>([&](auto B_) { constexpr int b = B_; acc[tri(a, b)] = fma(ja, Jc[sd][k][b], acc[tri(a, b)]); }); }); }); });
    if constexpr (WLD::enabled)
      static_for<6>([&](auto Rr) { constexpr int r = Rr;
        static_for<6>([&](auto A_) { constexpr int a = A_; const real ja = Dw[r / 3] * Jw[r][a];
          static_for<a + 1>([&](auto B_) { constexpr int b = B_; acc[tri(a, b)] = fma(ja, Jw[r][b], acc[tri(a, b)]); }); }); });
    static_for<6>([&](auto A_) { constexpr int a = A_;
      static_for<a + 1>([&](auto B_) { constexpr int b = B_; MS.st(LDS_HEQ + tri(a, b), acc[tri(a, b)]); }); });
  }
  static_for<2>([&](auto Sd) {
    constexpr int sd = Sd;
    constexpr int idx[9] = {0, 1, 2, 3, 4, 5, 6 + 2 * sd, 7 + 2 * sd, 10 + sd};
    static_for<3>([&](auto Ga) {                         // rows gear, finger, hinge of this side: columns 0 .. own index
      constexpr int a = 6 + Ga; constexpr int i = idx[a];
      real row[9];
      static_for<a + 1>([&](auto B_) { constexpr int b = B_; constexpr int j = idx[b];
        if constexpr (PAT_M.nz[i][j] && WITH_M) row[b] = MS.ld(LDS_M + tri(i, j)); else row[b] = 0.0; });
      static_for<2>([&](auto Kk) { constexpr int k = 2 * Kk;            // the y row has no gripper entries
        const real ja = Dc[sd] * Jc[sd][k][a];
        static_for<a + 1>([&](auto B_) { constexpr int b = B_; row[b] = fma(ja, Jc[sd][k][b], row[b]); }); });
      if constexpr (a == 6) row[6] += Dj;                               // gear coupling q6 - q8: +Dj on (6,6), (8,8)
      static_for<a + 1>([&](auto B_) { constexpr int b = B_; MS.st(LDS_HEQ + tri(i, idx[b]), row[b]); });
    });
  });
  MS.st(LDS_HEQ + tri(8, 6), -Dj);                                      // ... and -Dj on (8,6) (structurally zero in M)
  };
  // ---- P8/P9: g0 = qfrc_smooth + J^T D aref over the equality rows                          (Newton system)
  real g0[NB];
  static_for<NB>([&](auto I) { constexpr int i = I; g0[i] = 0; });
  static_for<2>([&](auto Sd) {
    constexpr int sd = Sd;
    constexpr int idx[9] = {0, 1, 2, 3, 4, 5, 6 + 2 * sd, 7 + 2 * sd, 10 + sd};
    static_for<3>([&](auto Kk) {
      constexpr int k = Kk; constexpr int ncol = (k == 1) ? 6 : 9;
      static_for<ncol>([&](auto A_) { constexpr int a = A_;
        g0[idx[a]] = fma(Dc[sd] * Jc[sd][k][a], arefc[sd][k], g0[idx[a]]); });
    });
  });
  g0[6] += Dj * arefj; g0[8] -= Dj * arefj;
  if constexpr (WLD::enabled)
    static_for<6>([&](auto Rr) { constexpr int r = Rr; const real da = Dw[r / 3] * arefw[r];
      static_for<6>([&](auto I) { constexpr int j = I; g0[j] = fma(Jw[r][j], da, g0[j]); }); });
  if constexpr (SPL::early_heq) assemble_heq(std::false_type{});      // J^T D J only: M is not there yet
#if MCG_DUP == 3        // critical-path probe: the robot wave's assembly twice (same numbers to the same slots)
  if constexpr (SPL::early_heq && SPL::mesh_split) { MCG_FENCE(); static_for<2>([&](auto Sd) { constexpr int sd = Sd; pin(Dc[sd]); }); assemble_heq(std::false_type{}); }
#endif
  // Four-wave PickAndPlace kernel: M and passive - bias are in LDS at S1b already; S1c and S2 only fence the other three waves' mesh
  // phase and solver numbers.  When the workgroup HAS a mesh phase (some lane's candidate mask is set: the same masks every wave reads)
  // this wave runs its whole solve now, under it, and passes S1c / S2 when it is done; when it has none the solve stays behind S2, where
  // it runs beside the cube wave's own solve as before.
  bool ahead = false;
  if constexpr (SPL::enabled) {
    if constexpr (SPL::mesh_split) {
      __syncthreads();                                              // S1b
      ahead = __any(MS.ld(SPL::MASK0) != 0.0 || MS.ld(SPL::MASK0 + 1) != 0.0 || MS.ld(SPL::MASK0 + 2) != 0.0);
      if (!ahead) { __syncthreads(); __syncthreads(); }             // S1c, S2
    } else __syncthreads();                                         // S2: M and passive - bias are in LDS
    if constexpr (SPL::rne_remote) static_for<NB>([&](auto I) { constexpr int i = I; fs[i] += MS.ld(SPL::FS + i); });
  }
  static_for<NB>([&](auto I) { constexpr int i = I; g0[i] += fs[i]; });
  if constexpr (CPL::publishes) CP->publish(g0, Dl, arefl, sgl, S.qd, S.warm, !ahead);
  if constexpr (!SPL::early_heq) assemble_heq(std::true_type{});
  auto build_H = [&](real* H, const bool* act_) {
    static_for<NB>([&](auto I) { constexpr int i = I;
      static_for<i + 1>([&](auto Jj) { constexpr int j = Jj;
        if constexpr (PAT_E.nz[i][j]) {
          H[tri(i, j)] = MS.ld(LDS_HEQ + tri(i, j));
          if constexpr (SPL::early_heq && PAT_M.nz[i][j]) H[tri(i, j)] += MS.ld(LDS_M + tri(i, j));
        } else if constexpr (PAT_H.nz[i][j]) H[tri(i, j)] = 0.0; }); });
    static_for<10>([&](auto I) { constexpr int j = I; H[tri(j, j)] += act_[j] ? Dl[j] : 0.0; });
  };

  MCG_TICK_PIN(g0, NB);
  MCG_TICK(ST_G0);
  // ---- Newton iterations over the limit rows' active set, exact line search               (mj_fwdConstraint)
  // Wave-uniform loop; a lane that has converged keeps recomputing its own (unchanged) system and commits nothing.
  real a[NB];
  bool act[10];
  if constexpr (SPL::warm_lds && !CPL::enabled) {     // the warm start is read below, and only if the general iteration runs
    static_for<NB>([&](auto I) { constexpr int i = I; a[i] = 0; });
    static_for<10>([&](auto I) { constexpr int j = I; act[j] = false; });
  } else {
    static_for<NB>([&](auto I) { constexpr int i = I; a[i] = S.warm[i]; });
    static_for<10>([&](auto I) { constexpr int j = I; act[j] = (sgl[j] != 0) && (sgl[j] * a[j] - arefl[j] < 0); });
  }
  bool conv = false;
  if constexpr (CPL::enabled) {
    if (__any(CP->any_pad)) {     // wave-uniform: every lane of the wave takes the coupled path (same minimiser)
      MCG_TICK(ST_G0);
      CP->template solve_coupled_call<SPL::early_heq>(g0, Dl, arefl, sgl, S.qd, a);      // out of line, on copies (mcg_cube.hpp)
      conv = true;
      MCG_TICK(ST_COUPLED);
    }
  }
  // Direct solve (the usual case: at most two violated limit rows per lane -- in practice the two gear joints, which start
  // every episode ON their lower limit, qpos0 = range[0] = 0).  A limit row is J = +-e_j, so with Z = H_eq^-1 the minimiser is
  //   a = abar - sum_s z_s sg_s nu_s,   abar = Z g0,  z_s = Z e_{j_s},  nu_s = D_s r_s on active rows (r_s < 0), else 0,
  //   r_s = rbar_s - sum_s' W_ss' nu_s',  rbar_s = sg_s abar_{j_s} - aref_s,  W_ss' = sg_s sg_s' Z[j_s, j_s'],
  // a 2x2 complementarity problem whose four active sets are enumerated in closed form: exactly one is consistent (the cost is
  // strictly convex).  One factorisation and at most three solves per sub-step whatever the state, no line search, no warm
  // start: with desynchronised episodes every wave always holds a freshly reset environment, and the iterative path below
  // (two factorisations + a line search for the whole wave whenever one lane's active set moves) was 1.4-2.2x slower there.
  // Same minimiser as the iteration (and as the oracle's Newton solver) to rounding.
  {
    int nviol = 0, j0 = -1, j1 = -1;
    static_for<10>([&](auto I) { constexpr int j = I;
      const bool v = sgl[j] != 0;
      j1 = (v && nviol == 1) ? j : j1; j0 = (v && nviol == 0) ? j : j0; nviol += v ? 1 : 0; });
    if (!conv && !__any(nviol > 2)) {                        // wave-uniform (conv is wave-uniform here)
      MCG_COUNT(CN_NEWTON_IT);
      real L[NB * (NB + 1) / 2], dinv[NB], x[NB];
      const bool none[10] = {false, false, false, false, false, false, false, false, false, false};
      MCG_TICK(ST_NEWTON);
      build_H(L, none);
      static_for<NB>([&](auto I) { constexpr int i = I; x[i] = g0[i]; });
      MCG_TICK_PIN(L, 0); MCG_TICK_PIN(x, NB);
      MCG_TICK(ST_N_BUILD);
      ldl_factor<PAT_H>(L, dinv);
      MCG_TICK_PIN(dinv, NB);
      MCG_TICK(ST_N_FACTOR);
      ldl_solve<PAT_H>(L, dinv, x);
      MCG_TICK_PIN(x, NB);
      MCG_TICK(ST_N_SOLVE);
      if (__any(nviol > 0)) {
        MCG_COUNT(CN_LINESEARCH);                            // counted in the old line search's slot: "sub-steps with limit rows"
        real D0 = 1, D1 = 1, ar0 = 0, ar1 = 0, s0 = 0, s1 = 0, xa0 = 0, xa1 = 0;
        static_for<10>([&](auto I) { constexpr int j = I;
          const bool m0 = (j == j0), m1 = (j == j1);
          D0 = m0 ? Dl[j] : D0; ar0 = m0 ? arefl[j] : ar0; s0 = m0 ? sgl[j] : s0; xa0 = m0 ? x[j] : xa0;
          D1 = m1 ? Dl[j] : D1; ar1 = m1 ? arefl[j] : ar1; s1 = m1 ? sgl[j] : s1; xa1 = m1 ? x[j] : xa1; });
        real z0[NB], z1[NB];
        static_for<NB>([&](auto I) { constexpr int i = I; z0[i] = (i == j0) ? 1.0 : 0.0; z1[i] = (i == j1) ? 1.0 : 0.0; });
        ldl_solve<PAT_H>(L, dinv, z0);
        if (__any(nviol > 1)) ldl_solve<PAT_H>(L, dinv, z1);
        real W00 = 0, W01 = 0, W11 = 0;
        static_for<10>([&](auto I) { constexpr int j = I;
          W00 = (j == j0) ? z0[j] : W00; W01 = (j == j1) ? z0[j] : W01; W11 = (j == j1) ? z1[j] : W11; });
        W01 *= s0 * s1;
        const bool e0 = nviol > 0, e1 = nviol > 1;
        const real rb0 = s0 * xa0 - ar0, rb1 = s1 * xa1 - ar1;
        const real A00 = rcp_nr(D0) + W00, A11 = rcp_nr(D1) + W11;
        const real n0 = rb0 * rcp_nr(A00), n1 = rb1 * rcp_nr(A11);               // one active row
        const real idet = rcp_nr(A00 * A11 - W01 * W01);                          // both active
        const real b0 = (A11 * rb0 - W01 * rb1) * idet, b1 = (A00 * rb1 - W01 * rb0) * idet;
        const bool cN = (!e0 || rb0 >= 0) && (!e1 || rb1 >= 0);
        const bool cB = e0 && e1 && (b0 < 0) && (b1 < 0);
        const bool c0 = e0 && (n0 < 0) && (!e1 || rb1 - W01 * n0 >= 0);
        const bool c1 = e1 && (n1 < 0) && (rb0 - W01 * n1 >= 0);
        // on a boundary (some r or nu within rounding of 0) two sets or none may pass: their solutions coincide there
        real nu0 = e1 ? b0 : n0, nu1 = e1 ? b1 : 0.0;
        nu0 = c1 ? 0.0 : nu0; nu1 = c1 ? n1 : nu1;
        nu0 = c0 ? n0 : nu0;  nu1 = c0 ? 0.0 : nu1;
        nu0 = cB ? b0 : nu0;  nu1 = cB ? b1 : nu1;
        nu0 = cN ? 0.0 : nu0; nu1 = cN ? 0.0 : nu1;
        const real f0 = s0 * nu0, f1 = s1 * nu1;
        static_for<NB>([&](auto I) { constexpr int i = I; x[i] = fma(-z1[i], f1, fma(-z0[i], f0, x[i])); });
      }
      static_for<NB>([&](auto I) { constexpr int i = I; a[i] = x[i]; });
      conv = true;
      MCG_TICK_PIN(a, NB);
      MCG_TICK(ST_N_CHECK);
    }
  }
  // General iteration (three or more violated limit rows in some lane of the wave: arm or finger limits, rare)
  if constexpr (SPL::warm_lds && !CPL::enabled) {
    if (__any(!conv)) {
      static_for<NB>([&](auto I) { constexpr int i = I; a[i] = MS.ld(SPL::WARM + i); });
      static_for<10>([&](auto I) { constexpr int j = I; act[j] = (sgl[j] != 0) && (sgl[j] * a[j] - arefl[j] < 0); });
    }
  }
  for (int it = 0; it < 50 && __any(!conv); it++) {
    MCG_COUNT(CN_NEWTON_IT);
    real L[NB * (NB + 1) / 2], dinv[NB], x[NB];
    MCG_TICK_PIN(a, NB);
  MCG_TICK(ST_NEWTON);
    build_H(L, act);
    static_for<NB>([&](auto I) { constexpr int i = I; x[i] = g0[i]; });
    static_for<10>([&](auto I) { constexpr int j = I; x[j] += act[j] ? sgl[j] * Dl[j] * arefl[j] : 0.0; });
    MCG_TICK_PIN(L, 0); MCG_TICK_PIN(x, NB);
    MCG_TICK(ST_N_BUILD);
    ldl_factor<PAT_H>(L, dinv);
    MCG_TICK_PIN(dinv, NB);
    MCG_TICK(ST_N_FACTOR);
    ldl_solve<PAT_H>(L, dinv, x);
    MCG_TICK_PIN(x, NB);
    MCG_TICK(ST_N_SOLVE);
    bool same = true;            // does the minimiser of this quadratic piece keep the assumed active set?
    static_for<10>([&](auto I) { constexpr int j = I;
      const bool now = (sgl[j] != 0) && (sgl[j] * x[j] - arefl[j] < 0);
      same = same && (now == act[j]); });
    const bool finish = !conv && (!any_limit || same);
    static_for<NB>([&](auto I) { constexpr int i = I; a[i] = finish ? x[i] : a[i]; });
    conv = conv || finish;
    MCG_TICK_PIN(a, NB);
    MCG_TICK(ST_N_CHECK);
    if (!__any(!conv)) break;
    // some lane crossed a breakpoint: exact line search from a along p = x - a (committed where !conv only);
    // phi'(alpha) = (alpha - 1) p^T H p on the first piece, its slope changes by +-D p_j^2 at each breakpoint
    MCG_COUNT(CN_LINESEARCH);
    real p[NB], Hp[NB];
    static_for<NB>([&](auto I) { constexpr int i = I; p[i] = x[i] - a[i]; });
    build_H(L, act);
    static_for<NB>([&](auto I) { constexpr int i = I; real sacc = 0;
      static_for<NB>([&](auto Jj) { constexpr int j = Jj;
        if constexpr (PAT_H.nz[i > j ? i : j][i > j ? j : i]) sacc = fma(L[tri(i, j)], p[j], sacc); }); Hp[i] = sacc; });
    real slope = 0;
    static_for<NB>([&](auto I) { constexpr int i = I; slope = fma(p[i], Hp[i], slope); });
    real val = -slope, alpha = 0;
    real bp[10];
    static_for<10>([&](auto I) { constexpr int j = I;
      const real rj = sgl[j] * a[j] - arefl[j], dj = sgl[j] * p[j];
      const real al_ = (sgl[j] != 0 && dj != 0) ? -rj / dj : -1.0;
      bp[j] = al_ > 0 ? al_ : INFINITY; });
    bool ls_done = false;
    for (int step = 0; step <= 10; step++) {               // uniform trip count, predicated body
      real nxt = INFINITY; int jn = -1;
      static_for<10>([&](auto I) { constexpr int j = I; const bool lt = bp[j] < nxt; nxt = lt ? bp[j] : nxt; jn = lt ? j : jn; });
      const bool root = !ls_done && (slope > 0) && (val + slope * (nxt - alpha) >= 0);
      alpha = root ? alpha - val / slope : alpha;
      ls_done = ls_done || root;
      const bool none = !ls_done && (jn < 0);
      alpha = none ? 1.0 : alpha;
      ls_done = ls_done || none;
      const bool adv = !ls_done;
      val = adv ? val + slope * (nxt - alpha) : val;
      alpha = adv ? nxt : alpha;
      static_for<10>([&](auto I) { constexpr int j = I;
        const bool hit = adv && (j == jn);
        const real rj = sgl[j] * a[j] - arefl[j];
        const real dsl = Dl[j] * p[j] * p[j];
        slope = hit ? (rj < 0 ? slope - dsl : slope + dsl) : slope;
        bp[j] = hit ? INFINITY : bp[j]; });
    }
    static_for<NB>([&](auto I) { constexpr int i = I; a[i] = conv ? a[i] : fma(alpha, p[i], a[i]); });
    static_for<10>([&](auto I) { constexpr int j = I;
      const bool now = (sgl[j] != 0) && (sgl[j] * a[j] - arefl[j] < 0);
      act[j] = conv ? act[j] : now; });
  }

  MCG_TICK_PIN(a, NB);
  MCG_TICK(ST_NEWTON);
  // ---- constraint forces -> qfrc_constraint; P10 implicit-damping Euler                  (mj_Euler, mj_advance)
  real rhs[NB];
  if constexpr (SPL::factor_remote) {
    __syncthreads();                                                // S3: the helper's factor of M + hB is in LDS
    real Lf[NB * (NB + 1) / 2], dinv[NB];
    static_for<NB>([&](auto I) { constexpr int i = I; dinv[i] = MS.ld(LDS_FDINV + i);
      static_for<i>([&](auto Jj) { constexpr int j = Jj; if constexpr (PAT_M.nz[i][j]) Lf[tri(i, j)] = MS.ld(LDS_FAC + tri(i, j)); }); });
    { ModelPtr Q = launder(Pm); static_for<NB>([&](auto I) { constexpr int i = I; rhs[i] = Q->body[i].damping * a[i]; }); }      // one batch of s_loads
    ldl_solve<PAT_M>(Lf, dinv, rhs);
    static_for<NB>([&](auto I) { constexpr int i = I; rhs[i] = fma(-h, rhs[i], a[i]); });
  } else {
  MCG_TICK(ST_E_RHS);
  euler_accel(Pm, h, MS, a, rhs);
  }
  real qn[NB], qdn[NB];
  bool bad = false;
  static_for<NB>([&](auto I) { constexpr int i = I;
    real q_old, qd_old;
    if constexpr (SPL::warm_lds && !CPL::enabled) { q_old = MS.ld(SPL::QB + i); qd_old = MS.ld(SPL::QDB + i); if constexpr (i < 6) MS.st(SPL::QLAG + i, q_old); }
    else { q_old = S.q[i]; qd_old = S.qd[i]; }
    qdn[i] = fma(h, rhs[i], qd_old); qn[i] = fma(h, qdn[i], q_old);
    if constexpr (COMMIT) bad = bad || bad_value(qn[i]) || bad_value(qdn[i]) || bad_value(a[i]); });
  // mj_step checks qpos / qvel when it starts and qacc after mj_forward, and calls mj_resetData (qpos0, zero velocity, controls and
  // warm start) on a bad value [RECALL]: the new state is checked here, at the end of the sub-step that produced it -- the same
  // state the next mj_step would check first.  (Round 2 checked once per env-step.)
  if constexpr (COMMIT) {
    if (__any(bad)) {                                       // wave-uniform; rare
      static_for<NB>([&](auto I) { constexpr int i = I; qn[i] = sel(bad, 0.0, qn[i]); qdn[i] = sel(bad, 0.0, qdn[i]); a[i] = sel(bad, 0.0, a[i]); });
      static_for<7>([&](auto I) { constexpr int k = I; S.ctrl[k] = sel(bad, 0.0, S.ctrl[k]); });
    }
  }
  static_for<NB>([&](auto I) { constexpr int i = I;
    const real qd_new = qdn[i], q_new = qn[i];
    if constexpr (COMMIT) { S.qd[i] = qd_new; S.q[i] = q_new; if constexpr (SPL::warm_lds && !CPL::enabled) MS.st(SPL::WARM + i, a[i]); else S.warm[i] = a[i]; }
    else { next->qd[i] = qd_new; next->q[i] = q_new; next->warm[i] = a[i]; }
    if constexpr (SPL::enabled && COMMIT) { MS.st(SPL::QB + i, q_new); MS.st(SPL::QDB + i, qd_new); } });
  MCG_TICK_PIN(S.q, NB); MCG_TICK_PIN(S.qd, NB);
  MCG_TICK(ST_EULER);
  if constexpr (SPL::enabled && SPL::mesh_split) { if (ahead) { __syncthreads(); __syncthreads(); } }      // S1c, S2
  return bad;
}

// The helper wave's share of one sub-step (see SplitMain).
template <class SPL = SplitMain, class LS, class SIDE = NoSideWork>
MCG_DEV void helper_substep(ModelPtr Pm, const LS MS, const SIDE& side = SIDE{}) {
  __syncthreads();                                                  // S1
  real cs[NB], sn[NB];
  {
    const TrigC T = load_trig();
    static_for<NB>([&](auto I) { constexpr int i = I; sincos_cw(T, AXS[i] * MS.ld(SPL::QB + i), sn[i], cs[i]); });
  }
  static_for<NB>([&](auto I) { constexpr int i = I; pin(sn[i]); pin(cs[i]); });
  MCG_FENCE();
  crb_to_lds(Pm, cs, sn, MS);
  static_assert(!SPL::mesh_split, "the four-wave PickAndPlace kernel drives its side waves itself: helper_pre / rne_pre");
  (void)side;
  __syncthreads();                                                  // S2
  if constexpr (SPL::factor_remote) {
    const real h = launder(Pm)->timestep;
    real Mh[NB * (NB + 1) / 2], dinv[NB];
    static_for<NB>([&](auto I) { constexpr int i = I;
      static_for<i + 1>([&](auto Jj) { constexpr int j = Jj; if constexpr (PAT_M.nz[i][j]) Mh[tri(i, j)] = MS.ld(LDS_M + tri(i, j)); }); });
    { ModelPtr Q = launder(Pm); static_for<NB>([&](auto I) { constexpr int i = I; Mh[tri(i, i)] = fma(h, Q->body[i].damping, Mh[tri(i, i)]); }); }
    ldl_factor<PAT_M>(Mh, dinv);
    static_for<NB>([&](auto I) { constexpr int i = I; MS.st(LDS_FDINV + i, dinv[i]);
      static_for<i>([&](auto Jj) { constexpr int j = Jj; if constexpr (PAT_M.nz[i][j]) MS.st(LDS_FAC + tri(i, j), Mh[tri(i, j)]); }); });
    __syncthreads();                                                // S3
  }
}

// The RNE wave's share of one sub-step (see SplitMain).
template <class SPL = SplitMain, class LS, class SIDE = NoSideWork>
MCG_DEV void rne_substep(ModelPtr Pm, const LS MS, const SIDE& side = SIDE{}) {
  __syncthreads();                                                  // S1
  real cs[NB], sn[NB], qd[NB], fs[NB];
  {
    const TrigC T = load_trig();
    static_for<NB>([&](auto I) { constexpr int i = I; sincos_cw(T, AXS[i] * MS.ld(SPL::QB + i), sn[i], cs[i]); qd[i] = MS.ld(SPL::QDB + i); });
  }
  static_for<NB>([&](auto I) { constexpr int i = I; pin(sn[i]); pin(cs[i]); });
  MCG_FENCE();
  rne_bias(Pm, cs, sn, qd, fs);
  static_for<NB>([&](auto I) { constexpr int i = I; MS.st(SPL::FS + i, fs[i]); });
  static_assert(!SPL::mesh_split, "the four-wave PickAndPlace kernel drives its side waves itself: helper_pre / rne_pre");
  (void)side;
  __syncthreads();                                                  // S2
  if constexpr (SPL::factor_remote) __syncthreads();                // S3
}

// The same two shares for the four-wave PickAndPlace kernel, up to its barrier S1b: S1, the wave's own work (M into LDS / passive - bias
// into LDS), then `side(sn, cs)` -- the broad phase of the arm meshes, from the sines / cosines the wave holds anyway.  The caller runs
// the barriers S1b .. S5 and the phases between them (mcg_hip.hip: pnp_side_wave).
template <class SPL, class LS, class SIDE>
MCG_DEV void helper_pre(ModelPtr Pm, const LS MS, const SIDE& side) {
  __syncthreads();                                                  // S1
  MCG_TICK(ST_C_CHECK);
  real cs[NB], sn[NB];
  {
    const TrigC T = load_trig();
    static_for<NB>([&](auto I) { constexpr int i = I; sincos_cw(T, AXS[i] * MS.ld(SPL::QB + i), sn[i], cs[i]); });
  }
  static_for<NB>([&](auto I) { constexpr int i = I; pin(sn[i]); pin(cs[i]); });
  MCG_FENCE();
  crb_to_lds(Pm, cs, sn, MS);
  MCG_TICK(ST_C_MASK);
  side(sn, cs);
  MCG_TICK(ST_C_ASSEMBLE);
}
template <class SPL, class LS, class SIDE>
MCG_DEV void rne_pre(ModelPtr Pm, const LS MS, const SIDE& side) {
  __syncthreads();                                                  // S1
  MCG_TICK(ST_C_CHECK);
  real cs[NB], sn[NB], qd[NB], fs[NB];
  {
    const TrigC T = load_trig();
    static_for<NB>([&](auto I) { constexpr int i = I; sincos_cw(T, AXS[i] * MS.ld(SPL::QB + i), sn[i], cs[i]); qd[i] = MS.ld(SPL::QDB + i); });
  }
  static_for<NB>([&](auto I) { constexpr int i = I; pin(sn[i]); pin(cs[i]); });
  MCG_FENCE();
  rne_bias(Pm, cs, sn, qd, fs);
  static_for<NB>([&](auto I) { constexpr int i = I; MS.st(SPL::FS + i, fs[i]); });
  MCG_TICK(ST_C_MASK);
  side(sn, cs);
  MCG_TICK(ST_C_ASSEMBLE);
}

// ---------------------------------------------------------------------------------- world-frame arm kinematics
// EEF site pose and its 6x6 Jacobian at arm angles q6 (mj_kinematics + mj_jacSite for site EEF, P1/P11).
struct EefPose { real pos[3], mat[9], jacp[3][6], jacr[3][6]; };

MCG_DEV void eef_forward(ModelPtr P, const real* q6, EefPose& E, bool want_jac) {
  real R[9], p[3], anchor[6][3], axis[6][3];
  const TrigC T = load_trig();
  for (int k = 0; k < 9; k++) R[k] = P->base_mat[k];
  for (int k = 0; k < 3; k++) p[k] = P->base_pos[k];
  static_for<6>([&](auto I) {
    constexpr int i = I; constexpr int K = AXK[i]; constexpr int A = (K + 1) % 3, B = (K + 2) % 3;
    real r[3]; ldc<3>(P->body[i].r, r);
    for (int k = 0; k < 3; k++) p[k] += R[3 * k] * r[0] + R[3 * k + 1] * r[1] + R[3 * k + 2] * r[2];
    for (int k = 0; k < 3; k++) { anchor[i][k] = p[k]; axis[i][k] = AXS[i] * R[3 * k + K]; }
    real s, c; sincos_cw(T, AXS[i] * q6[i], s, c);
    for (int k = 0; k < 3; k++) {      // R <- R * Rot(e_K, theta): mixes columns A and B
      real ca = R[3 * k + A], cb = R[3 * k + B];
      R[3 * k + A] = c * ca + s * cb; R[3 * k + B] = -s * ca + c * cb;
    }
  });
  real se[3]; ldc<3>(P->site_eef, se);
  for (int k = 0; k < 3; k++) E.pos[k] = p[k] + R[3 * k] * se[0] + R[3 * k + 1] * se[1] + R[3 * k + 2] * se[2];
  for (int k = 0; k < 9; k++) E.mat[k] = R[k];
  if (want_jac) {
    static_for<6>([&](auto I) {
      constexpr int i = I;
      real d[3] = {E.pos[0] - anchor[i][0], E.pos[1] - anchor[i][1], E.pos[2] - anchor[i][2]}, c3[3];
      cross(axis[i], d, c3);
      for (int k = 0; k < 3; k++) { E.jacp[k][i] = c3[k]; E.jacr[k][i] = axis[i][k]; }
    });
  }
}

// mju_mat2Quat / mju_mulQuat / mju_quat2Vel as the reference's IK controller uses them (utils.py:525-528) [RECALL]
MCG_DEV void normalize4(real* q) {
  real n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { real inv = 1 / n; (void)inv; q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
MCG_DEV void mat2quat(const real* m, real* q) {
  if (m[0] + m[4] + m[8] > 0) {
    q[0] = 0.5 * sqrt(1 + m[0] + m[4] + m[8]);
    q[1] = 0.25 * (m[7] - m[5]) / q[0]; q[2] = 0.25 * (m[2] - m[6]) / q[0]; q[3] = 0.25 * (m[3] - m[1]) / q[0];
  } else if (m[0] > m[4] && m[0] > m[8]) {
    q[1] = 0.5 * sqrt(1 + m[0] - m[4] - m[8]);
    q[0] = 0.25 * (m[7] - m[5]) / q[1]; q[2] = 0.25 * (m[1] + m[3]) / q[1]; q[3] = 0.25 * (m[2] + m[6]) / q[1];
  } else if (m[4] > m[8]) {
    q[2] = 0.5 * sqrt(1 - m[0] + m[4] - m[8]);
    q[0] = 0.25 * (m[2] - m[6]) / q[2]; q[1] = 0.25 * (m[1] + m[3]) / q[2]; q[3] = 0.25 * (m[5] + m[7]) / q[2];
  } else {
    q[3] = 0.5 * sqrt(1 - m[0] - m[4] + m[8]);
    q[0] = 0.25 * (m[3] - m[1]) / q[3]; q[1] = 0.25 * (m[2] + m[6]) / q[3]; q[2] = 0.25 * (m[5] + m[7]) / q[3];
  }
  normalize4(q);
}
MCG_DEV void mulquat(const real* a, const real* b, real* r) {
  real t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  real t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  real t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  real t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
MCG_DEV void quat2vel(const real* q, real dt, real* res) {
  real ax[3] = {q[1], q[2], q[3]};
  real n = sqrt(dot3(ax, ax));
  if (n < MINVAL) { ax[0] = 1; ax[1] = 0; ax[2] = 0; } else { ax[0] /= n; ax[1] /= n; ax[2] /= n; }
  real speed = 2 * atan2(n, q[0]);
  if (speed > 3.14159265358979323846) speed -= 2 * 3.14159265358979323846;
  speed /= dt;
  res[0] = ax[0] * speed; res[1] = ax[1] * speed; res[2] = ax[2] * speed;
}

// IKController.compute_qpos_delta + solve_DLS (utils.py:499-556): only the six arm columns of the site
// Jacobian are non-zero, so the 18x18 lstsq reduces exactly to this 6x6 SPD solve.
MCG_DEV void ik_delta(const EefPose& E, const real* target_pos, const real* target_quat, real* dq6) {
  real err[6], q[4], nq[4], eq[4];
  for (int k = 0; k < 3; k++) err[k] = target_pos[k] - E.pos[k];
  mat2quat(E.mat, q);
  nq[0] = q[0]; nq[1] = -q[1]; nq[2] = -q[2]; nq[3] = -q[3];
  mulquat(target_quat, nq, eq);
  quat2vel(eq, 50.0, err + 3);
  real A[21], x[6];
  static_for<6>([&](auto I) {
    constexpr int i = I;
    static_for<i + 1>([&](auto Jj) {
      constexpr int j = Jj;
      real s = 0;
      for (int k = 0; k < 3; k++) s += E.jacp[k][i] * E.jacp[k][j] + E.jacr[k][i] * E.jacr[k][j];
      A[tri(i, j)] = s + (i == j ? 0.3 : 0.0);
    });
    real s = 0;
    for (int k = 0; k < 3; k++) s += E.jacp[k][i] * err[k] + E.jacr[k][i] * err[3 + k];
    x[i] = s;
  });
  chol_factor<6>(A);
  chol_solve<6>(A, x);
  for (int k = 0; k < 6; k++) dq6[k] = x[k];
}

// rotations.euler2quat (gymnasium_robotics) as called at mycobot.py:142 [RECALL]
MCG_DEV void euler2quat(const real* e, real* q) {
  real ai = e[2] / 2, aj = -e[1] / 2, ak = e[0] / 2;
  real si, ci, sj, cj, sk, ck;
  sincos(ai, &si, &ci); sincos(aj, &sj, &cj); sincos(ak, &sk, &ck);
  real cc = ci * ck, cs_ = ci * sk, sc = si * ck, ss = si * sk;
  q[0] = cj * cc + sj * ss; q[3] = cj * sc - sj * cs_; q[2] = -(cj * ss + sj * cc); q[1] = cj * cs_ - sj * sc;
}

// ------------------------------------------------------------------------------------------------ Philox4x32-10
MCG_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

}  // namespace mcg
