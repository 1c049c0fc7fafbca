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

// Impedance sigmoid (MuJoCo getimpedance [RECALL]); par = K B d0 dmax width midpoint power
MCG_DEV real impedance(const real* par, real dist) {
  real d0 = par[2], dmax = par[3], width = par[4], mid = par[5], power = par[6];
  real imp;
  if (d0 == dmax || width <= MINVAL) imp = 0.5 * (d0 + dmax);
  else {
    real x = fabs(dist) / width;
    if (x >= 1) imp = dmax;
    else if (x == 0) imp = d0;
    else {
      real y;
      if (power == 1) y = x;
      else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
      else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
      imp = d0 + y * (dmax - d0);
    }
  }
  return fmin(fmax(imp, MINIMP), MAXIMP);
}

// dense symmetric positive definite solve, n = 12, lower triangle packed row-major: A[i*(i+1)/2 + j]
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
constexpr int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// ------------------------------------------------------------------------------------- robot sub-step state
struct Robot {
  real q[NB], qd[NB], ctrl[7], warm[NB];
};

// One physics sub-step (mj_step) of the 12-dof robot.  `qlag` receives the positions the forward pass used.
MCG_DEV void robot_substep(const mcg_model* __restrict__ P, Robot& S, real* qlag6) {
  const real h = P->timestep;
  real cs[NB], sn[NB];
  static_for<NB>([&](auto I) { constexpr int i = I; sincos(AXS[i] * S.q[i], &sn[i], &cs[i]); });
  static_for<6>([&](auto I) { constexpr int i = I; qlag6[i] = S.q[i]; });

  // ---- P6 recursive Newton-Euler, q'' = 0: bias = Coriolis + centrifugal + gravity        (mj_rne, flg_acc=0)
  real F[NB][3], Nn[NB][3];                 // net force / moment about the body origin, body frame
  real w[NB][3], al[NB][3], ac[NB][3];      // angular velocity, angular acceleration, linear acceleration of the origin
  static_for<NB>([&](auto I) {
    constexpr int i = I; constexpr int p = PAR[i]; constexpr int K = AXK[i];
    real wp[3], alp[3], ap[3];
    if constexpr (p < 0) {
      wp[0] = wp[1] = wp[2] = 0; alp[0] = alp[1] = alp[2] = 0;
      ap[0] = P->gravity_base[0]; ap[1] = P->gravity_base[1]; ap[2] = P->gravity_base[2];
    } else {
      for (int k = 0; k < 3; k++) { wp[k] = w[p][k]; alp[k] = al[p][k]; ap[k] = ac[p][k]; }
    }
    const real* r = P->r[i];
    real t[3], accp[3];
    cross(wp, r, t);
    accp[0] = ap[0]; accp[1] = ap[1]; accp[2] = ap[2];
    cross_add(alp, r, accp); cross_add(wp, t, accp);
    rot_down<K>(cs[i], sn[i], accp, ac[i]);
    real we[3];
    rot_down<K>(cs[i], sn[i], wp, we);
    rot_down<K>(cs[i], sn[i], alp, al[i]);
    const real g = AXS[i] * S.qd[i];
    constexpr int A = (K + 1) % 3, B = (K + 2) % 3;
    // al += (E wp) x (g e_K)
    al[i][A] += we[B] * g; al[i][B] -= we[A] * g;
    w[i][0] = we[0]; w[i][1] = we[1]; w[i][2] = we[2]; w[i][K] += g;
    // wrench
    const real m = P->mass[i]; const real* mc = P->mc[i]; const real* In = P->inertia[i];
    real t2[3], Iw[3];
    cross(w[i], mc, t2);
    F[i][0] = m * ac[i][0]; F[i][1] = m * ac[i][1]; F[i][2] = m * ac[i][2];
    cross_add(al[i], mc, F[i]); cross_add(w[i], t2, F[i]);
    sym_mul(In, al[i], Nn[i]); sym_mul(In, w[i], Iw);
    cross_add(w[i], Iw, Nn[i]); cross_add(mc, ac[i], Nn[i]);
  });
  real bias[NB];
  static_for<NB>([&](auto I) {
    constexpr int i = NB - 1 - I; constexpr int p = PAR[i]; constexpr int K = AXK[i];
    bias[i] = AXS[i] * Nn[i][K];
    if constexpr (p >= 0) {
      real fp[3], np[3];
      rot_up<K>(cs[i], sn[i], F[i], fp); rot_up<K>(cs[i], sn[i], Nn[i], np);
      cross_add(P->r[i], fp, np);
      for (int k = 0; k < 3; k++) { F[p][k] += fp[k]; Nn[p][k] += np[k]; }
    }
  });

  // ---- P3 composite rigid bodies -> joint-space inertia M (packed lower triangle)             (mj_crb)
  real M[NB * (NB + 1) / 2];
  for (int k = 0; k < NB * (NB + 1) / 2; k++) M[k] = 0;
  real cm[NB], cmc[NB][3], cI[NB][6];
  static_for<NB>([&](auto I) {
    constexpr int i = I;
    cm[i] = P->mass[i];
    for (int k = 0; k < 3; k++) cmc[i][k] = P->mc[i][k];
    for (int k = 0; k < 6; k++) cI[i][k] = P->inertia[i][k];
  });
  static_for<NB>([&](auto I) {
    constexpr int i = NB - 1 - I; constexpr int K = AXK[i]; constexpr int A = (K + 1) % 3, B = (K + 2) % 3;
    // wrench of a unit acceleration about joint i acting on the composite body i (frame i, about origin i)
    real f[3], n[3];
    constexpr int iKA = (K + A == 1) ? 3 : (K + A == 2) ? 4 : 5;
    constexpr int iKB = (K + B == 1) ? 3 : (K + B == 2) ? 4 : 5;
    const real sg = AXS[i];
    n[K] = sg * cI[i][K]; n[A] = sg * cI[i][iKA]; n[B] = sg * cI[i][iKB];
    f[K] = 0; f[A] = -sg * cmc[i][B]; f[B] = sg * cmc[i][A];          // (sg e_K) x mc
    M[tri(i, i)] = cI[i][K] + P->armature[i];
    // walk to the root: M[i][j] = axis_j . moment about origin j
    real fj[3] = {f[0], f[1], f[2]}, nj[3] = {n[0], n[1], n[2]};
    auto up = [&](auto self, auto Cur) -> void {
      constexpr int cur = Cur; constexpr int pj = PAR[cur];
      if constexpr (pj >= 0) {
        real f2[3], n2[3];
        rot_up<AXK[cur]>(cs[cur], sn[cur], fj, f2); rot_up<AXK[cur]>(cs[cur], sn[cur], nj, n2);
        cross_add(P->r[cur], f2, n2);
        for (int k = 0; k < 3; k++) { fj[k] = f2[k]; nj[k] = n2[k]; }
        M[tri(i, pj)] = AXS[pj] * nj[AXK[pj]];
        self(self, std::integral_constant<int, pj>{});
      }
    };
    up(up, std::integral_constant<int, i>{});
    // add composite i to its parent
    constexpr int p = PAR[i];
    if constexpr (p >= 0) {
      real It[6], h3[3];
      for (int k = 0; k < 6; k++) It[k] = cI[i][k];
      sym_rot_up<K>(cs[i], sn[i], It);
      rot_up<K>(cs[i], sn[i], cmc[i], h3);
      const real* r = P->r[i]; const real m = cm[i];
      real rr = dot3(r, r), rh = dot3(r, h3);
      real d = m * rr + 2 * rh;
      cI[p][0] += It[0] + d - (m * r[0] * r[0] + 2 * r[0] * h3[0]);
      cI[p][1] += It[1] + d - (m * r[1] * r[1] + 2 * r[1] * h3[1]);
      cI[p][2] += It[2] + d - (m * r[2] * r[2] + 2 * r[2] * h3[2]);
      cI[p][3] += It[3] - (m * r[0] * r[1] + r[0] * h3[1] + h3[0] * r[1]);
      cI[p][4] += It[4] - (m * r[0] * r[2] + r[0] * h3[2] + h3[0] * r[2]);
      cI[p][5] += It[5] - (m * r[1] * r[2] + r[1] * h3[2] + h3[1] * r[2]);
      cm[p] += m;
      for (int k = 0; k < 3; k++) cmc[p][k] += h3[k] + m * r[k];
    }
  });

  // ---- P7 actuation + passive damping -> qfrc_smooth                       (mj_fwdActuation, mj_passive)
  real fs[NB];
  static_for<NB>([&](auto I) { constexpr int i = I; fs[i] = -P->damping[i] * S.qd[i] - bias[i]; });
  static_for<6>([&](auto I) {
    constexpr int u = I;
    real c = fmin(fmax(S.ctrl[u], P->act_ctrlrange[u][0]), P->act_ctrlrange[u][1]);
    real f = P->act_gain[u] * c + P->act_bias[u][0] + P->act_bias[u][1] * S.q[u] + P->act_bias[u][2] * S.qd[u];
    f = fmin(fmax(f, P->act_forcerange[u][0]), P->act_forcerange[u][1]);
    fs[u] += f;
  });
  {
    const real c0 = P->tendon_coef[0], c1 = P->tendon_coef[1];
    real len = c0 * S.q[6] + c1 * S.q[8], vel = c0 * S.qd[6] + c1 * S.qd[8];
    real c = fmin(fmax(S.ctrl[6], P->act_ctrlrange[6][0]), P->act_ctrlrange[6][1]);
    real f = P->act_gain[6] * c + P->act_bias[6][0] + P->act_bias[6][1] * len + P->act_bias[6][2] * vel;
    f = fmin(fmax(f, P->act_forcerange[6][0]), P->act_forcerange[6][1]);
    fs[6] += c0 * f; fs[8] += c1 * f;
  }

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
  // two connects; side 0: gear 6 / finger 7 / hinge 10, side 1: gear 8 / finger 9 / hinge 11
  real Jc[2][3][9];      // rows x (arm 0..5, gear, finger, hinge)
  real Dc[2], arefc[2][3];
  static_for<2>([&](auto Sd) {
    constexpr int sd = Sd; constexpr int g = 6 + 2 * sd, fi = 7 + 2 * sd, hg = 10 + sd;
    // all three joints turn about +-y of the link6 frame: compose by plane rotations
    real o_g[3] = {P->r[g][0], P->r[g][1], P->r[g][2]};
    real t[3], o_f[3], p1[3], p2[3];
    rot_up<1>(cs[g], sn[g], P->r[fi], t);
    for (int k = 0; k < 3; k++) o_f[k] = o_g[k] + t[k];
    real t2[3];
    rot_up<1>(cs[fi], sn[fi], P->eq_anchor1[sd], t2); rot_up<1>(cs[g], sn[g], t2, t);
    for (int k = 0; k < 3; k++) p1[k] = o_f[k] + t[k];
    rot_up<1>(cs[hg], sn[hg], P->eq_anchor2[sd], t);
    for (int k = 0; k < 3; k++) p2[k] = P->r[hg][k] + t[k];
    real pos[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    // Jacobian columns: axis x lever
    static_for<6>([&](auto I) { constexpr int i = I; real c3[3]; cross(ax5[i], pos, c3); for (int k = 0; k < 3; k++) Jc[sd][k][i] = c3[k]; });
    real lg[3] = {p1[0] - o_g[0], p1[1] - o_g[1], p1[2] - o_g[2]};
    real lf[3] = {p1[0] - o_f[0], p1[1] - o_f[1], p1[2] - o_f[2]};
    real lh[3] = {p2[0] - P->r[hg][0], p2[1] - P->r[hg][1], p2[2] - P->r[hg][2]};
    // (s e_y) x d = s (d_z, 0, -d_x)
    Jc[sd][0][6] = AXS[g] * lg[2];  Jc[sd][1][6] = 0; Jc[sd][2][6] = -AXS[g] * lg[0];
    Jc[sd][0][7] = AXS[fi] * lf[2]; Jc[sd][1][7] = 0; Jc[sd][2][7] = -AXS[fi] * lf[0];
    Jc[sd][0][8] = -AXS[hg] * lh[2]; Jc[sd][1][8] = 0; Jc[sd][2][8] = AXS[hg] * lh[0];
    real nrm = sqrt(dot3(pos, pos));
    real imp = impedance(P->eq_par[sd], nrm);
    real R = fmax(MINVAL, (1 - imp) * P->eq_diag[sd] / imp);
    Dc[sd] = 1 / R;
    for (int k = 0; k < 3; k++) {
      real vel = 0;
      static_for<6>([&](auto I) { constexpr int i = I; vel += Jc[sd][k][i] * S.qd[i]; });
      vel += Jc[sd][k][6] * S.qd[g] + Jc[sd][k][7] * S.qd[fi] + Jc[sd][k][8] * S.qd[hg];
      arefc[sd][k] = -P->eq_par[sd][1] * vel - P->eq_par[sd][0] * imp * pos[k];
    }
  });
  // joint coupling q6 - q8 = 0
  real Dj, arefj;
  {
    real pos = S.q[6] - S.q[8];
    real imp = impedance(P->eq_par[2], pos);
    Dj = 1 / fmax(MINVAL, (1 - imp) * P->eq_diag[2] / imp);
    arefj = -P->eq_par[2][1] * (S.qd[6] - S.qd[8]) - P->eq_par[2][0] * imp * pos;
  }
  // joint limits on dofs 0..9: a row exists while violated; sign = d(dist)/dq
  real Dl[10], arefl[10], sgl[10];
  bool any_limit = false;
  static_for<10>([&](auto I) {
    constexpr int j = I;
    real lo = S.q[j] - P->jnt_range[j][0], hi = P->jnt_range[j][1] - S.q[j];
    real dist = 0, sg = 0;
    if (lo < 0) { dist = lo; sg = 1; }
    if (hi < 0) { dist = hi; sg = -1; }
    sgl[j] = sg; Dl[j] = 0; arefl[j] = 0;
    if (sg != 0) {
      real imp = impedance(P->limit_par[j], dist);
      Dl[j] = 1 / fmax(MINVAL, (1 - imp) * P->limit_diag[j] / imp);
      arefl[j] = -P->limit_par[j][1] * (sg * S.qd[j]) - P->limit_par[j][0] * imp * dist;
      any_limit = true;
    }
  });

  // ---- P8/P9: H0 = M + J^T D J over the equality rows, g0 = qfrc_smooth + J^T D aref      (Newton system)
  real H0[NB * (NB + 1) / 2], g0[NB];
  for (int k = 0; k < NB * (NB + 1) / 2; k++) H0[k] = M[k];
  static_for<NB>([&](auto I) { constexpr int i = I; g0[i] = fs[i]; });
  static_for<2>([&](auto Sd) {
    constexpr int sd = Sd;
    constexpr int idx[9] = {0, 1, 2, 3, 4, 5, 6 + 2 * sd, 7 + 2 * sd, 10 + sd};
    for (int k = 0; k < 3; k++) {
      if (k == 1) {   // y row: gripper entries are structurally zero
        static_for<6>([&](auto A_) { constexpr int a = A_;
          const real ja = Dc[sd] * Jc[sd][1][a];
          g0[a] += ja * arefc[sd][1];
          static_for<a + 1>([&](auto B_) { constexpr int b = B_; H0[tri(a, b)] += ja * Jc[sd][1][b]; }); });
      } else {
        static_for<9>([&](auto A_) { constexpr int a = A_;
          const real ja = Dc[sd] * Jc[sd][k][a];
          g0[idx[a]] += ja * arefc[sd][k];
          static_for<a + 1>([&](auto B_) { constexpr int b = B_; H0[tri(idx[a], idx[b])] += ja * Jc[sd][k][b]; }); });
      }
    }
  });
  H0[tri(6, 6)] += Dj; H0[tri(8, 8)] += Dj; H0[tri(8, 6)] -= Dj;
  g0[6] += Dj * arefj; g0[8] -= Dj * arefj;

  // ---- Newton iterations over the limit rows' active set, exact line search               (mj_fwdConstraint)
  real a[NB];
  static_for<NB>([&](auto I) { constexpr int i = I; a[i] = S.warm[i]; });
  bool act[10];
  static_for<10>([&](auto I) { constexpr int j = I; act[j] = (sgl[j] != 0) && (sgl[j] * a[j] - arefl[j] < 0); });
  real L[NB * (NB + 1) / 2];
  for (int it = 0; it < 50; it++) {
    real x[NB];
    for (int k = 0; k < NB * (NB + 1) / 2; k++) L[k] = H0[k];
    static_for<NB>([&](auto I) { constexpr int i = I; x[i] = g0[i]; });
    static_for<10>([&](auto I) { constexpr int j = I;
      if (act[j]) { L[tri(j, j)] += Dl[j]; x[j] += sgl[j] * Dl[j] * arefl[j]; } });
    chol_factor<NB>(L);
    chol_solve<NB>(L, x);
    if (!any_limit) { static_for<NB>([&](auto I) { constexpr int i = I; a[i] = x[i]; }); break; }
    // direction p = x - a; phi'(alpha) = (alpha - 1) p^T H p on the first piece
    real p[NB], Hp[NB];
    static_for<NB>([&](auto I) { constexpr int i = I; p[i] = x[i] - a[i]; });
    static_for<NB>([&](auto I) { constexpr int i = I; real s = 0;
      static_for<NB>([&](auto Jj) { constexpr int j = Jj; s += H0[tri(i, j)] * p[j]; }); Hp[i] = s; });
    real slope = 0;
    static_for<NB>([&](auto I) { constexpr int i = I; slope += p[i] * Hp[i]; });
    static_for<10>([&](auto I) { constexpr int j = I; if (act[j]) slope += Dl[j] * p[j] * p[j]; });
    real val = -slope, alpha = 0;
    bool crossed = false;
    // breakpoints alpha_j = -r_j / (sg_j p_j) of the existing rows, visited in increasing order
    real bp[10];
    static_for<10>([&](auto I) { constexpr int j = I;
      real rj = sgl[j] * a[j] - arefl[j], dj = sgl[j] * p[j];
      real al_ = (sgl[j] != 0 && dj != 0) ? -rj / dj : -1;
      bp[j] = al_ > 0 ? al_ : INFINITY; });
    for (int step = 0; step <= 10; step++) {
      real nxt = INFINITY; int jn = -1;
      static_for<10>([&](auto I) { constexpr int j = I; if (bp[j] < nxt) { nxt = bp[j]; jn = j; } });
      if (slope > 0 && val + slope * (nxt - alpha) >= 0) { alpha = alpha - val / slope; break; }
      if (jn < 0) { alpha = 1; break; }
      val += slope * (nxt - alpha); alpha = nxt;
      static_for<10>([&](auto I) { constexpr int j = I;
        if (j == jn) {
          real rj = sgl[j] * a[j] - arefl[j];
          if (rj < 0) slope -= Dl[j] * p[j] * p[j]; else slope += Dl[j] * p[j] * p[j];
          bp[j] = INFINITY;
        } });
      crossed = true;
    }
    static_for<NB>([&](auto I) { constexpr int i = I; a[i] += alpha * p[i]; });
    bool same = true;
    static_for<10>([&](auto I) { constexpr int j = I;
      bool now = (sgl[j] != 0) && (sgl[j] * a[j] - arefl[j] < 0);
      if (now != act[j]) same = false;
      act[j] = now; });
    if (!crossed && same) break;
  }

  // ---- constraint forces -> qfrc_constraint; P10 implicit-damping Euler                  (mj_Euler, mj_advance)
  real rhs[NB];
  static_for<NB>([&](auto I) { constexpr int i = I; rhs[i] = fs[i]; });
  static_for<2>([&](auto Sd) {
    constexpr int sd = Sd;
    constexpr int idx[9] = {0, 1, 2, 3, 4, 5, 6 + 2 * sd, 7 + 2 * sd, 10 + sd};
    for (int k = 0; k < 3; k++) {
      real jar = -arefc[sd][k];
      static_for<9>([&](auto A_) { constexpr int c = A_; jar += Jc[sd][k][c] * a[idx[c]]; });
      real force = -Dc[sd] * jar;
      static_for<9>([&](auto A_) { constexpr int c = A_; rhs[idx[c]] += Jc[sd][k][c] * force; });
    }
  });
  { real force = -Dj * (a[6] - a[8] - arefj); rhs[6] += force; rhs[8] -= force; }
  static_for<10>([&](auto I) { constexpr int j = I;
    if (sgl[j] != 0) { real jar = sgl[j] * a[j] - arefl[j]; if (jar < 0) rhs[j] += sgl[j] * (-Dl[j] * jar); } });
  static_for<NB>([&](auto I) { constexpr int i = I; M[tri(i, i)] += h * P->damping[i]; });
  chol_factor<NB>(M);
  chol_solve<NB>(M, rhs);
  static_for<NB>([&](auto I) { constexpr int i = I;
    S.qd[i] += h * rhs[i];
    S.q[i] += h * S.qd[i];
    S.warm[i] = a[i]; });
}

// ---------------------------------------------------------------------------------- world-frame arm kinematics
// EEF site pose and its 6x6 Jacobian at arm angles q6 (mj_kinematics + mj_jacSite for site EEF, P1/P11).
struct EefPose { real pos[3], mat[9], jacp[3][6], jacr[3][6]; };

MCG_DEV void eef_forward(const mcg_model* __restrict__ P, const real* q6, EefPose& E, bool want_jac) {
  real R[9], p[3], anchor[6][3], axis[6][3];
  for (int k = 0; k < 9; k++) R[k] = P->base_mat[k];
  for (int k = 0; k < 3; k++) p[k] = P->base_pos[k];
  static_for<6>([&](auto I) {
    constexpr int i = I; constexpr int K = AXK[i]; constexpr int A = (K + 1) % 3, B = (K + 2) % 3;
    const real* r = P->r[i];
    for (int k = 0; k < 3; k++) p[k] += R[3 * k] * r[0] + R[3 * k + 1] * r[1] + R[3 * k + 2] * r[2];
    for (int k = 0; k < 3; k++) { anchor[i][k] = p[k]; axis[i][k] = AXS[i] * R[3 * k + K]; }
    real s, c; sincos(AXS[i] * q6[i], &s, &c);
    for (int k = 0; k < 3; k++) {      // R <- R * Rot(e_K, theta): mixes columns A and B
      real ca = R[3 * k + A], cb = R[3 * k + B];
      R[3 * k + A] = c * ca + s * cb; R[3 * k + B] = -s * ca + c * cb;
    }
  });
  const real* se = P->site_eef;
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
