// mcg_cube.hpp -- the free cube of PickAndPlace: the lane-parallel part of the collision pass (the primitive geoms and the broad phase of
// the mesh geoms), the cube's pyramidal contact rows, its own (cube-alone) primal Newton solve, quaternion integration.  One env per lane.
//
// Replaces for this scene what mujoco.mj_step does for the `object0` body and its contacts
// (/root/reference/mycobotgym/envs/assets/mycobot280_main.xml:81,87,105-247,260-265; call sites mycobot.py:170,193): mj_collision (P4)
// over the primitive geoms here and over the mesh geoms' collision polytopes in mcg_mesh.hpp (one pair per wave, lane = vertex / face /
// edge); mj_makeConstraint / mj_projectConstraint for the pyramidal contacts (P5); for an environment in which nothing touches the robot,
// the cube's part of mj_fwdConstraint (P9); mj_Euler for a free joint (P10).  An environment in which a contact reaches the robot is
// solved by mcg_coop.hpp (one environment per 32 lanes).
//
// Contacts of one env live in LDS (runtime-indexed lists cannot live in registers): 10 slots per contact.
#pragma once

#include "mcg_dynamics.hpp"

namespace mcg {

constexpr int MAXCON = 16;           // list ENTRIES kept per env (a mesh's twin geoms: one entry of double weight); the oracle cuts its list at the
                                     // same entry (mco_model.maxentry).  MuJoCo has no cap; mcg_counters.contacts_dropped counts what this one cuts
constexpr int CON_STRIDE = 10;       // pos[3] n[3] dist D kterm type   (the tangents are re-derived from n: make_frame)
constexpr int CON_DIST = 6, CON_D = 7, CON_KTERM = 8, CON_TYPE = 9;
constexpr int PNP_LANES = 32;        // envs per wave in the PickAndPlace kernels: twice the LDS per env; a wave costs the
                                     // same with 32 or 64 active lanes (measured), and 8192 envs then cover all 256 CUs
constexpr int ROW_SLOTS = 144;       // the row area: line-search rows of the cube-alone solve, cooperative workspaces, parked inputs (mcg_coop.hpp)
constexpr int NJW = 12;              // joints whose world axis + anchor are kept for the contact rows: all twelve
constexpr int LDS_CON = LDS_SLOTS;
constexpr int LDS_POLY = LDS_CON + MAXCON * CON_STRIDE;     // two clip polygons of 8 x 2 in the first 32 slots (a quadrilateral clipped by
                                                            // four half-planes has at most 8 vertices); the other 32: exchange slots of the four-wave kernel
constexpr int LDS_ROW = LDS_POLY + 64;                      // per pyramid row: r0, dr (line search)
constexpr int LDS_ACT = LDS_ROW + ROW_SLOTS;                // per contact: active-row bit mask (as a double)
constexpr int LDS_WJ = LDS_ACT + MAXCON;                    // world axis + anchor of the twelve joints
constexpr int PNP_SLOTS = LDS_WJ + 6 * NJW;
typedef LaneScratchT<PNP_LANES> PnpScratch;

// Pair types (slot CON_TYPE of a list entry; rows of mcg_model.pair_tran).  The ground plane carries the table's parameters (both are
// default geoms).  NMESH collision polytopes (mycobotgym_amd/model/polytope.py: MESH_NAMES): links 1-6, flange, gripper base, right gear /
// finger link, left gear / finger link, right / left hinge link; polytope m rides on robot body mesh_body(m).
//   0 static-cube | 1, 2 right / left pad-cube | 3, 4 static-right / left pad | 5 + m static-mesh m (condim 3) | 19 + m mesh m-cube
constexpr int NMESH = 14;
enum { PAIR_TABLE_CUBE = 0, PAIR_PADR_CUBE = 1, PAIR_PADL_CUBE = 2, PAIR_TABLE_PADR = 3, PAIR_TABLE_PADL = 4, PAIR_STATIC_MESH0 = 5,
       PAIR_MESH0_CUBE = PAIR_STATIC_MESH0 + NMESH, NPAIR = PAIR_MESH0_CUBE + NMESH,
       PAR_TABLE_MESH = 5, PAR_MESH_CUBE = 6 };                  // rows of mcg_model.contact_par of the two mesh classes
MCG_DEV bool pair_has_cube(int type) { return type < PAIR_TABLE_PADR || type >= PAIR_MESH0_CUBE; }
MCG_DEV bool pair_mesh_static(int type) { return type >= PAIR_STATIC_MESH0 && type < PAIR_MESH0_CUBE; }
MCG_DEV bool pair_mesh_cube(int type) { return type >= PAIR_MESH0_CUBE; }
MCG_DEV int mesh_body(int m) { return m < 6 ? m : (m < 8 ? 5 : m - 2); }
// the robot body of a pair (-1: none): pads ride on the finger links (bodies 7, 9)
MCG_DEV int pair_robot_body(int type) {
  const int m = type >= PAIR_MESH0_CUBE ? type - PAIR_MESH0_CUBE : type - PAIR_STATIC_MESH0;
  return type == PAIR_TABLE_CUBE ? -1 : ((type == PAIR_PADR_CUBE || type == PAIR_TABLE_PADR) ? 7 : ((type == PAIR_PADL_CUBE || type == PAIR_TABLE_PADL) ? 9 : mesh_body(m)));
}
// joint j in the chain of robot body rb (the arm's six, then gear / finger right, gear / finger left, hinge right, hinge left)
MCG_DEV bool joint_in_chain(int j, int rb) {
  return rb >= 0 && (j < 6 ? j <= (rb < 5 ? rb : 5) : (j == 6 ? (rb == 6 || rb == 7) : (j == 8 ? (rb == 8 || rb == 9) : j == rb)));
}

// What the mesh phase (mcg_mesh.hpp) reads, columns of every lane.  MP_FRAME: the row area, free between barriers S1 and S2 (its solves run
// after S2): the world frame (rotation 9, origin 3) of each of the twelve robot bodies that carries a candidate mesh, parked by the wave
// whose broad phase found it -- ONE producer per body (M wave: bodies 0-3, RNE wave: 4, 5, cube wave: 6-11), so that a frame is one
// wave's arithmetic.  MP_*: twelve slots of their own behind the robot-side exchange slots.
constexpr int MP_FRAME = LDS_ROW;
static_assert(12 * NB <= ROW_SLOTS, "frames of the twelve bodies");
constexpr int MP_CUBE = PNP_SLOTS + NB;                          // the cube's position (published when a sub-step ENDS: the M / RNE waves' broad phase reads it
                                                                 // before S1b) and its normalised quaternion (written after S1, read in the mesh phase)
constexpr int MP_MASK = MP_CUBE + 7;                             // candidate pairs of the broad phases: M wave, RNE wave, cube wave (bit 3 m + o; o: ground, table, cube)
constexpr int MP_NCON = MP_MASK + 3, MP_DROP = MP_NCON + 1;     // the list's length (the mesh phase appends); contacts the cap cut there
constexpr int PNP_SLOTS_ALL = MP_DROP + 1;

struct Cube {
  real pos[3], quat[4], vel[6], warm[6];     // vel = world linear velocity, body-frame angular velocity (MuJoCo free joint)
};


// mju_makeFrame [RECALL]: tangents completing a unit normal
MCG_DEV void make_frame(const real* n, real* t1, real* t2) {
  const bool usey = (n[1] < 0.5 && n[1] > -0.5);
  real tmp[3] = {0.0, usey ? 1.0 : 0.0, usey ? 0.0 : 1.0};
  const real d = dot3(n, tmp);
  _Pragma("unroll") for (int k = 0; k < 3; k++) t1[k] = tmp[k] - d * n[k];
  const real il = 1.0 / sqrt(dot3(t1, t1));
  _Pragma("unroll") for (int k = 0; k < 3; k++) t1[k] *= il;
  cross(n, t1, t2);
}

template <class LS>
struct ContactList {
  const LS S;
  int n;              // entries stored
  int ndrop = 0;      // contacts the cap of MAXCON entries cut off
  MCG_DEV void add(const real* pos, const real* normal, real dist, int type) {
    const bool hit = dist < 0, ok = hit && n < MAXCON;
    if (ok) {                           // plain LDS stores of live registers (no value is merged across this branch)
      const int b = LDS_CON + n * CON_STRIDE;
      _Pragma("unroll") for (int k = 0; k < 3; k++) { S.st(b + k, pos[k]); S.st(b + 3 + k, normal[k]); }
      S.st(b + CON_DIST, dist); S.st(b + CON_D, 1.0); S.st(b + CON_TYPE, (real)type);       // slot CON_D: multiplicity until the solver numbers turn it into D
    }
    n += sel(ok, 1, 0);
    ndrop += sel(hit && !ok, 1, 0);
  }
};

// mjc_PlaneBox restated: every box vertex below the plane z = 0 of the world (the scene's only plane)
template <class LS>
MCG_DEV void ground_box(ContactList<LS>& CL, const real* pb, const real* Rb, const real* hb, int type) {
  const real n[3] = {0, 0, 1};
  _Pragma("unroll") for (int v = 0; v < 8; v++) {
    const real lx = (v & 1) ? hb[0] : -hb[0], ly = (v & 2) ? hb[1] : -hb[1], lz = sel((v & 4), hb[2], -hb[2]);
    real w[3];
    _Pragma("unroll") for (int k = 0; k < 3; k++) w[k] = pb[k] + Rb[3*k]*lx + Rb[3*k+1]*ly + Rb[3*k+2]*lz;
    const real dist = w[2];
    const real pos[3] = {w[0], w[1], w[2] - 0.5 * dist};
    CL.add(pos, n, dist, type);
  }
}

// mjc_BoxBox restated by behaviour: separating-axis test over 15 axes, then face clipping (<= 8 points) or one
// edge-edge point; position = midpoint between the surfaces, normal from box A to box B, dist < 0.
// A, B: rotation matrices row-major (world <- box).  Lanes with `live == false` do nothing.
static constexpr real EDGE_MIN_SIN = 1e-6;      // edges closer to parallel than this make no edge-edge axis: the face axes cover them
template <class LS>
MCG_DEV void box_box(ContactList<LS>& CL, bool live, const real* pa, const real* Ra, const real* ha,
                     const real* pb, const real* Rb, const real* hb, int type) {
  const LS& S = CL.S;
  real A[3][3], B[3][3], p[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  _Pragma("unroll") for (int k = 0; k < 3; k++) for (int r = 0; r < 3; r++) { A[k][r] = Ra[3*r + k]; B[k][r] = Rb[3*r + k]; }
  real Cm[3][3], Q[3][3];
  _Pragma("unroll") for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Cm[i][j] = dot3(A[i], B[j]); Q[i][j] = fabs(Cm[i][j]); }
  const real pA[3] = {dot3(A[0], p), dot3(A[1], p), dot3(A[2], p)};
  const real pB[3] = {dot3(B[0], p), dot3(B[1], p), dot3(B[2], p)};
  real best = -INFINITY; int code = -1; real nrm[3] = {0, 0, 0}; bool invert = false; bool sep = !live;
  _Pragma("unroll") for (int i = 0; i < 3; i++) {
    const real s = fabs(pA[i]) - (ha[i] + hb[0]*Q[i][0] + hb[1]*Q[i][1] + hb[2]*Q[i][2]);
    sep = sep || (s > 0);
    const bool tk = s > best;
    best = sel(tk, s, best); code = sel(tk, i, code); invert = sel(tk, (pA[i] < 0), invert);
    _Pragma("unroll") for (int k = 0; k < 3; k++) nrm[k] = sel(tk, A[i][k], nrm[k]);
  }
  _Pragma("unroll") for (int j = 0; j < 3; j++) {
    const real s = fabs(pB[j]) - (hb[j] + ha[0]*Q[0][j] + ha[1]*Q[1][j] + ha[2]*Q[2][j]);
    sep = sep || (s > 0);
    const bool tk = s > best;
    best = sel(tk, s, best); code = sel(tk, 3 + j, code); invert = sel(tk, (pB[j] < 0), invert);
    _Pragma("unroll") for (int k = 0; k < 3; k++) nrm[k] = sel(tk, B[j][k], nrm[k]);
  }
  _Pragma("unroll") for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    const real expr = pA[i2]*Cm[i1][j] - pA[i1]*Cm[i2][j];
    // |A_i x B_j| from the cross product itself: 1 - C^2 cancels to rounding noise of 1e-8 for parallel edges, and an axis made of
    // that noise then beat the face axes (a cube at rest lost its four table contacts for a sub-step)
    real L[3]; cross(A[i], B[j], L);
    const real len = sqrt(dot3(L, L));
    const bool valid = len >= EDGE_MIN_SIN;
    const real il = 1.0 / (valid ? len : 1.0);
    const real s = (fabs(expr) - (ha[i1]*Q[i2][j] + ha[i2]*Q[i1][j] + hb[j1]*Q[i][j2] + hb[j2]*Q[i][j1])) * il;
    sep = sep || (valid && s > 0);
    const bool tk = valid && (s * 1.05 > best);
    best = sel(tk, s, best); code = sel(tk, 6 + 3*i + j, code); invert = sel(tk, (expr < 0), invert);
    _Pragma("unroll") for (int k = 0; k < 3; k++) nrm[k] = sel(tk, L[k] * il, nrm[k]);
  }
  const bool hit = !sep && code >= 0;
  if (!__any(hit)) return;
  real normal[3];
  _Pragma("unroll") for (int k = 0; k < 3; k++) normal[k] = sel(invert, -nrm[k], nrm[k]);

  // ---- edge-edge (rare): one point
  if (__any(hit && code >= 6)) {
    const int ce = sel(code >= 6, code - 6, 0);
    const int i = ce / 3, j = ce % 3;
    real ea[3], eb[3], Ai[3], Bj[3];
    _Pragma("unroll") for (int k = 0; k < 3; k++) { ea[k] = pa[k]; eb[k] = pb[k]; Ai[k] = sel3(i, A[0][k], A[1][k], A[2][k]); Bj[k] = sel3(j, B[0][k], B[1][k], B[2][k]); }
    _Pragma("unroll") for (int a = 0; a < 3; a++) { const real sg = (a == i) ? 0.0 : (dot3(normal, A[a]) > 0 ? 1.0 : -1.0); _Pragma("unroll") for (int k = 0; k < 3; k++) ea[k] += sg * ha[a] * A[a][k]; }
    _Pragma("unroll") for (int b = 0; b < 3; b++) { const real sg = (b == j) ? 0.0 : (dot3(normal, B[b]) > 0 ? -1.0 : 1.0); _Pragma("unroll") for (int k = 0; k < 3; k++) eb[k] += sg * hb[b] * B[b][k]; }
    const real w[3] = {eb[0] - ea[0], eb[1] - ea[1], eb[2] - ea[2]};
    const real uaub = dot3(Ai, Bj), q1 = dot3(Ai, w), q2 = -dot3(Bj, w), dd = 1 - uaub*uaub;
    const real s = dd <= 1e-12 ? 0.0 : (q1 + uaub*q2) / dd, t = dd <= 1e-12 ? 0.0 : (uaub*q1 + q2) / dd;
    real pos[3];
    _Pragma("unroll") for (int k = 0; k < 3; k++) pos[k] = 0.5 * ((ea[k] + s*Ai[k]) + (eb[k] + t*Bj[k]));
    CL.add(pos, normal, (hit && code >= 6) ? best : 1.0, type);
  }
  if (!__any(hit && code < 6)) return;

  // ---- face contact: clip the incident face against the reference face
  const bool face = hit && code < 6;
  const bool refA = code < 3;
  const int ax = sel(face, code % 3, 0);
  real Rr[3][3], Ri[3][3], pr[3], pi[3], hr[3], hi[3], n2[3];
  _Pragma("unroll") for (int k = 0; k < 3; k++) {
    _Pragma("unroll") for (int r = 0; r < 3; r++) { Rr[k][r] = sel(refA, A[k][r], B[k][r]); Ri[k][r] = sel(refA, B[k][r], A[k][r]); }
    pr[k] = sel(refA, pa[k], pb[k]); pi[k] = sel(refA, pb[k], pa[k]); hr[k] = sel(refA, ha[k], hb[k]); hi[k] = sel(refA, hb[k], ha[k]);
    n2[k] = sel(refA, normal[k], -normal[k]);
  }
  const real nr[3] = {dot3(n2, Ri[0]), dot3(n2, Ri[1]), dot3(n2, Ri[2])};
  const int lan = fabs(nr[0]) > fabs(nr[1]) ? (fabs(nr[0]) > fabs(nr[2]) ? 0 : 2) : (fabs(nr[1]) > fabs(nr[2]) ? 1 : 2);
  const int a1 = (lan + 1) % 3, a2 = (lan + 2) % 3, c1 = (ax + 1) % 3, c2 = (ax + 2) % 3;
  auto pick = [](const real (*M)[3], int idx, real* out) { _Pragma("unroll") for (int k = 0; k < 3; k++) out[k] = sel3(idx, M[0][k], M[1][k], M[2][k]); };
  auto pickv = [](const real* v, int idx) { return sel3(idx, v[0], v[1], v[2]); };
  real Rilan[3], Ria1[3], Ria2[3], Rrc1[3], Rrc2[3];
  pick(Ri, lan, Rilan); pick(Ri, a1, Ria1); pick(Ri, a2, Ria2); pick(Rr, c1, Rrc1); pick(Rr, c2, Rrc2);
  const real hilan = pickv(hi, lan), hia1 = pickv(hi, a1), hia2 = pickv(hi, a2), hrax = pickv(hr, ax);
  real center[3];
  _Pragma("unroll") for (int k = 0; k < 3; k++) center[k] = pi[k] - pr[k] + (pickv(nr, lan) < 0 ? hilan : -hilan) * Rilan[k];
  const real cx = dot3(center, Rrc1), cy = dot3(center, Rrc2);
  const real m11 = dot3(Rrc1, Ria1), m12 = dot3(Rrc1, Ria2), m21 = dot3(Rrc2, Ria1), m22 = dot3(Rrc2, Ria2);
  const real k1 = m11*hia1, k2 = m21*hia1, k3 = m12*hia2, k4 = m22*hia2;
  const real rect[2] = {pickv(hr, c1), pickv(hr, c2)};
  // polygons in LDS: P at LDS_POLY + 2 v + {0,1}, T at LDS_POLY + 16 + 2 v + {0,1}
  S.st(LDS_POLY + 0, cx - k1 - k3); S.st(LDS_POLY + 1, cy - k2 - k4);
  S.st(LDS_POLY + 2, cx - k1 + k3); S.st(LDS_POLY + 3, cy - k2 + k4);
  S.st(LDS_POLY + 4, cx + k1 + k3); S.st(LDS_POLY + 5, cy + k2 + k4);
  S.st(LDS_POLY + 6, cx + k1 - k3); S.st(LDS_POLY + 7, cy + k2 - k4);
  int np = sel(face, 4, 0);
  int src = LDS_POLY, dst = LDS_POLY + 16;
  // Sutherland-Hodgman keeps a polygon that lies strictly inside all four limits as it is (same vertices, same order):
  // the usual case of the cube on the table needs no clipping pass at all.
  const bool inside = fabs(cx) + fabs(k1) + fabs(k3) < rect[0] && fabs(cy) + fabs(k2) + fabs(k4) < rect[1];
  if (__any(face && !inside))
  _Pragma("unroll") for (int dir = 0; dir < 2; dir++) for (int sgn = -1; sgn <= 1; sgn += 2) {
    int nq = 0;
    const real lim = sel(dir == 0, rect[0], rect[1]);
    for (int v = 0; __any(v < np); v++) {
      const bool on = v < np;
      const int vn = sel((v + 1 < np), v + 1, 0);
      const real Pd = S.ld(src + 2*v + dir), Po = S.ld(src + 2*v + 1 - dir);
      const real Nd = S.ld(src + 2*vn + dir), No = S.ld(src + 2*vn + 1 - dir);
      const bool inP = sgn * Pd < lim, inN = sgn * Nd < lim;
      if (on && inP && nq < 8) { S.st(dst + 2*nq + dir, Pd); S.st(dst + 2*nq + 1 - dir, Po); }
      nq += sel((on && inP && nq < 8), 1, 0);
      const real tt = (sgn * lim - Pd) / (Nd - Pd);
      if (on && (inP != inN) && nq < 8) { S.st(dst + 2*nq + dir, sgn * lim); S.st(dst + 2*nq + 1 - dir, Po + tt * (No - Po)); }
      nq += (on && (inP != inN) && nq < 8) ? 1 : 0;
    }
    np = nq;
    const int tswap = src; src = dst; dst = tswap;
  }
  const real det1 = 1.0 / (m11*m22 - m12*m21);
  const real im11 = m22*det1, im12 = -m12*det1, im21 = -m21*det1, im22 = m11*det1;
  int kept = 0;
  for (int v = 0; __any(v < np); v++) {
    const bool on = (v < np) && (kept < 8);
    const real qx = S.ld(src + 2*v) - cx, qy = S.ld(src + 2*v + 1) - cy;
    const real u1 = im11*qx + im12*qy, u2 = im21*qx + im22*qy;
    real pt[3];
    _Pragma("unroll") for (int k = 0; k < 3; k++) pt[k] = center[k] + u1*Ria1[k] + u2*Ria2[k];
    const real depth = hrax - dot3(n2, pt);
    real pos[3];
    _Pragma("unroll") for (int k = 0; k < 3; k++) pos[k] = pr[k] + pt[k] + 0.5*depth*n2[k];
    const bool take = on && (depth > 0);
    CL.add(pos, normal, take ? -depth : 1.0, type);
    kept += sel(take, 1, 0);
  }
}

// ------------------------------------------------------------------------------------------------ cube sub-step
// Row vectors of a contact in the cube's 6 dofs: J = [dir ; Rc^T (arm x dir)] for a translational direction,
// [0 ; Rc^T n] for torsion.  The cube is always geom2 of its pairs, so the cube side enters with +.
struct CubeRows { real Jn[6], J1[6], J2[6], Jt[6]; };

template <class LS>
MCG_DEV void cube_rows(const LS& S, int c, const real* Rc, const real* cpos, CubeRows& R) {
  const int b = LDS_CON + c * CON_STRIDE;
  real pos[3], n[3], t1[3], t2[3], arm[3], x[3];
  _Pragma("unroll") for (int k = 0; k < 3; k++) { pos[k] = S.ld(b + k); n[k] = S.ld(b + 3 + k); arm[k] = pos[k] - cpos[k]; }
  make_frame(n, t1, t2);
  auto fill = [&](const real* d, real* J) {
    cross(arm, d, x);
    _Pragma("unroll") for (int k = 0; k < 3; k++) { J[k] = d[k]; J[3 + k] = Rc[k]*x[0] + Rc[3 + k]*x[1] + Rc[6 + k]*x[2]; }
  };
  fill(n, R.Jn); fill(t1, R.J1); fill(t2, R.J2);
  _Pragma("unroll") for (int k = 0; k < 3; k++) { R.Jt[k] = 0; R.Jt[3 + k] = Rc[k]*n[0] + Rc[3 + k]*n[1] + Rc[6 + k]*n[2]; }
}

// pyramid row r (0..5) of a contact: Jn + sign * mu * Jk
MCG_DEV void pyramid_row(const CubeRows& R, int r, const real* mu, real* j) {
  const int k = r >> 1; const real sg = sel((r & 1), -1.0, 1.0);
  const real m = sg * (sel3(k, mu[0], mu[1], mu[2]));
  _Pragma("unroll") for (int d = 0; d < 6; d++) { const real jk = sel3(k, R.J1[d], R.J2[d], R.Jt[d]); j[d] = R.Jn[d] + m * jk; }
}

// ---- broad phase of the mesh geoms (lane = env).  Every mesh carries the bounding box of its collision polytope in the frame of the body
// it rides on (mcg_model.mesh_box: centre, half extents).  All three tests are CONSERVATIVE -- the box contains the polytope, a subset of
// the separating axes is tried -- so a pair they reject is separated; a pair they pass goes to the exact narrow phase (mcg_mesh.hpp),
// which decides.  Returns the pair's candidate bits: 1 ground, 2 table, 4 cube.
// CUBE_BOX: the caller knows the cube's attitude (rows of Rc = world <- cube, half sizes hc): the cube's three face axes and the box's three
// instead of the cube's bounding sphere (a held cube sits 2 mm from both finger links: the sphere passed 2.0 pairs per grasping
// environment to the narrow phase, the six axes pass 0.5)
template <bool CUBE_BOX = false>
MCG_DEV int mesh_broad(ModelPtr Pm, int m, const real* R, const real* p, const real* tp, const real* th, bool statics, bool cube, const real* cpos, real crad,
                       const real* Rc = nullptr, const real* hc = nullptr) {
  ModelPtr H = launder(Pm);
  real bx[6]; ldc<6>(H->mesh_box[m], bx);
  real c[3], e[3];                                                      // the box's centre; its half extents along the world axes
  _Pragma("unroll") for (int r = 0; r < 3; r++) {
    c[r] = p[r] + R[3*r]*bx[0] + R[3*r+1]*bx[1] + R[3*r+2]*bx[2];
    e[r] = fabs(R[3*r])*bx[3] + fabs(R[3*r+1])*bx[4] + fabs(R[3*r+2])*bx[5];
  }
  // (per-lane selects, no lane-divergent branch: see the compiler hazard in mcg_dynamics.hpp)
  const bool ground = c[2] - e[2] < 0;                                  // ground: the box's lowest point
  bool table = true;                                                    // table: its three face axes, then the box's own three
  _Pragma("unroll") for (int r = 0; r < 3; r++) table = table && !(fabs(c[r] - tp[r]) > th[r] + e[r]);
  bool near = true;                                                     // the cube against the box
  const real t[3] = {cpos[0] - c[0], cpos[1] - c[1], cpos[2] - c[2]};
  _Pragma("unroll") for (int k = 0; k < 3; k++) {
    const real rel = R[k]*(c[0] - tp[0]) + R[3 + k]*(c[1] - tp[1]) + R[6 + k]*(c[2] - tp[2]);
    const real rad = fabs(R[k])*th[0] + fabs(R[3 + k])*th[1] + fabs(R[6 + k])*th[2];
    table = table && !(fabs(rel) > bx[3 + k] + rad);
    const real tk = R[k]*t[0] + R[3 + k]*t[1] + R[6 + k]*t[2];           // along the box's axis k
    if constexpr (CUBE_BOX) {
      real rb = 0;
      _Pragma("unroll") for (int a = 0; a < 3; a++) rb += hc[a] * fabs(R[k]*Rc[a] + R[3 + k]*Rc[3 + a] + R[6 + k]*Rc[6 + a]);
      near = near && !(fabs(tk) > bx[3 + k] + rb);
      const real ta = Rc[k]*t[0] + Rc[3 + k]*t[1] + Rc[6 + k]*t[2];    // along the cube's axis k
      real ra = 0;
      _Pragma("unroll") for (int a = 0; a < 3; a++) ra += bx[3 + a] * fabs(R[a]*Rc[k] + R[3 + a]*Rc[3 + k] + R[6 + a]*Rc[6 + k]);
      near = near && !(fabs(ta) > hc[k] + ra);
    } else near = near && !(fabs(tk) > bx[3 + k] + crad);
  }
  return sel(statics && ground, 1, 0) | sel(statics && table, 2, 0) | sel(cube && near, 4, 0);
}

// ---- the arm meshes' broad phase against the table and the ground, on the M / RNE waves of the four-wave kernel (they hold the arm's
// sines / cosines anyway and are done long before the cube wave: round 3 had them run the whole 16-axis test here, ~27 k clocks; the
// exact narrow phase now runs one pair per wave behind this filter).  M: meshes 0-3, RNE: 4-7 (links 5, 6, flange, gripper base).
MCG_DEV void park_frame(const PnpScratch& S, int body, const real* R, const real* p, bool doit) {
  if (__any(doit)) {                                               // wave-uniform
    if (doit) {                                                    // plain LDS stores of live registers
      _Pragma("unroll") for (int k = 0; k < 9; k++) S.st(MP_FRAME + body * 12 + k, R[k]);
      _Pragma("unroll") for (int k = 0; k < 3; k++) S.st(MP_FRAME + body * 12 + 9 + k, p[k]);
    }
  }
}
template <int P0, int P1, class LS>
MCG_DEV void arm_broad_stage(ModelPtr Pm, const LS S, const real* sn, const real* cs, int mask_slot) {
  ModelPtr Q = launder(Pm);
  real tp[3], th[3], hc[3]; ldc<3>(Q->table_pos, tp); ldc<3>(Q->table_half, th); ldc<3>(Q->cube_half, hc);
  const real crad = sqrt(dot3(hc, hc));
  const real cpos[3] = {S.ld(MP_CUBE), S.ld(MP_CUBE + 1), S.ld(MP_CUBE + 2)};      // (published when the last sub-step ended)
  real R[9], p[3];
  _Pragma("unroll") for (int k = 0; k < 9; k++) R[k] = Q->base_mat[k];
  _Pragma("unroll") for (int k = 0; k < 3; k++) p[k] = Q->base_pos[k];
  int bits = 0;
  constexpr int LASTB = P1 - 1 < 5 ? P1 - 1 : 5;                 // polytope m rides on arm body min(m, 5)
  static_for<LASTB + 1>([&](auto I) { constexpr int i = I; constexpr int K = AXK[i]; constexpr int A = (K + 1) % 3, B = (K + 2) % 3;
    real r[3]; ldc<3>(Q->body[i].r, r);
    const real rad = Q->body[i].hull_rad;
    _Pragma("unroll") for (int k = 0; k < 3; k++) p[k] += R[3*k]*r[0] + R[3*k+1]*r[1] + R[3*k+2]*r[2];
    const real sn_ = sn[i], cs_ = cs[i];                         // of AXS[i] * q[i]
    _Pragma("unroll") for (int k = 0; k < 3; k++) {
      const real ca = R[3*k + A], cb = R[3*k + B];
      R[3*k + A] = cs_ * ca + sn_ * cb; R[3*k + B] = -sn_ * ca + cs_ * cb;
    }
    if constexpr (i >= (P0 < 5 ? P0 : 5)) {
      const real dx = p[0] - cpos[0], dy = p[1] - cpos[1], dz = p[2] - cpos[2];
      const bool ncube = dx*dx + dy*dy + dz*dz < (rad + crad) * (rad + crad);      // the body's bounding sphere reaches the cube's
      const bool nstat = p[2] - rad < tp[2] + th[2];                               // ... the table top's height
      if (__any(ncube || nstat)) {                                // wave-uniform
        int bb = 0;
        if constexpr (i < 5) bb = mesh_broad(Pm, i, R, p, tp, th, nstat, ncube, cpos, crad) << (3 * i);
        else { static_for<P1 - (P0 > 5 ? P0 : 5)>([&](auto Mm) { constexpr int m = (P0 > 5 ? P0 : 5) + Mm; bb |= mesh_broad(Pm, m, R, p, tp, th, nstat, ncube, cpos, crad) << (3 * m); }); }
        park_frame(S, i, R, p, bb != 0);
        bits |= bb;
      }
    } });
  S.st(mask_slot, (real)bits);
}

// ---- P5 in the four-wave kernel: the solver numbers of list positions r, r + 3, r + 6, ... of every lane -- the cube wave takes r = 0, the
// M wave 1, the RNE wave 2, between barriers S1c (the list is complete: the mesh phase has appended its contacts) and S2.  The
// contact-at-a-time pass of the cube wave alone sat on the workgroup's critical path with three waves waiting
// (profiles/r03/ab_critical_path_probes.log).  The independent contacts of a wave's share are separate chains in one stretch of code.
constexpr int NUM_GROUP = 2;                                      // list positions per stretch of code: r0 + 3 u, u = 2 g .. 2 g + 1
template <class LS>
MCG_DEV void solver_numbers_group(ModelPtr Pm, const LS S, real dr1, int r0, int ncon) {
  ModelPtr Q = launder(Pm);
  real dist[NUM_GROUP], mult[NUM_GROUP]; int type[NUM_GROUP]; bool in[NUM_GROUP];
  _Pragma("unroll") for (int u = 0; u < NUM_GROUP; u++) {
    const int c = r0 + 3 * u; in[u] = c < ncon;
    const int b = LDS_CON + sel(c < MAXCON, c, MAXCON - 1) * CON_STRIDE;
    dist[u] = S.ld(b + CON_DIST); mult[u] = S.ld(b + CON_D); type[u] = sel(in[u], (int)S.ld(b + CON_TYPE), 0);
  }
  bool amc = false, atp = false, ams = false;
  _Pragma("unroll") for (int u = 0; u < NUM_GROUP; u++) {
    amc = amc || pair_mesh_cube(type[u]); atp = atp || type[u] == PAIR_TABLE_PADR || type[u] == PAIR_TABLE_PADL; ams = ams || pair_mesh_static(type[u]);
  }
  amc = __any(amc); atp = __any(atp); ams = __any(ams);
  const real ft = Q->geom_friction0[0], fp = Q->geom_friction0[1] * dr1, fcb = Q->geom_friction0[2] * dr1;
  const real mu_tc0 = fmax(ft, fcb), mu_pc0 = fmax(fp, fcb), mu_tp0 = fmax(ft, fp);
  real par_t[10], par_p[10];
  ldc<10>(Q->contact_par[PAIR_TABLE_CUBE], par_t); ldc<10>(Q->contact_par[PAIR_PADR_CUBE], par_p);
  const real rpy = Q->contact_rpy;
  real imp[NUM_GROUP], kk[NUM_GROUP], m0[NUM_GROUP], tran[NUM_GROUP];
  _Pragma("unroll") for (int u = 0; u < NUM_GROUP; u++) {
    const bool padcube = type[u] == PAIR_PADR_CUBE || type[u] == PAIR_PADL_CUBE;
    imp[u] = sel(padcube, impedance(par_p, dist[u]), impedance(par_t, dist[u]));
    kk[u] = sel(padcube, par_p[0], par_t[0]);
    m0[u] = sel(padcube, mu_pc0, mu_tc0);
    { const CRealPtr pt = &Pm->pair_tran[0]; tran[u] = pt[type[u]]; }      // (a per-lane index: a vector load from the model block)
  }
  if (amc) {
    ModelPtr Qb = launder(Pm);
    real par_mc[10]; ldc<10>(Qb->contact_par[PAR_MESH_CUBE], par_mc);
    const real mu_mc0 = fmax(Qb->mesh_fric, fcb);
    _Pragma("unroll") for (int u = 0; u < NUM_GROUP; u++) {
      const bool mc = pair_mesh_cube(type[u]);
      imp[u] = sel(mc, impedance(par_mc, dist[u]), imp[u]); kk[u] = sel(mc, par_mc[0], kk[u]); m0[u] = sel(mc, mu_mc0, m0[u]);
    }
  }
  if (atp) {
    ModelPtr Qb = launder(Pm);
    real par_tp[10]; ldc<10>(Qb->contact_par[PAIR_TABLE_PADR], par_tp);
    _Pragma("unroll") for (int u = 0; u < NUM_GROUP; u++) {
      const bool tabp = type[u] == PAIR_TABLE_PADR || type[u] == PAIR_TABLE_PADL;
      imp[u] = sel(tabp, impedance(par_tp, dist[u]), imp[u]); kk[u] = sel(tabp, par_tp[0], kk[u]); m0[u] = sel(tabp, mu_tp0, m0[u]);
    }
  }
  if (ams) {
    ModelPtr Qb = launder(Pm);
    real par_tl[11]; ldc<11>(Qb->contact_par[PAR_TABLE_MESH], par_tl);
    _Pragma("unroll") for (int u = 0; u < NUM_GROUP; u++) {
      const bool ms = pair_mesh_static(type[u]);
      imp[u] = sel(ms, impedance(par_tl, dist[u]), imp[u]); kk[u] = sel(ms, par_tl[0], kk[u]); m0[u] = sel(ms, par_tl[10], m0[u]);
    }
  }
  _Pragma("unroll") for (int u = 0; u < NUM_GROUP; u++) {
    const real Rn = fmax(MINVAL, (1 - imp[u]) * tran[u] * (1 + m0[u]*m0[u]) / imp[u]);
    const real Rpy = fmax(MINVAL, rpy * m0[u]*m0[u] * Rn);
    if (in[u]) { const int b = LDS_CON + (r0 + 3 * u) * CON_STRIDE; S.st(b + CON_D, mult[u] / Rpy); S.st(b + CON_KTERM, kk[u] * imp[u] * dist[u]); }
  }
}
template <class LS>
MCG_DEV void solver_numbers_share(ModelPtr Pm, const LS S, real dr1, int r0) {
  const int ncon = (int)S.ld(MP_NCON);
  for (int g = 0; __any(r0 + 3 * NUM_GROUP * g < ncon); g++)                      // wave-uniform: as many groups as the longest list needs
    solver_numbers_group(Pm, S, dr1, r0 + 3 * NUM_GROUP * g, ncon);
}

// The cube and its contacts for one sub-step: prepared before the robot's Newton solve, finished after it.
template <class LS>
struct CubeSys {
  const LS S; Cube Cb; real dr[2];             // by value: a reference into the env struct pins that struct in memory
                                               // (the model pointer is passed in: it must stay a scalar register)
  unsigned long long pm_bits;                  // the model pointer's bits, for stages reached through the robot's hook
  unsigned long long* cnt = nullptr;           // mcg_counters on the device (slot 2: contacts dropped by the cap), or null
  MCG_DEV CubeSys(const LS s_, const Cube& c, const real* d) : S(s_), Cb(c), pm_bits(0) { dr[0] = d[0]; dr[1] = d[1]; }
  real h, Rc[9], Md[6], damp[6], fs[6];
  real B_tc, B_pc, B_tp, B_tl, B_mc, mu_tc[3], mu_pc[3], mu_tp[3], mu_tl[3], mu_mc[3];
  bool stat_on;                                // a contact between a static geom and the robot alone is in the list
  int ndropped = 0;                            // contacts of this pass that the cap cut off
  int cube_lo, cube_hi, c0 = 0;                // lowest / highest list position of a contact that involves the cube (hi -1: none); list offset of the cube-alone solve
  int ncon; bool any_pad, solved, touch[2];    // touch: this forward pass has a right / left pad-cube contact; any_pad: a contact reaches the robot
  real a_c[6];

  // ------------------------------------------------------------------------------------------------- prepare
  // Cheap quantities of the current state (rotation, inertia, smooth force, friction): re-derived by each stage that
  // needs them rather than kept live across the robot's pipeline (they would be spilled there).
  MCG_DEV void derive(ModelPtr Pm) {
    ModelPtr Q = launder(Pm);
    h = Q->timestep;
    quat_to_mat(Cb.quat, Rc);
    const real mass = Q->body[12].mass * dr[0];
    const real In[3] = {Q->body[12].inertia[0] * dr[0], Q->body[12].inertia[1] * dr[0], Q->body[12].inertia[2] * dr[0]};
    Md[0] = Md[1] = Md[2] = mass; Md[3] = In[0]; Md[4] = In[1]; Md[5] = In[2];
    ldc<6>(Q->cube_damping, damp);
    real gb[3]; ldc<3>(Q->gravity_base, gb);
    const real* w = Cb.vel + 3;
    const real Iw[3] = {In[0]*w[0], In[1]*w[1], In[2]*w[2]};
    real gyro[3]; cross(w, Iw, gyro);
    fs[0] = -damp[0]*Cb.vel[0]; fs[1] = -damp[1]*Cb.vel[1]; fs[2] = -damp[2]*Cb.vel[2] - mass * gb[2];
    _Pragma("unroll") for (int k = 0; k < 3; k++) fs[3 + k] = -damp[3 + k]*w[k] - gyro[k];
    // friction after domain randomisation: element-wise max of the (scaled) geom frictions
    const real ft = Q->geom_friction0[0], fp = Q->geom_friction0[1] * dr[1], fcb = Q->geom_friction0[2] * dr[1];
    mu_tc[0] = mu_tc[1] = fmax(ft, fcb); mu_tc[2] = Q->contact_par[PAIR_TABLE_CUBE][12];
    mu_pc[0] = mu_pc[1] = fmax(fp, fcb); mu_pc[2] = Q->contact_par[PAIR_PADR_CUBE][12];
    mu_tp[0] = mu_tp[1] = fmax(ft, fp); mu_tp[2] = Q->contact_par[PAIR_TABLE_PADR][12];
    B_tc = Q->contact_par[PAIR_TABLE_CUBE][1]; B_pc = Q->contact_par[PAIR_PADR_CUBE][1]; B_tp = Q->contact_par[PAIR_TABLE_PADR][1];
    mu_tl[0] = mu_tl[1] = Q->contact_par[PAR_TABLE_MESH][10]; mu_tl[2] = 0; B_tl = Q->contact_par[PAR_TABLE_MESH][1];      // condim 3: no torsional rows
    mu_mc[0] = mu_mc[1] = fmax(Q->mesh_fric, fcb); mu_mc[2] = Q->contact_par[PAR_MESH_CUBE][12]; B_mc = Q->contact_par[PAR_MESH_CUBE][1];
  }

  MCG_DEV ModelPtr model() const {                // wave-uniform pointer rebuilt as a scalar
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pm_bits), hi = __builtin_amdgcn_readfirstlane((unsigned)(pm_bits >> 32));
    return (ModelPtr)(((unsigned long long)hi << 32) | lo);
  }

  // what the completed list holds decides the routing: flags, the stretch of the list the cube-alone solve walks
  MCG_DEV void scan_list() {
    stat_on = false; cube_hi = -1; cube_lo = 0; any_pad = false; touch[0] = touch[1] = false;
    for (int c = 0; __any(c < ncon); c++) {
      const bool in = c < ncon;
      const int type = sel(in, (int)S.ld(LDS_CON + sel(c < MAXCON, c, MAXCON - 1) * CON_STRIDE + CON_TYPE), PAIR_TABLE_CUBE);
      any_pad = any_pad || type != PAIR_TABLE_CUBE;
      touch[0] = touch[0] || type == PAIR_PADR_CUBE; touch[1] = touch[1] || type == PAIR_PADL_CUBE;
      cube_lo = sel(in && pair_has_cube(type) && cube_hi < 0, c, cube_lo);
      cube_hi = sel(in && pair_has_cube(type), c, cube_hi);
      stat_on = stat_on || (in && !pair_has_cube(type));
    }
  }

  // ---- P4, the lane-parallel part: the primitive geoms' contacts and the candidate pairs of the mesh geoms.
  // qr: the robot's twelve joint angles.  MESHES: fill the mesh phase's slots (the cube's pose, this wave's candidate mask, the list's
  // length); ARM: this wave also runs the arm meshes' broad phase against the table / the ground (the four-wave kernel leaves that to
  // the M / RNE waves: arm_broad_stage).
  template <bool MESHES, bool ARM>
  MCG_DEV void collide_primitives(ModelPtr Pm, const real* qr) {
    pm_bits = (unsigned long long)Pm;
    {   // mj_kinematics normalises the stored quaternion
      const real nq = sqrt(Cb.quat[0]*Cb.quat[0] + Cb.quat[1]*Cb.quat[1] + Cb.quat[2]*Cb.quat[2] + Cb.quat[3]*Cb.quat[3]);
      const bool tiny = nq < MINVAL;
      _Pragma("unroll") for (int k = 0; k < 4; k++) Cb.quat[k] = sel(tiny, k == 0 ? 1.0 : 0.0, Cb.quat[k] / nq);
    }
    derive(Pm);
    ModelPtr Q = launder(Pm);
    solved = false;
    _Pragma("unroll") for (int k = 0; k < 6; k++) a_c[k] = Cb.warm[k];

    MCG_TICK2(ST_A_ENTRY);
    ContactList<LS> CL{S, 0};
    real hc[3]; ldc<3>(Q->cube_half, hc);
    real tp[3], th[3]; ldc<3>(Q->table_pos, tp); ldc<3>(Q->table_half, th);
    const real crad = sqrt(dot3(hc, hc));
    any_pad = false; touch[0] = touch[1] = false;
    long long mbits = 0;                               // candidate pairs: bit 3 m + o (o: 0 ground, 1 table, 2 cube)
    // world frames of the arm joints and of the gripper's six joints (mj_kinematics; the contact rows' twist columns: LDS_WJ)
    real Rs[2][9], pc[2][3], ph[2][3];                 // finger frames, pad centres and half sizes
    _Pragma("unroll") for (int sd = 0; sd < 2; sd++) {     // defined values for lanes / waves whose pads are not posed
      _Pragma("unroll") for (int k = 0; k < 9; k++) Rs[sd][k] = (k % 4 == 0) ? 1.0 : 0.0;
      pc[sd][0] = pc[sd][1] = 0.0; pc[sd][2] = 1.0; ph[sd][0] = ph[sd][1] = ph[sd][2] = 0.0;
    }
    bool reach, padlive;
    {
      const TrigC T = load_trig();
      real R[9], p[3];
      _Pragma("unroll") for (int k = 0; k < 9; k++) R[k] = Q->base_mat[k];
      _Pragma("unroll") for (int k = 0; k < 3; k++) p[k] = Q->base_pos[k];
      auto joint = [&](int slot, int K, int sg, const real* r, real ang, real* Rio, real* pio) {
        _Pragma("unroll") for (int k = 0; k < 3; k++) pio[k] += Rio[3*k]*r[0] + Rio[3*k+1]*r[1] + Rio[3*k+2]*r[2];
        _Pragma("unroll") for (int k = 0; k < 3; k++) { S.st(LDS_WJ + slot*6 + k, sg * Rio[3*k + K]); S.st(LDS_WJ + slot*6 + 3 + k, pio[k]); }
        real sn_, cs_; sincos_cw(T, sg * ang, sn_, cs_);
        const int A = (K + 1) % 3, B = (K + 2) % 3;
        _Pragma("unroll") for (int k = 0; k < 3; k++) {
          const real ca = Rio[3*k + A], cb = Rio[3*k + B];
          Rio[3*k + A] = cs_ * ca + sn_ * cb; Rio[3*k + B] = -sn_ * ca + cs_ * cb;
        }
      };
      // a mesh's candidate bits and, with a candidate, its body's frame for the mesh phase.  The gripper's parts here; the arm's in the
      // M / RNE waves (arm_broad_stage) unless ARM
      auto broad = [&](int m, int body, const real* Rio, const real* pio, bool statics, bool cube) {
        if constexpr (MESHES) {
          const int bb = mesh_broad<true>(Pm, m, Rio, pio, tp, th, statics, cube, Cb.pos, crad, Rc, hc);
          mbits |= (long long)bb << (3 * m);
          return bb != 0;
        } else return false;
      };
      static_for<6>([&](auto I) { constexpr int i = I; real r[3]; ldc<3>(Q->body[i].r, r);
        const real rad = Q->body[i].hull_rad;                             // comes with the same batch of scalar loads as r
        joint(i, AXK[i], AXS[i], r, qr[i], R, p);
        if constexpr (MESHES && ARM) {
          const real dx = p[0] - Cb.pos[0], dy = p[1] - Cb.pos[1], dz = p[2] - Cb.pos[2];
          const bool ncube = dx*dx + dy*dy + dz*dz < (rad + crad) * (rad + crad);      // the body's bounding sphere reaches the cube's
          const bool nstat = p[2] - rad < tp[2] + th[2];                               // ... the table top's height
          if (__any(ncube || nstat)) {
            bool any = broad(i, i, R, p, nstat, ncube);
            if constexpr (i == 5) { any = broad(6, 5, R, p, nstat, ncube) || any; any = broad(7, 5, R, p, nstat, ncube) || any; }
            park_frame(S, i, R, p, any);
          }
        } });
      MCG_TICK2(ST_A_G);
      const real dxe = p[0] - Cb.pos[0], dye = p[1] - Cb.pos[1], dze = p[2] - Cb.pos[2];
      reach = dxe*dxe + dye*dye + dze*dze < 0.2 * 0.2;                  // link6 origin within 20 cm of the cube
      // a pad's far corner is at most 0.16 m from the link6 origin (and so is every point of the gripper's links): the gripper can only
      // touch the table / the ground from within 0.17 m
      real dtab = 0;
      _Pragma("unroll") for (int k = 0; k < 3; k++) { const real e = fmax(fabs(p[k] - tp[k]) - th[k], 0.0); dtab = fma(e, e, dtab); }
      const bool nearstat = dtab < 0.17 * 0.17 || p[2] < 0.17;
      padlive = reach || nearstat;
      if (__any(padlive)) {
        static_for<2>([&](auto Sd) {
          constexpr int sd = Sd; constexpr int g = 6 + 2 * sd, f = 7 + 2 * sd;
          real ps[3];
          _Pragma("unroll") for (int k = 0; k < 9; k++) Rs[sd][k] = R[k];
          _Pragma("unroll") for (int k = 0; k < 3; k++) ps[k] = p[k];
          real r[3]; ldc<3>(Q->body[g].r, r); joint(g, 1, AXS[g], r, qr[g], Rs[sd], ps);
          park_frame(S, g, Rs[sd], ps, broad(8 + 2 * sd, g, Rs[sd], ps, nearstat, reach));      // gear link
          ldc<3>(Q->body[f].r, r); joint(f, 1, AXS[f], r, qr[f], Rs[sd], ps);
          park_frame(S, f, Rs[sd], ps, broad(9 + 2 * sd, f, Rs[sd], ps, nearstat, reach));      // finger link
          real pb[6]; ldc<6>(Q->pad_box[sd], pb);
          _Pragma("unroll") for (int k = 0; k < 3; k++) { pc[sd][k] = ps[k] + Rs[sd][3*k]*pb[0] + Rs[sd][3*k+1]*pb[1] + Rs[sd][3*k+2]*pb[2]; ph[sd][k] = pb[3 + k]; }
          // the hinge link of this side (its joint hangs on link6)
          real Rh[9], phg[3];
          _Pragma("unroll") for (int k = 0; k < 9; k++) Rh[k] = R[k];
          _Pragma("unroll") for (int k = 0; k < 3; k++) phg[k] = p[k];
          ldc<3>(Q->body[10 + sd].r, r); joint(10 + sd, 1, AXS[10 + sd], r, qr[10 + sd], Rh, phg);
          park_frame(S, 10 + sd, Rh, phg, broad(12 + sd, 10 + sd, Rh, phg, nearstat, reach));
        });
      }
    }
    // ground plane: pads, cube
    if (__any(padlive && (pc[0][2] < 0.02 || pc[1][2] < 0.02))) {
      static_for<2>([&](auto Sd) { constexpr int sd = Sd;
        const bool low = padlive && pc[sd][2] < 0.02;
        const real far[3] = {pc[sd][0], pc[sd][1], low ? pc[sd][2] : 1.0};
        ground_box(CL, far, Rs[sd], ph[sd], PAIR_TABLE_PADR + sd); });
    }
    if (__any(Cb.pos[2] < 0.05)) {
      const bool low = Cb.pos[2] < 0.05;
      const real far[3] = {Cb.pos[0], Cb.pos[1], low ? Cb.pos[2] : 1.0};
      ground_box(CL, far, Rc, hc, PAIR_TABLE_CUBE);
    }
    // table: pads, cube.  The table is a static axis-aligned box: its three face axes are separating axes of the SAT, so
    // "the pad's extent along one of them clears the table" skips the pair with the result the full test would give.
    const real Rt[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    MCG_TICK2(ST_A_TWIST);
    static_for<2>([&](auto Sd) { constexpr int sd = Sd;
      bool near = padlive;
      _Pragma("unroll") for (int k = 0; k < 3; k++) {
        const real ext = fabs(Rs[sd][3*k]) * ph[sd][0] + fabs(Rs[sd][3*k+1]) * ph[sd][1] + fabs(Rs[sd][3*k+2]) * ph[sd][2];
        near = near && !(fabs(pc[sd][k] - tp[k]) - (th[k] + ext) > 0);
      }
      if (__any(near)) box_box(CL, near, tp, Rt, th, pc[sd], Rs[sd], ph[sd], PAIR_TABLE_PADR + sd); });
    MCG_TICK2(ST_A_LOOP);
    {
      const real dx = Cb.pos[0] - tp[0], dy = Cb.pos[1] - tp[1], dz = Cb.pos[2] - tp[2];
      const real rs = sqrt(dot3(th, th)) + crad;
      const bool near = dx*dx + dy*dy + dz*dz <= rs*rs;
      if (__any(near)) box_box(CL, near, tp, Rt, th, Cb.pos, Rc, hc, PAIR_TABLE_CUBE);
    }
    // pads - cube
    if (__any(reach)) {
      static_for<2>([&](auto Sd) { constexpr int sd = Sd;
        const real dx = Cb.pos[0] - pc[sd][0], dy = Cb.pos[1] - pc[sd][1], dz = Cb.pos[2] - pc[sd][2];
        const real rs = sqrt(dot3(ph[sd], ph[sd])) + crad;
        const bool near = reach && (dx*dx + dy*dy + dz*dz <= rs*rs);
        if (__any(near)) box_box(CL, near, pc[sd], Rs[sd], ph[sd], Cb.pos, Rc, hc, PAIR_PADR_CUBE + sd); });
    }
    MCG_TICK2(ST_A_MAP);
    ncon = CL.n; ndropped = CL.ndrop;
    if constexpr (MESHES) {
      if constexpr (ARM) { _Pragma("unroll") for (int k = 0; k < 3; k++) S.st(MP_CUBE + k, Cb.pos[k]); }      // (else: published when the last sub-step ended)
      _Pragma("unroll") for (int k = 0; k < 4; k++) S.st(MP_CUBE + 3 + k, Cb.quat[k]);
      S.st(MP_MASK + 2, (real)mbits); S.st(MP_NCON, (real)ncon); S.st(MP_DROP, 0.0);
      if constexpr (ARM) { S.st(MP_MASK, 0.0); S.st(MP_MASK + 1, 0.0); }
    }
    MCG_TICK2(ST_A_STORE);
  }

  // ---- after the mesh phase (mcg_mesh.hpp) has appended the mesh geoms' contacts: the list as it now stands
  MCG_DEV void collect_list() {
    ncon = (int)S.ld(MP_NCON); ndropped += (int)S.ld(MP_DROP);
    if (__any(ndropped > 0)) { if (ndropped > 0 && cnt) atomicAdd(cnt + 2, (unsigned long long)ndropped); }      // MAXCON cut the list (MuJoCo has no such cap)
    scan_list();
#ifdef MCG_STAGE_CLOCKS
    { int mx = 0; for (int c = 0; __any(c < ncon); c++) mx = c + 1; if ((threadIdx.x & 63) == 0) atomicAdd(&g_stage_clocks[ST_COUNT + CN_CONTACTS], (unsigned long long)mx); }   // wave-max contacts
#endif
    // Lanes with fewer contacts than their wave-mates still walk the longer list with zero weights: give them clean
    // zeros to multiply (uninitialised LDS may hold NaN / inf, and 0 * NaN would poison the sums).  Only the contact entries here:
    // the line-search rows and the masks are cleared by the solves that use them (clean_rows).
    for (int c = 0; __any(c < ncon); c++) {
      if (c >= ncon) {
        for (int k = 0; k < CON_STRIDE; k++) S.st(LDS_CON + c * CON_STRIDE + k, 0.0);
      }
    }
    MCG_TICK2(ST_CUBE);
  }

  // identical geoms list entry c stands for (debug export)
  MCG_DEV real mult_of(int c) const {
    const int type = (int)S.ld(LDS_CON + c * CON_STRIDE + CON_TYPE);
    return (type >= PAIR_STATIC_MESH0) ? model()->mesh_mult : 1.0;
  }
  // rows of contact c in the cube's dofs / in the robot's dofs of the pad's side
  MCG_DEV void rows_cube(int c, CubeRows& R) const { cube_rows(S, c, Rc, Cb.pos, R); }
  // ------------------------------------------------------------------------------------- cube alone (no pad contact)
  // Two passes over the contact list per Newton iteration: (A) active mask (from the warm start on the first
  // iteration) + assembly of H and g; (B) consistency of the mask at the solution x, the constraint forces at x and the
  // line-search data.  When the mask is consistent -- the usual case -- x is the minimiser and B's forces are final.
  // zeros in the line-search rows and masks of the list positions this lane does not fill (see prepare()).  `skip`: the lane's column
  // of the row area is not this solve's to touch (four-wave kernel: the robot wave parks the coupled solve's inputs there).
  MCG_DEV void clean_rows(bool skip = false) const {
    for (int c = 0; __any(c < ncon); c++) {
      if (c >= ncon && !skip) {
        _Pragma("unroll") for (int k = 0; k < 12; k++) S.st(LDS_ROW + c * 12 + k, 0.0);
        S.st(LDS_ACT + c, 0.0);
      }
    }
  }
  // `skip`: this lane's env goes through the cooperative coupled solve instead; it walks the loops with an empty list and
  // stores nothing (its result is discarded).
  // A wave that holds static geom - robot contacts somewhere (rare; wave-uniform) runs the OFF = true instance: each lane's loops then
  // cover only the stretch of its list that holds the cube's contacts, positions c0 .. c0 + ncon - 1 (the robot's entries in front of
  // it -- arm meshes, pads on the ground -- would otherwise make the whole wave walk lists three times as long), and entries inside the
  // stretch that are not the cube's get D = 0.  The rows and masks in LDS are indexed by the position within the stretch.
  MCG_DEV void solve_alone(bool skip = false) {
    const int ncon_all = ncon;
    if (__any(stat_on && !skip)) {
      c0 = sel(cube_hi >= 0, cube_lo, 0);
      ncon = sel(skip || cube_hi < 0, 0, cube_hi - cube_lo + 1);
      clean_rows(skip);
      solve_alone_impl<true>();
    } else {
      ncon = sel(skip, 0, ncon_all);
      clean_rows(skip);
      solve_alone_impl<false>();
    }
    ncon = ncon_all;
  }
  // D of list entry c as the cube-alone solve sees it: zero for the entries that are not the cube's (a pad or an arm mesh on the table /
  // the ground: the robot's own, cooperative solve carries those).  filt: wave-uniform, some lane of the wave holds such an entry.
  // list position of the stretch's c-th entry.  Past the lane's own stretch (a wave-mate's is longer) the lane re-reads its last entry --
  // finite numbers, weighted by D = 0 -- rather than whatever lies behind the list (prepare() pads with zeros only up to the wave's
  // longest LIST, and 0 x garbage is not 0)
  template <bool OFF> MCG_DEV int li(int c) const {
    if constexpr (OFF) { const int k = c0 + sel(c < ncon, c, ncon - 1); return sel(k > 0, k, 0); } else return c;
  }
  template <bool OFF> MCG_DEV real alone_D(int c) const {
    const int b = LDS_CON + li<OFF>(c) * CON_STRIDE;
    bool keep = c < ncon;
    if constexpr (OFF) keep = keep && pair_has_cube((int)S.ld(b + CON_TYPE));
    return sel(keep, S.ld(b + CON_D), 0.0);
  }
  template <bool OFF> MCG_DEV void solve_alone_impl() {
    derive(model());
    const real Bc = B_tc; const real mu[3] = {mu_tc[0], mu_tc[1], mu_tc[2]};
    real a[6];
    _Pragma("unroll") for (int k = 0; k < 6; k++) a[k] = a_c[k];
    // The six pyramid rows of a contact are Jn +- mu_k J_k over the basis B = [Jn J1 J2 Jt]: every J.v is a combination
    // of four dot products, and sum_r w_r j_r j_r^T = B W B^T with a 4x4 arrow matrix W (7 numbers), sum_r w_r a_r j_r = B t.
    auto dots = [](const CubeRows& R, const real* v, real* o) {
      o[0] = o[1] = o[2] = o[3] = 0;
      _Pragma("unroll") for (int d = 0; d < 6; d++) { o[0] = fma(R.Jn[d], v[d], o[0]); o[1] = fma(R.J1[d], v[d], o[1]); o[2] = fma(R.J2[d], v[d], o[2]); o[3] = fma(R.Jt[d], v[d], o[3]); }
    };
    bool conv = false;
    for (int it = 0; it < 50; it++) {
      MCG_COUNTW(CN_CUBE_IT, 1);
#ifdef MCG_STAGE_CLOCKS
      if ((threadIdx.x & 63) == 0) atomicAdd(&g_wg_stat[(blockIdx.x & 4095) * 4 + 2], 1ull);
#endif
      real H[21], g[6];
      _Pragma("unroll") for (int k = 0; k < 21; k++) H[k] = 0;
      _Pragma("unroll") for (int k = 0; k < 6; k++) { H[tri(k, k)] = Md[k]; g[k] = fs[k]; }
      for (int c = 0; __any(c < ncon); c++) {                    // pass A: mask (first iteration) + assembly
        CubeRows R; rows_cube(li<OFF>(c), R);
        const int b = LDS_CON + li<OFF>(c) * CON_STRIDE;
        const real D = alone_D<OFF>(c), kterm = S.ld(b + CON_KTERM);
        const int mask = (int)S.ld(LDS_ACT + c);
        real dv[4], da[4]; dots(R, Cb.vel, dv); dots(R, a, da);
        real W00 = 0, t0 = 0, W0[3], Wd[3], t[3];
        int m0 = 0;
        static_for<3>([&](auto Kk) {
          constexpr int k = Kk;
          const real m = mu[k];
          const real arp = -Bc * fma(m, dv[1 + k], dv[0]) - kterm, arm = -Bc * fma(-m, dv[1 + k], dv[0]) - kterm;
          const bool bp0 = fma(m, da[1 + k], da[0]) - arp < 0, bm0 = fma(-m, da[1 + k], da[0]) - arm < 0;
          m0 |= (bp0 ? (1 << (2 * k)) : 0) | (bm0 ? (1 << (2 * k + 1)) : 0);
          const bool bp = sel(it == 0, bp0, ((mask >> (2 * k)) & 1) != 0), bm = sel(it == 0, bm0, ((mask >> (2 * k + 1)) & 1) != 0);
          const real wp = sel(bp, D, 0.0), wm = sel(bm, D, 0.0);
          W00 += wp + wm; t0 = fma(wp, arp, fma(wm, arm, t0));
          W0[k] = m * (wp - wm); Wd[k] = m * m * (wp + wm); t[k] = m * (wp * arp - wm * arm);
        });
        real U0[6], U1[6], U2[6], U3[6];
        _Pragma("unroll") for (int d = 0; d < 6; d++) {
          g[d] += R.Jn[d] * t0 + R.J1[d] * t[0] + R.J2[d] * t[1] + R.Jt[d] * t[2];
          U0[d] = W00 * R.Jn[d] + W0[0] * R.J1[d] + W0[1] * R.J2[d] + W0[2] * R.Jt[d];
          U1[d] = W0[0] * R.Jn[d] + Wd[0] * R.J1[d];
          U2[d] = W0[1] * R.Jn[d] + Wd[1] * R.J2[d];
          U3[d] = W0[2] * R.Jn[d] + Wd[2] * R.Jt[d];
        }
        static_for<6>([&](auto Dd) { constexpr int d = Dd;
          static_for<d + 1>([&](auto Ee) { constexpr int e = Ee;
            H[tri(d, e)] += R.Jn[d] * U0[e] + R.J1[d] * U1[e] + R.J2[d] * U2[e] + R.Jt[d] * U3[e]; }); });
        if (it == 0 && c < ncon) S.st(LDS_ACT + c, (real)m0);
      }
      real x[6], dinv[6];
      _Pragma("unroll") for (int k = 0; k < 6; k++) x[k] = g[k];
      spd_factor<6>(H, dinv); spd_solve<6>(H, dinv, x);
      real p[6]; _Pragma("unroll") for (int k = 0; k < 6; k++) p[k] = x[k] - a[k];
      bool same = true;
      for (int c = 0; __any(c < ncon); c++) {                    // pass B: consistency at x, line-search data
        CubeRows R; rows_cube(li<OFF>(c), R);
        const int b = LDS_CON + li<OFF>(c) * CON_STRIDE;
        const real D = alone_D<OFF>(c), kterm = S.ld(b + CON_KTERM);
        const bool mine = D != 0.0;                               // (an entry of this lane's list that is the cube's)
        const int mask = (int)S.ld(LDS_ACT + c);
        real dv[4], da[4], dp[4]; dots(R, Cb.vel, dv); dots(R, a, da); dots(R, p, dp);
        static_for<3>([&](auto Kk) {
          constexpr int k = Kk;
          static_for<2>([&](auto Od) {
            constexpr int odd = Od; constexpr int r = 2 * k + odd;
            const real m = odd ? -mu[k] : mu[k];
            const real r0 = fma(m, da[1 + k], da[0]) - (-Bc * fma(m, dv[1 + k], dv[0]) - kterm), jp = fma(m, dp[1 + k], dp[0]);
            if (mine) { S.st(LDS_ROW + (c * 6 + r) * 2, r0); S.st(LDS_ROW + (c * 6 + r) * 2 + 1, jp); }
            const real rx = r0 + jp;
            same = same && (!mine || (rx < 0) == (((mask >> r) & 1) != 0));
          });
        });
      }
      const bool finish = !conv && same;
      _Pragma("unroll") for (int k = 0; k < 6; k++) a[k] = sel(finish, x[k], a[k]);
      conv = conv || finish;
      if (!__any(!conv)) break;
      // Line search along p.  Any descent step that ends in a consistent active set yields the exact minimiser at the
      // final full step, so bisection of phi' on [0, 2] is enough (the oracle walks the breakpoints exactly).
      real lin0 = 0, quad = 0;
      _Pragma("unroll") for (int k = 0; k < 6; k++) { const real as = fs[k] / Md[k]; lin0 += Md[k] * (a[k] - as) * p[k]; quad += Md[k] * p[k] * p[k]; }
      MCG_COUNTW(CN_CUBE_LS, 1);
#ifdef MCG_STAGE_CLOCKS
      if ((threadIdx.x & 63) == 0) atomicAdd(&g_wg_stat[(blockIdx.x & 4095) * 4 + 3], 1ull);
#endif
      const real alpha = bisect<OFF>(lin0, quad, conv);
      _Pragma("unroll") for (int k = 0; k < 6; k++) a[k] = sel(conv, a[k], a[k] + alpha * p[k]);
      remask(alpha, conv);
    }
    _Pragma("unroll") for (int k = 0; k < 6; k++) a_c[k] = a[k];
    solved = true;
  }

  // phi'(alpha) = lin0 + alpha quad + sum_rows D min(0, r0 + alpha dr) dr over the contact rows stored at LDS_ROW
  template <bool OFF> MCG_DEV void dphi_rows_alone(real al, real& s, real& slope) const {     // the rows' part of phi' and phi''
    for (int c = 0; __any(c < ncon); c++) {
      const real D = alone_D<OFF>(c);
      _Pragma("unroll") for (int r = 0; r < 6; r++) {
        const real r0 = S.ld(LDS_ROW + (c * 6 + r) * 2), dr_ = S.ld(LDS_ROW + (c * 6 + r) * 2 + 1);
        const real rr = r0 + al * dr_;
        s += sel((rr < 0), D * rr * dr_, 0.0); slope += sel((rr < 0), D * dr_ * dr_, 0.0);
      }
    }
  }
  // Exact line search of the cube-alone solve: phi' is piecewise linear and increasing, so Newton on it (bracketed, bisection as the
  // fallback) lands on the root once it is on the root's own piece -- 3 to 6 evaluations where the 24-step bisection of round 2 made
  // 25 (a tumbling cube then held up its whole wave for ~25 us per sub-step).  `done`: lanes that do not search (wave-uniform exit).
  template <bool OFF> MCG_DEV real bisect(real lin0, real quad, bool done) const {
    auto dphi = [&](real al, real& slope) { real sacc = lin0 + al * quad; slope = quad; dphi_rows_alone<OFF>(al, sacc, slope); return sacc; };
    real lo = 0, hi = 2, sl;
    const bool beyond = dphi(hi, sl) < 0;
    real al = 1.0;
    for (int b = 0; b < 12; b++) {
      const real f = dphi(al, sl);
      const bool neg = f < 0;
      lo = sel(neg, al, lo); hi = sel(neg, hi, al);
      const real nw = al - f / sl;
      const real nx = sel(nw >= lo && nw <= hi, nw, 0.5 * (lo + hi));      // closed: AT the root the step is zero and nw == al == lo or hi
      const bool moved = fabs(nx - al) > 1e-12 * fmax(1.0, fabs(al));
      al = nx;
      if (!__any(moved && !done && !beyond)) break;
    }
    return beyond ? 2.0 : al;
  }
  MCG_DEV void remask(real alpha, bool conv) const {
    for (int c = 0; __any(c < ncon); c++) {
      int mask = 0;
      _Pragma("unroll") for (int r = 0; r < 6; r++) {
        const real r0 = S.ld(LDS_ROW + (c * 6 + r) * 2), dr_ = S.ld(LDS_ROW + (c * 6 + r) * 2 + 1);
        mask |= sel((r0 + alpha * dr_ < 0), (1 << r), 0);
      }
      if (c < ncon && !conv) S.st(LDS_ACT + c, (real)mask);
    }
  }

  // ------------------------------------------------------------------------------------------------- finish
  // implicit-damping Euler on the diagonal cube inertia, free-joint integration (mj_Euler / mj_advance)
  MCG_DEV void finish(real* qlag7) {
    derive(model());
    _Pragma("unroll") for (int k = 0; k < 3; k++) qlag7[k] = Cb.pos[k];
    _Pragma("unroll") for (int k = 0; k < 4; k++) qlag7[3 + k] = Cb.quat[k];
    _Pragma("unroll") for (int k = 0; k < 6; k++) {
      const real acc = (Md[k] * a_c[k]) / (Md[k] + h * damp[k]);      // fs + J^T f = M a at the minimiser: no force needed
      Cb.vel[k] += h * acc;
      Cb.warm[k] = a_c[k];
    }
    _Pragma("unroll") for (int k = 0; k < 3; k++) Cb.pos[k] += h * Cb.vel[k];
    real ax[3] = {Cb.vel[3], Cb.vel[4], Cb.vel[5]};
    const real nw = sqrt(dot3(ax, ax));
    const bool tiny = nw < MINVAL;
    _Pragma("unroll") for (int k = 0; k < 3; k++) ax[k] = sel(tiny, k == 0 ? 1.0 : 0.0, ax[k] / nw);
    const real ang = h * nw;
    real sh, ch; sincos(0.5 * ang, &sh, &ch);
    const real qr[4] = {ch, ax[0]*sh, ax[1]*sh, ax[2]*sh};
    real qn[4]; mulquat(Cb.quat, qr, qn);
    _Pragma("unroll") for (int k = 0; k < 4; k++) Cb.quat[k] = qn[k];
  }
};

// rotations.mat2euler (gymnasium_robotics) as called at mycobot.py:355-357 [RECALL]
MCG_DEV void mat2euler(const real* m, real* e) {
  const real cy = sqrt(m[8]*m[8] + m[5]*m[5]);
  const bool ok = cy > 4 * 2.220446049250313e-16;
  e[2] = sel(ok, -atan2(m[1], m[0]), -atan2(-m[3], m[4]));
  e[1] = -atan2(-m[2], cy);
  e[0] = sel(ok, -atan2(m[5], m[8]), 0.0);
}

}  // namespace mcg
