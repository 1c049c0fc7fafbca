// mcg_mesh.hpp -- the narrow phase of the mesh geoms' collision polytopes, ONE (environment, mesh, other) PAIR PER WAVE: lane = vertex /
// face / edge of the polytope.
//
// Replaces, for the fourteen mesh geoms of the arm and the gripper against the ground plane, the table and the cube
// (/root/reference/mycobotgym/envs/assets/mycobot280_main.xml:81,87,105-247,260-263; excludes :27-37 concern mesh-mesh pairs, not built),
// what mj_collision (P4) does through libccd on the meshes' convex hulls: one contact per pair.  The rule is the oracle's
// (oracle/mco_collision.c: plane_polytope, box_polytope), EXACT on the polytopes: the shapes overlap iff no facet normal of their
// Minkowski difference separates them -- the box's face axes (B), the polytope's face normals (P), the directions e_k x b_j that lie in
// the normal cone of polytope edge k (E) -- and the contact sits on the axis of least penetration.
//
// Why one pair per wave.  Round 3 ran a 16-axis test lane = env: every lane of a wave walked a mesh's vertices through scalar loads
// whenever ONE lane's link came near the table (27 k clocks of the M / RNE waves per sub-step under a random policy), on axes that
// reported contacts that did not exist (3.8 - 23 % of them).  Now a per-lane broad phase (mcg_cube.hpp: mesh_broad)
// leaves a candidate mask per environment, and between barriers S1b and S1c the cube, M and RNE waves -- all 64 lanes -- take the
// environments that have candidates (dealt out by their share of the candidate pairs); per candidate pair the body's frame -- parked in
// LDS by the broad phase that found the candidate -- carries the box into the mesh's frame and every lane evaluates its vertices /
// faces / edges from the tables
// (struct-of-arrays in global memory, padded to the wave width: coalesced loads), DPP reductions pick the winner, and the contact is
// appended to the environment's list.  A separated pair usually ends after the B and P families (~300 instructions).
#pragma once

#include "mcg_cube.hpp"

namespace mcg {

constexpr real MESH_TIE = 1e-12, MESH_EDGE_MIN_SIN = 1e-6, MESH_FACE_MARGIN = 1e-9;

// ---- cross-lane helpers, 64 lanes
MCG_DEV real mesh_rdlane(real v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
template <int CTRL> MCG_DEV real mesh_dpp(real v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
MCG_DEV real wave_max(real v) {        // the same (uniform) number in every lane
  v = fmax(v, mesh_dpp<0xB1>(v)); v = fmax(v, mesh_dpp<0x4E>(v)); v = fmax(v, mesh_dpp<0x141>(v)); v = fmax(v, mesh_dpp<0x140>(v));
  return fmax(fmax(mesh_rdlane(v, 0), mesh_rdlane(v, 16)), fmax(mesh_rdlane(v, 32), mesh_rdlane(v, 48)));
}
MCG_DEV int wave_min_int(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false)); v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false)); v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// the largest of the lanes' values and, among the lanes that hold it, the smallest id; `lane` = that lane
MCG_DEV void wave_argmax(real v, int id, real& vmax, int& idmin, int& lane) {
  vmax = wave_max(v);
  idmin = wave_min_int(sel(v == vmax, id, 0x7fffffff));
  lane = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(__ballot(v == vmax && id == idmin) | (1ull << 63)));
}

// the tables of one mesh (uniform): struct-of-arrays, see mycobotgym_amd/model/polytope.py: pack
struct MeshTab { const real* v; const real* f; const real* e; int nv, nf, ne, vp, fp, ep; };
MCG_DEV MeshTab mesh_tab(const real* __restrict__ poly, int m) {
  const real* meta = poly + 8 * m;
  MeshTab T;
  T.nv = __builtin_amdgcn_readfirstlane((int)meta[0]); T.nf = __builtin_amdgcn_readfirstlane((int)meta[1]); T.ne = __builtin_amdgcn_readfirstlane((int)meta[2]);
  const int off = __builtin_amdgcn_readfirstlane((int)meta[3]);
  T.vp = __builtin_amdgcn_readfirstlane((int)meta[4]); T.fp = __builtin_amdgcn_readfirstlane((int)meta[5]); T.ep = __builtin_amdgcn_readfirstlane((int)meta[6]);
  T.v = poly + off; T.f = T.v + 3 * T.vp; T.e = T.f + 4 * T.fp;
  return T;
}

// Table sizes the routines below rely on (asserted when a polytope block is accepted: mcg_create; polytope.py builds within them): at most
// MESH_VCH x 64 vertices, MESH_FCH x 64 faces.  Every load of a family is issued before the first use: a pair's cost is a few L2
// round trips, not one per 64 elements (the first version walked the tables chunk by chunk: 25 dependent round trips per true contact,
// 27 k clocks per pair under a random policy, profiles/r04b/clocks--pnp-ik.log).
constexpr int MESH_VCH = 2, MESH_FCH = 4, MESH_ECH = 5;

// Ground plane z = 0 against the polytope in the frame (R, p): its lowest vertex (first occurrence).  dist < 0 = a contact.
MCG_DEV bool mesh_ground(const MeshTab& T, int L, const real* R, const real* p, real* pos, real* nrm, real& dist) {
  real vx[MESH_VCH], vy[MESH_VCH], vz[MESH_VCH]; int kk[MESH_VCH];
  _Pragma("unroll") for (int c = 0; c < MESH_VCH; c++) {
    kk[c] = sel(64 * c + L < T.vp, 64 * c + L, L);                 // (a chunk beyond the table repeats the first: same values, higher index)
    vx[c] = T.v[kk[c]]; vy[c] = T.v[T.vp + kk[c]]; vz[c] = T.v[2 * T.vp + kk[c]];
  }
  real best = INFINITY; int kb = 0x7fffffff;
  _Pragma("unroll") for (int c = 0; c < MESH_VCH; c++) {
    const real hgt = p[2] + R[6]*vx[c] + R[7]*vy[c] + R[8]*vz[c];
    const bool lower = hgt < best;                                  // (the padding repeats vertex 0: never lower than the first of its value)
    best = sel(lower, hgt, best); kb = sel(lower, kk[c], kb);
  }
  real lo; int kmin, lane;
  wave_argmax(-best, kb, lo, kmin, lane);
  lo = -lo;
  if (!(lo < 0)) return false;                                      // uniform
  const real wx = T.v[kmin], wy = T.v[T.vp + kmin], wz = T.v[2 * T.vp + kmin];
  real w[3];
  _Pragma("unroll") for (int r = 0; r < 3; r++) w[r] = p[r] + R[3*r]*wx + R[3*r+1]*wy + R[3*r+2]*wz;
  pos[0] = w[0]; pos[1] = w[1]; pos[2] = w[2] - 0.5 * lo;
  nrm[0] = 0; nrm[1] = 0; nrm[2] = 1;
  dist = lo;
  return true;
}

// Box (centre pb, axes = the columns of Rb, half sizes h) against the polytope in the frame (Rm, pm).  flip: the mesh is geom1 (against
// the cube): the normal points from the mesh to the box; else from the box to the mesh (the static table is geom1).
MCG_DEV bool mesh_box(const MeshTab& T, int L, const real* Rm, const real* pm, const real* Rb, const real* pb, const real* h, bool flip,
                      real* pos, real* nrm, real& dist) {
  // ---- the vertex table (B family): all loads in flight before anything is used
  real vx[MESH_VCH], vy[MESH_VCH], vz[MESH_VCH]; int kv[MESH_VCH];
  _Pragma("unroll") for (int cch = 0; cch < MESH_VCH; cch++) {
    kv[cch] = sel(64 * cch + L < T.vp, 64 * cch + L, L);
    vx[cch] = T.v[kv[cch]]; vy[cch] = T.v[T.vp + kv[cch]]; vz[cch] = T.v[2 * T.vp + kv[cch]];
  }
  MCG_FENCE();
  real c[3], b[3][3];                                               // the box in the mesh frame: centre, axes (rows)
  { const real rel[3] = {pb[0] - pm[0], pb[1] - pm[1], pb[2] - pm[2]};
    _Pragma("unroll") for (int k = 0; k < 3; k++) c[k] = Rm[k]*rel[0] + Rm[3 + k]*rel[1] + Rm[6 + k]*rel[2];
    _Pragma("unroll") for (int j = 0; j < 3; j++) { _Pragma("unroll") for (int k = 0; k < 3; k++) b[j][k] = Rm[k]*Rb[j] + Rm[3 + k]*Rb[3 + j] + Rm[6 + k]*Rb[6 + j]; } }
  // ---- B: the box's face axes; the polytope's extent along b_j from its vertices
  real mx[3] = {-INFINITY, -INFINITY, -INFINITY}, mn[3] = {INFINITY, INFINITY, INFINITY}; int kx[3] = {0, 0, 0}, kn[3] = {0, 0, 0};
  _Pragma("unroll") for (int cch = 0; cch < MESH_VCH; cch++) {
    _Pragma("unroll") for (int j = 0; j < 3; j++) {
      const real t = vx[cch]*b[j][0] + vy[cch]*b[j][1] + vz[cch]*b[j][2];
      const bool up = t > mx[j], dn = t < mn[j];
      mx[j] = sel(up, t, mx[j]); kx[j] = sel(up, kv[cch], kx[j]); mn[j] = sel(dn, t, mn[j]); kn[j] = sel(dn, kv[cch], kn[j]);
    }
  }
  real sB = -INFINITY; int cB = 0;
  real wmx[3], wmn[3];
  _Pragma("unroll") for (int j = 0; j < 3; j++) {
    wmx[j] = wave_max(mx[j]); wmn[j] = -wave_max(-mn[j]);
    const real cb = dot3(c, b[j]);
    const real sp = (cb - h[j]) - wmx[j], sn = wmn[j] - (cb + h[j]);
    const bool t0 = sp > sB; sB = sel(t0, sp, sB); cB = sel(t0, 2 * j, cB);
    const bool t1 = sn > sB; sB = sel(t1, sn, sB); cB = sel(t1, 2 * j + 1, cB);
  }
  if (sB > 0) { MCG_COUNTW(CN_MP_EXIT_B, 1); return false; }        // (uniform: every lane holds the same numbers)
  // the deepest vertex along the winning box axis: the lowest index among the vertices that realise the extreme
  const int jB = cB >> 1; const bool negB = (cB & 1) != 0;
  int vB;
  { const real ext = sel(negB, sel3(jB, wmn[0], wmn[1], wmn[2]), sel3(jB, wmx[0], wmx[1], wmx[2]));
    const real mine = sel(negB, sel3(jB, mn[0], mn[1], mn[2]), sel3(jB, mx[0], mx[1], mx[2]));
    const int kmine = sel(negB, sel3(jB, kn[0], kn[1], kn[2]), sel3(jB, kx[0], kx[1], kx[2]));
    vB = wave_min_int(sel(mine == ext, kmine, 0x7fffffff)); }
  const real pB[3] = {T.v[vB], T.v[T.vp + vB], T.v[2 * T.vp + vB]};
  // ---- the shortcut of a contact in the middle of a box face.  With d = -sB the depth along box axis j and g_i = h_i - |b_i.(p* - c)|
  // the clearances of the deepest vertex p* to the box's faces along the other two axes: every facet of the Minkowski difference with
  // unit normal n lies at h_box(n) + h_poly(-n) >= h_box(n) - n.(p* - c) >= sum_i |n_i| (h_i - |x_i|) >= |n_j| d + (|n_i1| + |n_i2|) min(g)
  // from the origin, which is >= d (|n|_1 >= 1) once both clearances are: no face normal of the polytope and no edge axis can be the
  // axis of least penetration, and none can separate.  The exhaustive rule (the oracle's) then returns the B axis too -- ties go to
  // B -- so the P and E families, two thirds of a touching pair's instructions, are skipped for an arm link lying on the table away from
  // its rim.  (Not taken at d <= 1e-12, where the exhaustive rule's rounding decides whether the pair touches at all.)
  bool faceB;
  { real g = INFINITY;
    _Pragma("unroll") for (int i = 0; i < 3; i++) {
      const real xi = dot3(b[i], pB) - dot3(c, b[i]);
      g = sel(i == jB, g, fmin(g, h[i] - fabs(xi)));
    }
    faceB = -sB > 1e-12 && g >= -sB + MESH_FACE_MARGIN; }
  real s = sB; int kind = 0;
  real sE = -INFINITY; int cE = 0, lnE = 0, cP = 0;
  real nEl[3] = {0, 0, 0};
  if (!faceB) {                                                     // uniform
  real fn[MESH_FCH][4]; int kf[MESH_FCH];
  _Pragma("unroll") for (int cch = 0; cch < MESH_FCH; cch++) {
    kf[cch] = sel(64 * cch + L < T.fp, 64 * cch + L, L);
    _Pragma("unroll") for (int q = 0; q < 4; q++) fn[cch][q] = T.f[q * T.fp + kf[cch]];
  }
  real eA[12];
  _Pragma("unroll") for (int q = 0; q < 12; q++) eA[q] = T.e[q * T.ep + L];
  MCG_FENCE();
  // ---- P: the polytope's face normals
  real sPl = -INFINITY; int cPl = 0x7fffffff;
  _Pragma("unroll") for (int cch = 0; cch < MESH_FCH; cch++) {
    const real* n = fn[cch];
    const real s = (dot3(n, c) - (h[0]*fabs(dot3(n, b[0])) + h[1]*fabs(dot3(n, b[1])) + h[2]*fabs(dot3(n, b[2])))) - n[3];
    const bool up = s > sPl || (s == sPl && kf[cch] < cPl);         // (padding faces: d = 1e30, never the largest; a repeated chunk: same value, higher index)
    sPl = sel(up, s, sPl); cPl = sel(up, kf[cch], cPl);
  }
  real sP; int lnP;
  wave_argmax(sPl, cPl, sP, cP, lnP);
  if (sP > 0) { MCG_COUNTW(CN_MP_EXIT_P, 1); return false; }
  // ---- E: e_k x b_j inside the normal cone of edge k; the next chunk's loads are in flight while one is evaluated (two buffers, the loop
  // unrolled by two: no run-time index into a private array)
  real sEl = -INFINITY; int cEl = 0x7fffffff;
  auto edges = [&](const real* e12, int k) {
    _Pragma("unroll") for (int j = 0; j < 3; j++) {
      constexpr int I1[3] = {1, 2, 0}, I2[3] = {2, 0, 1};
      const int i1 = I1[j], i2 = I2[j];
      real x[3]; cross(e12 + 3, b[j], x);
      const real l2 = dot3(x, x);
      const bool ok = l2 >= MESH_EDGE_MIN_SIN * MESH_EDGE_MIN_SIN;  // (padding edges: e = 0)
      // 1 / |x|: v_rsq_f64 and two Newton steps (the oracle's sqrt and division cost 25 issue slots per candidate, 900 candidates per
      // pair; the results differ from its in the last bit)
      const real l2s = sel(ok, l2, 1.0);
      real il = __builtin_amdgcn_rsq(l2s);
      il = il * fma(-0.5 * l2s * il, il, 1.5); il = il * fma(-0.5 * l2s * il, il, 1.5);
      x[0] *= il; x[1] *= il; x[2] *= il;
      const real t1 = x[0]*e12[6] + x[1]*e12[7] + x[2]*e12[8], t2 = x[0]*e12[9] + x[1]*e12[10] + x[2]*e12[11];
      const real sg = sel(t1 >= 0 && t2 >= 0, 1.0, sel(t1 <= 0 && t2 <= 0, -1.0, 0.0));
      const real n[3] = {sg*x[0], sg*x[1], sg*x[2]};
      const real s = (dot3(n, c) - (h[i1]*fabs(dot3(n, b[i1])) + h[i2]*fabs(dot3(n, b[i2])))) - (n[0]*e12[0] + n[1]*e12[1] + n[2]*e12[2]);
      const int id = j * T.ne + k;
      const bool up = ok && sg != 0.0 && (s > sEl || (s == sEl && id < cEl));
      sEl = sel(up, s, sEl); cEl = sel(up, id, cEl);
      _Pragma("unroll") for (int r = 0; r < 3; r++) nEl[r] = sel(up, n[r], nEl[r]);
    }
  };
  real eB[12];
  for (int k0 = 0; k0 < T.ep; k0 += 128) {                         // (uniform)
    const bool second = k0 + 64 < T.ep;
    if (second) { _Pragma("unroll") for (int q = 0; q < 12; q++) eB[q] = T.e[q * T.ep + k0 + 64 + L]; }
    MCG_FENCE();
    edges(eA, k0 + L);
    if (second) {
      if (k0 + 128 < T.ep) { _Pragma("unroll") for (int q = 0; q < 12; q++) eA[q] = T.e[q * T.ep + k0 + 128 + L]; }
      MCG_FENCE();
      edges(eB, k0 + 64 + L);
    }
  }
  wave_argmax(sEl, cEl, sE, cE, lnE);
  if (sE > 0) { MCG_COUNTW(CN_MP_EXIT_E, 1); return false; }
  // ---- the axis of least penetration (families in the order B, E, P; a later one must be larger by more than MESH_TIE) and its contact
  if (sE > s + MESH_TIE) { s = sE; kind = 1; }
  if (sP > s + MESH_TIE) { s = sP; kind = 2; }
  MCG_COUNTW(CN_MP_KIND_E, kind == 1); MCG_COUNTW(CN_MP_KIND_P, kind == 2);
  } else MCG_COUNTW(CN_MP_FACE, 1);
  real n[3], q[3];
  if (kind == 1) {                                                  // uniform branches
    _Pragma("unroll") for (int r = 0; r < 3; r++) n[r] = mesh_rdlane(nEl[r], lnE);
    const int jE = cE / T.ne, kE = cE - jE * T.ne;
    const int i1 = (jE + 1) % 3, i2 = (jE + 2) % 3;
    real pe[3], e[3];
    _Pragma("unroll") for (int r = 0; r < 3; r++) { pe[r] = T.e[r * T.ep + kE]; e[r] = T.e[(3 + r) * T.ep + kE]; }
    const real elen = T.e[12 * T.ep + kE];
    real bj[3], b1[3], b2[3];
    _Pragma("unroll") for (int r = 0; r < 3; r++) { bj[r] = sel3(jE, b[0][r], b[1][r], b[2][r]); b1[r] = sel3(i1, b[0][r], b[1][r], b[2][r]); b2[r] = sel3(i2, b[0][r], b[1][r], b[2][r]); }
    const real h1 = sel3(i1, h[0], h[1], h[2]), h2 = sel3(i2, h[0], h[1], h[2]), hj = sel3(jE, h[0], h[1], h[2]);
    const real g1 = sel(dot3(b1, n) > 0, 1.0, -1.0) * h1, g2 = sel(dot3(b2, n) > 0, 1.0, -1.0) * h2;
    real qb[3];
    _Pragma("unroll") for (int r = 0; r < 3; r++) qb[r] = c[r] - g1*b1[r] - g2*b2[r];                   // a point of the box's supporting edge along -n
    const real w[3] = {qb[0] - pe[0], qb[1] - pe[1], qb[2] - pe[2]};
    const real uaub = dot3(e, bj), q1 = dot3(e, w), q2 = -dot3(bj, w), dd = 1 - uaub*uaub;
    real ts = dd <= 1e-12 ? 0.0 : (q1 + uaub*q2) / dd, tt = dd <= 1e-12 ? 0.0 : (uaub*q1 + q2) / dd;
    ts = fmin(fmax(ts, 0.0), elen); tt = fmin(fmax(tt, -hj), hj);
    _Pragma("unroll") for (int r = 0; r < 3; r++) q[r] = 0.5 * ((pe[r] + ts*e[r]) + (qb[r] + tt*bj[r]));
  } else if (kind == 2) {
    n[0] = T.f[cP]; n[1] = T.f[T.fp + cP]; n[2] = T.f[2 * T.fp + cP];
    const real g0 = sel(dot3(b[0], n) > 0, 1.0, -1.0) * h[0], g1 = sel(dot3(b[1], n) > 0, 1.0, -1.0) * h[1], g2 = sel(dot3(b[2], n) > 0, 1.0, -1.0) * h[2];
    _Pragma("unroll") for (int r = 0; r < 3; r++) q[r] = (c[r] - g0*b[0][r] - g1*b[1][r] - g2*b[2][r]) - 0.5*s*n[r];      // the box's deepest corner, half a depth forward
  } else {
    const real sg = negB ? -1.0 : 1.0;
    _Pragma("unroll") for (int r = 0; r < 3; r++) n[r] = sg * sel3(jB, b[0][r], b[1][r], b[2][r]);
    q[0] = pB[0] + 0.5*s*n[0]; q[1] = pB[1] + 0.5*s*n[1]; q[2] = pB[2] + 0.5*s*n[2];                                    // the deepest vertex, half a depth back
  }
  _Pragma("unroll") for (int r = 0; r < 3; r++) {
    pos[r] = pm[r] + Rm[3*r]*q[0] + Rm[3*r+1]*q[1] + Rm[3*r+2]*q[2];
    const real nw = Rm[3*r]*n[0] + Rm[3*r+1]*n[1] + Rm[3*r+2]*n[2];
    nrm[r] = flip ? nw : -nw;
  }
  dist = s;
  return true;
}

// One environment's candidate pairs, by the 64 lanes of the calling wave.  lds0 = slot 0 of lane 0 of the workgroup's array, e = the
// environment's lane (its LDS column).  The frames of the bodies that carry a candidate were parked in the column by the lane-parallel
// broad phases (MP_FRAME: one producer per body), the cube's pose by the cube wave.
MCG_DEV void mesh_env(ModelPtr Pm, const real* __restrict__ poly, LdsPtr lds0, int e) {
  const int L = threadIdx.x & 63;
  const PnpScratch ME(lds0 + e);                   // env e's column: a uniform slot index is a broadcast read
  unsigned long long bits = (unsigned long long)(long long)ME.ld(MP_MASK) | (unsigned long long)(long long)ME.ld(MP_MASK + 1) | (unsigned long long)(long long)ME.ld(MP_MASK + 2);
  bits = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(bits >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)bits);
  int ncon = __builtin_amdgcn_readfirstlane((int)ME.ld(MP_NCON)), ndrop = 0;
  // ---- the static box and the cube
  real tp[3], th[3], hc[3], Rc[9], cp[3];
  { ModelPtr Qb = launder(Pm); ldc<3>(Qb->table_pos, tp); ldc<3>(Qb->table_half, th); ldc<3>(Qb->cube_half, hc); }
  { real cq[4]; _Pragma("unroll") for (int k = 0; k < 3; k++) cp[k] = ME.ld(MP_CUBE + k);
    _Pragma("unroll") for (int k = 0; k < 4; k++) cq[k] = ME.ld(MP_CUBE + 3 + k);
    quat_to_mat(cq, Rc); }
  const real Rt[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const real mult = launder(Pm)->mesh_mult;
  // ---- the candidate pairs in the oracle's order: mesh by mesh; ground, table, cube
  while (bits != 0ull) {                                            // uniform
    const int bit = (int)__builtin_ctzll(bits); bits &= bits - 1ull;
    const int m = bit / 3, o = bit - 3 * m, body = mesh_body(m);
    real Rm[9], pm[3];
    _Pragma("unroll") for (int k = 0; k < 9; k++) Rm[k] = ME.ld(MP_FRAME + body * 12 + k);
    _Pragma("unroll") for (int k = 0; k < 3; k++) pm[k] = ME.ld(MP_FRAME + body * 12 + 9 + k);
    const MeshTab T = mesh_tab(poly, m);
    real pos[3], nrm[3], dist = 1.0; bool hit;
    MCG_COUNTW(CN_MP_PAIRS, 1); MCG_COUNTW(CN_MP_CUBE, o == 2);
    if (o == 0) hit = mesh_ground(T, L, Rm, pm, pos, nrm, dist);
    else {
      const bool cube = o == 2;
      real Rb[9], pb[3], hb[3];
      _Pragma("unroll") for (int k = 0; k < 9; k++) Rb[k] = sel(cube, Rc[k], Rt[k]);
      _Pragma("unroll") for (int k = 0; k < 3; k++) { pb[k] = sel(cube, cp[k], tp[k]); hb[k] = sel(cube, hc[k], th[k]); }
      hit = mesh_box(T, L, Rm, pm, Rb, pb, hb, cube, pos, nrm, dist);
    }
    if (hit) {                                                      // uniform
      MCG_COUNTW(CN_MP_HITS, 1);
      if (ncon < MAXCON) {
        const int bse = LDS_CON + ncon * CON_STRIDE;
        const int type = (o == 2 ? PAIR_MESH0_CUBE : PAIR_STATIC_MESH0) + m;
        if (L < CON_STRIDE) {
          real v = pos[0];
          v = sel(L == 1, pos[1], v); v = sel(L == 2, pos[2], v); v = sel(L == 3, nrm[0], v); v = sel(L == 4, nrm[1], v); v = sel(L == 5, nrm[2], v);
          v = sel(L == CON_DIST, dist, v); v = sel(L == CON_D, mult, v); v = sel(L == CON_KTERM, 0.0, v); v = sel(L == CON_TYPE, (real)type, v);
          ME.st(bse + L, v);
        }
        ncon++;
      } else ndrop += (int)mult;
    }
  }
  if (L == 0) { ME.st(MP_NCON, (real)ncon); ME.st(MP_DROP, (real)ndrop); }
}

// ---- the mesh phase of a sub-step: every wave that takes part calls it with all 64 lanes between barriers S1b and S1c.  The environments
// that have candidates are dealt out by their share of the candidate PAIRS: environment e (in lane order) goes to wave
// floor(nwaves x (pairs before e + half of e's) / all pairs) -- one wave keeps an environment's pairs, so that its contacts are appended in pair order.
// (The first version grabbed environments from a counter in LDS as the cooperative solves do; that loop hung on the GPU in this kernel
// and in no reduced copy of it, and was not understood: gpurun_out logs of round 4.  Which wave takes an environment does not change its
// result either way.)
MCG_DEV void mesh_phase(ModelPtr Pm, const real* __restrict__ poly, LdsPtr lds0, int wave, int nwaves) {
  const int T = threadIdx.x & 63;
  int cnt = 0;
  if (T < PNP_LANES) {
    const PnpScratch MS(lds0 + T);
    const unsigned long long b = (unsigned long long)(long long)MS.ld(MP_MASK) | (unsigned long long)(long long)MS.ld(MP_MASK + 1) | (unsigned long long)(long long)MS.ld(MP_MASK + 2);
    cnt = __popcll(b);
  }
  unsigned em = __builtin_amdgcn_readfirstlane((unsigned)__ballot(cnt != 0));
  if (em == 0u) return;                                             // uniform: the usual case away from the table
  int total = 0;
  for (unsigned mm = em; mm != 0u; mm &= mm - 1u) total += __builtin_amdgcn_readlane(cnt, __builtin_ctz(mm));
  int before = 0;
  while (em != 0u) {                                                // uniform
    const int e = __builtin_ctz(em); em &= em - 1u;
    const int c = __builtin_amdgcn_readlane(cnt, e);
    const int owner = ((2 * before + c) * nwaves) / (2 * total);   // by the share's midpoint: no wave is systematically the heaviest
    before += c;
    if (owner == wave) mesh_env(Pm, poly, lds0, e);
  }
}

}  // namespace mcg
