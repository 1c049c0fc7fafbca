/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see mco_physics.h).
 *
 * P4 `mj_collision` restated for the PickAndPlace scene: the primitive geoms -- the cube, the table box, the two finger-pad boxes and the
 * ground plane (/root/reference/mycobotgym/envs/assets/mycobot280_main.xml:81,87,195-199,222-225,262) -- among each other, and the mesh
 * geoms of the arm and the gripper (:105-247; convex hulls in MuJoCo, collision polytopes here: mycobotgym_amd/model/polytope.py)
 * against the ground, the table and the cube.  Not built: mesh <-> mesh pairs (self-collision) and the arm base's mesh (its STL is
 * absent from the reference checkout).  With `collide_scope_geom` set (the build's configuration) only primitive pairs that involve
 * the cube, or a static geom and a pad, are tested.
 *
 * Pair filter [RECALL mj_collision]: both geoms' bodies welded to the world -> skip; same weld body -> skip;
 * parent-child weld bodies -> skip unless the parent is the world; `<exclude>` pairs -> skip;
 * (contype1 & conaffinity2) | (contype2 & conaffinity1) must be non-zero; bounding-sphere rejection.
 * Pair parameters [RECALL mj_contactParam]: condim = max, friction = element-wise max, solref = mean if both
 * time constants are positive else element-wise min, solimp = mean, margin = gap = 0.
 *
 * Narrow phase.  MuJoCo's own mjc_BoxBox / mjc_PlaneBox are restated by their published behaviour, not line by
 * line: plane-box reports every box vertex below the plane; box-box is the classic separating-axis test over the
 * 15 axes followed by clipping of the incident face against the reference face (face contact, up to 8 points) or
 * the closest points of the two edges (edge contact, 1 point).  Contact position = midpoint between the two
 * surfaces, frame x axis = normal from geom1 to geom2, dist < 0 = penetration; only dist < margin (= 0) is kept.
 * The HIP kernel implements exactly this procedure and is checked against it.
 */
#include "mco_physics.h"

#include <math.h>
#include <string.h>

static double dot3(const double* a, const double* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static void col(const double* M, int k, double* v) { v[0] = M[k]; v[1] = M[3 + k]; v[2] = M[6 + k]; }

/* mju_makeFrame [RECALL]: complete a unit normal to a right-handed orthonormal frame (rows of `frame`) */
static void make_frame(double* frame) {
  double* n = frame; double* t1 = frame + 3; double* t2 = frame + 6;
  double tmp[3] = { 0, 0, 0 };
  if (n[1] < 0.5 && n[1] > -0.5) tmp[1] = 1; else tmp[2] = 1;
  double d = dot3(n, tmp);
  for (int k = 0; k < 3; k++) t1[k] = tmp[k] - d * n[k];
  double l = sqrt(dot3(t1, t1));
  for (int k = 0; k < 3; k++) t1[k] /= l;
  cross3(t2, n, t1);
}

static void mix_params(const mco_model* m, int g1, int g2, mco_contact* c) {
  c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
  double f[3];
  for (int k = 0; k < 3; k++) f[k] = fmax(m->geom_friction[g1][k], m->geom_friction[g2][k]);
  c->friction[0] = c->friction[1] = f[0]; c->friction[2] = f[1]; c->friction[3] = c->friction[4] = f[2];
  if (m->geom_solref[g1][0] > 0 && m->geom_solref[g2][0] > 0)
    for (int k = 0; k < 2; k++) c->solref[k] = 0.5 * (m->geom_solref[g1][k] + m->geom_solref[g2][k]);
  else
    for (int k = 0; k < 2; k++) c->solref[k] = fmin(m->geom_solref[g1][k], m->geom_solref[g2][k]);
  for (int k = 0; k < 5; k++) c->solimp[k] = 0.5 * (m->geom_solimp[g1][k] + m->geom_solimp[g2][k]);
  c->includemargin = 0;
}

static void add_contact(const mco_model* m, mco_data* d, int g1, int g2, const double* pos, const double* normal, double dist) {
  if (!(dist < 0)) return;
  if (d->nentry >= m->maxentry || d->ncon >= MCO_MAXCON) { d->ndrop++; return; }      /* the build's cap (MuJoCo has none) */
  d->nentry++;
  mco_contact* c = &d->contact[d->ncon++];
  memset(c, 0, sizeof(*c));
  c->geom1 = g1; c->geom2 = g2; c->dist = dist;
  memcpy(c->pos, pos, 3 * sizeof(double));
  memcpy(c->frame, normal, 3 * sizeof(double));
  make_frame(c->frame);
  mix_params(m, g1, g2, c);
}

/* ------------------------------------------------------------------------------------- plane - box */
static void plane_box(const mco_model* m, mco_data* d, int gp, int gb) {
  double n[3]; col(d->geom_xmat[gp], 2, n);
  const double* pp = d->geom_xpos[gp]; const double* pb = d->geom_xpos[gb]; const double* R = d->geom_xmat[gb];
  const double* h = m->geom_size[gb];
  for (int v = 0; v < 8; v++) {
    double loc[3] = { (v & 1 ? h[0] : -h[0]), (v & 2 ? h[1] : -h[1]), (v & 4 ? h[2] : -h[2]) }, w[3];
    for (int k = 0; k < 3; k++) w[k] = pb[k] + R[3*k]*loc[0] + R[3*k+1]*loc[1] + R[3*k+2]*loc[2];
    double rel[3] = { w[0] - pp[0], w[1] - pp[1], w[2] - pp[2] };
    double dist = dot3(rel, n);
    double pos[3] = { w[0] - 0.5*dist*n[0], w[1] - 0.5*dist*n[1], w[2] - 0.5*dist*n[2] };
    add_contact(m, d, gp, gb, pos, n, dist);
  }
}

/* --------------------------------------------------------------------------------------- box - box */
#define EDGE_MIN_SIN 1e-6    /* an edge axis needs edges at least this far from parallel (sine of their angle) */
#define EDGE_FUDGE 1.05      /* an edge axis must beat the best face axis by 5 % (avoids flicker on parallel faces) */

static void box_box(const mco_model* m, mco_data* d, int ga, int gb) {
  const double* Ra = d->geom_xmat[ga]; const double* Rb = d->geom_xmat[gb];
  const double* ha = m->geom_size[ga]; const double* hb = m->geom_size[gb];
  const double* pa = d->geom_xpos[ga]; const double* pb = d->geom_xpos[gb];
  double A[3][3], B[3][3], p[3] = { pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2] };
  for (int k = 0; k < 3; k++) { col(Ra, k, A[k]); col(Rb, k, B[k]); }
  double C[3][3], Q[3][3];              /* C = A^T B, Q = |C| */
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { C[i][j] = dot3(A[i], B[j]); Q[i][j] = fabs(C[i][j]); }
  double pA[3] = { dot3(A[0], p), dot3(A[1], p), dot3(A[2], p) };      /* p in A's frame */
  double pB[3] = { dot3(B[0], p), dot3(B[1], p), dot3(B[2], p) };
  double best = -INFINITY; int code = -1; double nrm[3] = { 0, 0, 0 }; int invert = 0;
  /* face axes of A, then of B */
  for (int i = 0; i < 3; i++) {
    double s = fabs(pA[i]) - (ha[i] + hb[0]*Q[i][0] + hb[1]*Q[i][1] + hb[2]*Q[i][2]);
    if (s > 0) return;
    if (s > best) { best = s; code = i; memcpy(nrm, A[i], sizeof(nrm)); invert = pA[i] < 0; }
  }
  for (int j = 0; j < 3; j++) {
    double s = fabs(pB[j]) - (hb[j] + ha[0]*Q[0][j] + ha[1]*Q[1][j] + ha[2]*Q[2][j]);
    if (s > 0) return;
    if (s > best) { best = s; code = 3 + j; memcpy(nrm, B[j], sizeof(nrm)); invert = pB[j] < 0; }
  }
  /* edge axes A_i x B_j */
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    double expr = pA[i2]*C[i1][j] - pA[i1]*C[i2][j];                  /* p . (A_i x B_j) */
    double L[3]; cross3(L, A[i], B[j]);
    double len = sqrt(dot3(L, L));       /* from the cross product itself: 1 - C^2 is rounding noise of 1e-8 for parallel edges */
    if (len < EDGE_MIN_SIN) continue;                                  /* (nearly) parallel edges: covered by the face axes */
    double s = (fabs(expr) - (ha[i1]*Q[i2][j] + ha[i2]*Q[i1][j] + hb[j1]*Q[i][j2] + hb[j2]*Q[i][j1])) / len;
    if (s > 0) return;
    if (s * EDGE_FUDGE > best) {
      best = s; code = 6 + 3*i + j;
      for (int k = 0; k < 3; k++) nrm[k] = L[k] / len;
      invert = expr < 0;
    }
  }
  if (code < 0) return;
  double normal[3];                                                    /* from box A (geom1) to box B (geom2) */
  for (int k = 0; k < 3; k++) normal[k] = invert ? -nrm[k] : nrm[k];

  if (code >= 6) {
    /* edge-edge: one contact at the midpoint of the closest points of the two supporting edges */
    int i = (code - 6) / 3, j = (code - 6) % 3;
    double ea[3], eb[3];
    for (int k = 0; k < 3; k++) { ea[k] = pa[k]; eb[k] = pb[k]; }
    for (int a = 0; a < 3; a++) if (a != i) { double sg = dot3(normal, A[a]) > 0 ? 1.0 : -1.0; for (int k = 0; k < 3; k++) ea[k] += sg * ha[a] * A[a][k]; }
    for (int b = 0; b < 3; b++) if (b != j) { double sg = dot3(normal, B[b]) > 0 ? -1.0 : 1.0; for (int k = 0; k < 3; k++) eb[k] += sg * hb[b] * B[b][k]; }
    /* closest points of lines ea + s A_i and eb + t B_j */
    double w[3] = { eb[0] - ea[0], eb[1] - ea[1], eb[2] - ea[2] };
    double uaub = C[i][j], q1 = dot3(A[i], w), q2 = -dot3(B[j], w), dd = 1 - uaub*uaub;
    double s = dd <= 1e-12 ? 0 : (q1 + uaub*q2) / dd, t = dd <= 1e-12 ? 0 : (uaub*q1 + q2) / dd;
    double pos[3];
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * ((ea[k] + s*A[i][k]) + (eb[k] + t*B[j][k]));
    add_contact(m, d, ga, gb, pos, normal, best);
    return;
  }

  /* face contact: reference box owns the axis, the incident box is clipped against its face */
  const double (*Rr)[3] = code < 3 ? A : B; const double (*Ri)[3] = code < 3 ? B : A;
  const double* pr = code < 3 ? pa : pb; const double* pi = code < 3 ? pb : pa;
  const double* hr = code < 3 ? ha : hb; const double* hi = code < 3 ? hb : ha;
  int ax = code % 3;
  double n2[3];                                                        /* reference-face normal, pointing at the incident box */
  for (int k = 0; k < 3; k++) n2[k] = code < 3 ? normal[k] : -normal[k];
  /* incident face: the face of the incident box most anti-parallel to n2 */
  double nr[3] = { dot3(n2, Ri[0]), dot3(n2, Ri[1]), dot3(n2, Ri[2]) };
  int lan = fabs(nr[0]) > fabs(nr[1]) ? (fabs(nr[0]) > fabs(nr[2]) ? 0 : 2) : (fabs(nr[1]) > fabs(nr[2]) ? 1 : 2);
  int a1 = (lan + 1) % 3, a2 = (lan + 2) % 3;
  double center[3];
  for (int k = 0; k < 3; k++) center[k] = pi[k] - pr[k] + (nr[lan] < 0 ? hi[lan] : -hi[lan]) * Ri[lan][k];
  int c1 = (ax + 1) % 3, c2 = (ax + 2) % 3;
  /* incident quad in the 2-D coordinates (c1, c2) of the reference face */
  double cx = dot3(center, Rr[c1]), cy = dot3(center, Rr[c2]);
  double m11 = dot3(Rr[c1], Ri[a1]), m12 = dot3(Rr[c1], Ri[a2]), m21 = dot3(Rr[c2], Ri[a1]), m22 = dot3(Rr[c2], Ri[a2]);
  double k1 = m11*hi[a1], k2 = m21*hi[a1], k3 = m12*hi[a2], k4 = m22*hi[a2];
  double poly[16][2] = { { cx - k1 - k3, cy - k2 - k4 }, { cx - k1 + k3, cy - k2 + k4 },
                         { cx + k1 + k3, cy + k2 + k4 }, { cx + k1 - k3, cy + k2 - k4 } }, tmp[16][2];
  int np = 4;
  double rect[2] = { hr[c1], hr[c2] };
  /* Sutherland-Hodgman against x <= r, x >= -r, y <= r, y >= -r */
  for (int dir = 0; dir < 2; dir++) for (int sgn = -1; sgn <= 1; sgn += 2) {
    int nq = 0;
    for (int v = 0; v < np; v++) {
      const double* P = poly[v]; const double* Nx = poly[(v + 1) % np];
      int inP = sgn * P[dir] < rect[dir], inN = sgn * Nx[dir] < rect[dir];
      if (inP) { tmp[nq][0] = P[0]; tmp[nq][1] = P[1]; nq++; }
      if (inP != inN) {
        double tt = (sgn * rect[dir] - P[dir]) / (Nx[dir] - P[dir]);
        tmp[nq][1 - dir] = P[1 - dir] + tt * (Nx[1 - dir] - P[1 - dir]); tmp[nq][dir] = sgn * rect[dir]; nq++;
      }
      if (nq >= 15) break;
    }
    np = nq; memcpy(poly, tmp, sizeof(poly));
    if (np == 0) return;
  }
  /* back to 3-D on the incident face; keep the points that lie below the reference face */
  double det1 = 1.0 / (m11*m22 - m12*m21);
  double im11 = m22*det1, im12 = -m12*det1, im21 = -m21*det1, im22 = m11*det1;
  int kept = 0;
  for (int v = 0; v < np && kept < 8; v++) {
    double qx = poly[v][0] - cx, qy = poly[v][1] - cy;
    double u1 = im11*qx + im12*qy, u2 = im21*qx + im22*qy, pt[3];
    for (int k = 0; k < 3; k++) pt[k] = center[k] + u1*Ri[a1][k] + u2*Ri[a2][k];      /* relative to pr */
    double depth = hr[ax] - dot3(n2, pt);
    if (depth > 0) {
      double pos[3];
      for (int k = 0; k < 3; k++) pos[k] = pr[k] + pt[k] + 0.5*depth*n2[k];
      add_contact(m, d, ga, gb, pos, normal, -depth);
      kept++;
    }
  }
}

static int filtered(const mco_model* m, int g1, int g2) {
  int b1 = m->geom_body[g1], b2 = m->geom_body[g2];
  int w1 = m->body_weldid[b1], w2 = m->body_weldid[b2];
  if (w1 == 0 && w2 == 0) return 1;
  if (w1 == w2) return 1;
  int p1 = m->body_weldid[m->body_parent[w1]], p2 = m->body_weldid[m->body_parent[w2]];
  if ((w1 != 0 && w2 != 0) && (p1 == w2 || p2 == w1)) return 1;
  for (int e = 0; e < m->nexclude; e++)
    if ((m->exclude[e][0] == b1 && m->exclude[e][1] == b2) || (m->exclude[e][0] == b2 && m->exclude[e][1] == b1)) return 1;
  if (!((m->geom_contype[g1] & m->geom_conaffinity[g2]) || (m->geom_contype[g2] & m->geom_conaffinity[g1]))) return 1;
  return 0;
}

/* --------------------------------------------------------------- ground plane / box <-> collision polytope of a mesh geom */
/* The tables (mycobotgym_amd/model/polytope.py: pack) in geom coordinates, struct-of-arrays, padded to multiples of 64. */
typedef struct { int nv, nf, ne; const double *vx, *vy, *vz, *fx, *fy, *fz, *fd, *e[13]; } poly_view;
static poly_view poly_of(const mco_model* m, int mesh) {
  const double* meta = m->poly + 8 * mesh;
  poly_view P; P.nv = (int)meta[0]; P.nf = (int)meta[1]; P.ne = (int)meta[2];
  const double* b = m->poly + (long)meta[3]; int vp = (int)meta[4], fp = (int)meta[5], ep = (int)meta[6];
  P.vx = b; P.vy = b + vp; P.vz = b + 2*vp; b += 3*vp;
  P.fx = b; P.fy = b + fp; P.fz = b + 2*fp; P.fd = b + 3*fp; b += 4*fp;
  for (int k = 0; k < 13; k++) P.e[k] = b + k*ep;
  return P;
}
static double sgn1(double x) { return x > 0 ? 1.0 : -1.0; }

/* Ground plane: the polytope's lowest vertex (first occurrence), if it is below the plane.  Returns 1 and fills pos / normal / dist. */
static int plane_polytope(const mco_model* m, const mco_data* d, int gp, int gm, double* pos, double* normal, double* dist) {
  poly_view P = poly_of(m, m->geom_poly[gm]);
  const double* R = d->geom_xmat[gm]; const double* p = d->geom_xpos[gm];
  double n[3]; col(d->geom_xmat[gp], 2, n);
  const double* pp = d->geom_xpos[gp];
  double best = INFINITY; int kb = 0;
  for (int k = 0; k < P.nv; k++) {
    double w[3]; for (int r = 0; r < 3; r++) w[r] = p[r] + R[3*r]*P.vx[k] + R[3*r+1]*P.vy[k] + R[3*r+2]*P.vz[k];
    double h = (w[0] - pp[0])*n[0] + (w[1] - pp[1])*n[1] + (w[2] - pp[2])*n[2];
    if (h < best) { best = h; kb = k; }
  }
  if (!(best < 0)) return 0;
  double w[3]; for (int r = 0; r < 3; r++) w[r] = p[r] + R[3*r]*P.vx[kb] + R[3*r+1]*P.vy[kb] + R[3*r+2]*P.vz[kb];
  for (int r = 0; r < 3; r++) { pos[r] = w[r] - 0.5*best*n[r]; normal[r] = n[r]; }
  *dist = best;
  return 1;
}

/* Box <-> polytope, EXACT: the two convex shapes overlap iff no facet normal of their Minkowski difference separates them, and the
 * least of the overlaps along those normals is the penetration depth.  The facet normals are (all in the MESH frame, n pointing from
 * the polytope to the box, sep(n) = min over the box of x.n  -  max over the polytope of v.n, > 0 = separated):
 *   B  the box's face axes, n = +b_j / -b_j (candidate 2 j, 2 j + 1): the polytope's extent along b_j from its vertices;
 *   P  the polytope's face normals n_f: sep = n_f.c - sum_j h_j |n_f.b_j| - d_f;
 *   E  n = +-(e_k x b_j) / |e_k x b_j| for polytope edge k and box axis j (candidate j ne + k), taken only if n lies in the edge's
 *      normal cone (n.u1 >= 0 and n.u2 >= 0: then the edge is the polytope's support set along n; a box always has a supporting edge
 *      along b_j for a direction perpendicular to it), and |e_k x b_j| >= 1e-6.
 * Within a family the largest sep wins, the lowest candidate on ties; between the families the order is B, E, P and a later family
 * replaces an earlier one only if its sep is larger by more than 1e-12 (parallel features: a rounding-level tie must not decide where
 * the contact sits).  There is NO preference of face axes as box-box has one (its 5 %): the witness features of an axis that is not the
 * axis of least penetration need not touch -- a finger link's deepest vertex along a face axis of the cube can lie 1.7 cm beside the
 * 2 cm cube, a polytope face's witness corner of the table 15 cm along the table's edge -- so the axis is the exact minimum, whose
 * witnesses do touch.
 * ONE contact, as MuJoCo's convex-convex pairs have:
 *   B: at the polytope's deepest vertex along n (lowest index), half a depth back;  P: at the box's deepest corner along -n, half a
 *   depth forward;  E: midway between the closest points of the polytope's edge and the box's supporting edge (line parameters
 *   clamped to the two segments).
 * flip = 0: the box is geom1 (the static table), the contact normal points from the box to the mesh; 1: the mesh is geom1. */
#define POLY_TIE 1e-12
static int box_polytope(const mco_model* m, const mco_data* d, int gb, int gm, int flip, double* pos, double* normal, double* dist) {
  poly_view P = poly_of(m, m->geom_poly[gm]);
  const double* Rm = d->geom_xmat[gm]; const double* pm = d->geom_xpos[gm];
  const double* Rb = d->geom_xmat[gb]; const double* pb = d->geom_xpos[gb]; const double* h = m->geom_size[gb];
  double c[3], b[3][3];                                  /* the box in the mesh frame: centre, axes (rows) */
  { double rel[3] = { pb[0] - pm[0], pb[1] - pm[1], pb[2] - pm[2] };
    for (int k = 0; k < 3; k++) c[k] = Rm[k]*rel[0] + Rm[3 + k]*rel[1] + Rm[6 + k]*rel[2];
    for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) b[j][k] = Rm[k]*Rb[j] + Rm[3 + k]*Rb[3 + j] + Rm[6 + k]*Rb[6 + j]; }
  /* B */
  double sB = -INFINITY; int cB = 0, vB = 0;
  for (int j = 0; j < 3; j++) {
    double mx = -INFINITY, mn = INFINITY; int kx = 0, kn = 0;
    for (int k = 0; k < P.nv; k++) {
      double t = P.vx[k]*b[j][0] + P.vy[k]*b[j][1] + P.vz[k]*b[j][2];
      if (t > mx) { mx = t; kx = k; }
      if (t < mn) { mn = t; kn = k; }
    }
    double cb = dot3(c, b[j]);
    double sp = (cb - h[j]) - mx, sn = mn - (cb + h[j]);
    if (sp > sB) { sB = sp; cB = 2*j; vB = kx; }
    if (sn > sB) { sB = sn; cB = 2*j + 1; vB = kn; }
  }
  if (sB > 0) return 0;
  /* P */
  double sP = -INFINITY; int cP = 0;
  for (int f = 0; f < P.nf; f++) {
    double n[3] = { P.fx[f], P.fy[f], P.fz[f] };
    double s = (dot3(n, c) - (h[0]*fabs(dot3(n, b[0])) + h[1]*fabs(dot3(n, b[1])) + h[2]*fabs(dot3(n, b[2])))) - P.fd[f];
    if (s > sP) { sP = s; cP = f; }
  }
  if (sP > 0) return 0;
  /* E */
  double sE = -INFINITY, nE[3] = { 0, 0, 0 }; int jE = 0, kE = 0;
  for (int j = 0; j < 3; j++) {
    int i1 = (j + 1) % 3, i2 = (j + 2) % 3;
    for (int k = 0; k < P.ne; k++) {
      double e[3] = { P.e[3][k], P.e[4][k], P.e[5][k] }, x[3];
      cross3(x, e, b[j]);
      double len = sqrt(dot3(x, x));
      if (len < EDGE_MIN_SIN) continue;
      double il = 1.0 / len; x[0] *= il; x[1] *= il; x[2] *= il;
      double t1 = x[0]*P.e[6][k] + x[1]*P.e[7][k] + x[2]*P.e[8][k], t2 = x[0]*P.e[9][k] + x[1]*P.e[10][k] + x[2]*P.e[11][k];
      double sg = (t1 >= 0 && t2 >= 0) ? 1.0 : ((t1 <= 0 && t2 <= 0) ? -1.0 : 0.0);
      if (sg == 0.0) continue;
      double n[3] = { sg*x[0], sg*x[1], sg*x[2] };
      double s = (dot3(n, c) - (h[i1]*fabs(dot3(n, b[i1])) + h[i2]*fabs(dot3(n, b[i2])))) - (n[0]*P.e[0][k] + n[1]*P.e[1][k] + n[2]*P.e[2][k]);
      if (s > sE) { sE = s; jE = j; kE = k; nE[0] = n[0]; nE[1] = n[1]; nE[2] = n[2]; }
    }
  }
  if (sE > 0) return 0;
  /* the axis of least penetration and its contact (mesh frame) */
  double n[3], q[3], s = sB; int kind = 0;
  if (sE > s + POLY_TIE) { s = sE; kind = 1; }
  if (sP > s + POLY_TIE) { s = sP; kind = 2; }
  if (kind == 1) { n[0] = nE[0]; n[1] = nE[1]; n[2] = nE[2];
    int i1 = (jE + 1) % 3, i2 = (jE + 2) % 3;
    double pe[3] = { P.e[0][kE], P.e[1][kE], P.e[2][kE] }, e[3] = { P.e[3][kE], P.e[4][kE], P.e[5][kE] }, qb[3];
    double g1 = sgn1(dot3(b[i1], n)) * h[i1], g2 = sgn1(dot3(b[i2], n)) * h[i2];
    for (int k = 0; k < 3; k++) qb[k] = c[k] - g1*b[i1][k] - g2*b[i2][k];           /* a point of the box's supporting edge along -n */
    double w[3] = { qb[0] - pe[0], qb[1] - pe[1], qb[2] - pe[2] };
    double uaub = dot3(e, b[jE]), q1 = dot3(e, w), q2 = -dot3(b[jE], w), dd = 1 - uaub*uaub;
    double ts = dd <= 1e-12 ? 0 : (q1 + uaub*q2) / dd, tt = dd <= 1e-12 ? 0 : (uaub*q1 + q2) / dd;
    ts = fmin(fmax(ts, 0.0), P.e[12][kE]); tt = fmin(fmax(tt, -h[jE]), h[jE]);
    for (int k = 0; k < 3; k++) q[k] = 0.5 * ((pe[k] + ts*e[k]) + (qb[k] + tt*b[jE][k]));
  } else if (kind == 2) {
    n[0] = P.fx[cP]; n[1] = P.fy[cP]; n[2] = P.fz[cP];
    double g0 = sgn1(dot3(b[0], n)) * h[0], g1 = sgn1(dot3(b[1], n)) * h[1], g2 = sgn1(dot3(b[2], n)) * h[2];
    for (int k = 0; k < 3; k++) q[k] = (c[k] - g0*b[0][k] - g1*b[1][k] - g2*b[2][k]) - 0.5*s*n[k];      /* the box's deepest corner, half a depth forward */
  } else {
    int j = cB >> 1; double sg = (cB & 1) ? -1.0 : 1.0;
    for (int k = 0; k < 3; k++) n[k] = sg * b[j][k];
    q[0] = P.vx[vB] + 0.5*s*n[0]; q[1] = P.vy[vB] + 0.5*s*n[1]; q[2] = P.vz[vB] + 0.5*s*n[2];             /* the deepest vertex, half a depth back */
  }
  for (int r = 0; r < 3; r++) {
    pos[r] = pm[r] + Rm[3*r]*q[0] + Rm[3*r+1]*q[1] + Rm[3*r+2]*q[2];
    double nw = Rm[3*r]*n[0] + Rm[3*r+1]*n[1] + Rm[3*r+2]*n[2];
    normal[r] = flip ? nw : -nw;
  }
  *dist = s;
  return 1;
}

/* conservative broad phase for the oracle's own speed (it cannot change a result): the polytope's bounding sphere against the box */
static int poly_far_from_box(const mco_model* m, const mco_data* d, int gb, int gm) {
  poly_view P = poly_of(m, m->geom_poly[gm]);
  double lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
  for (int k = 0; k < P.nv; k++) { double v[3] = { P.vx[k], P.vy[k], P.vz[k] }; for (int r = 0; r < 3; r++) { if (v[r] < lo[r]) lo[r] = v[r]; if (v[r] > hi[r]) hi[r] = v[r]; } }
  double cm[3] = { 0.5*(lo[0] + hi[0]), 0.5*(lo[1] + hi[1]), 0.5*(lo[2] + hi[2]) }, rad = 0.5 * sqrt((hi[0]-lo[0])*(hi[0]-lo[0]) + (hi[1]-lo[1])*(hi[1]-lo[1]) + (hi[2]-lo[2])*(hi[2]-lo[2]));
  const double* Rm = d->geom_xmat[gm]; const double* pm = d->geom_xpos[gm];
  const double* Rb = d->geom_xmat[gb]; const double* pb = d->geom_xpos[gb]; const double* h = m->geom_size[gb];
  double w[3]; for (int r = 0; r < 3; r++) w[r] = pm[r] + Rm[3*r]*cm[0] + Rm[3*r+1]*cm[1] + Rm[3*r+2]*cm[2] - pb[r];
  double d2 = 0;
  for (int a = 0; a < 3; a++) { double t = fabs(Rb[a]*w[0] + Rb[3 + a]*w[1] + Rb[6 + a]*w[2]) - h[a]; if (t > 0) d2 += t*t; }
  return d2 > (rad + 1e-6) * (rad + 1e-6);
}

/* census of the list lengths an uncapped run would need (tools / tests: how often does the kernels' cap of 16 entries cut?) */
long mco_entry_hist[MCO_MAXCON + 1];
static void collide(const mco_model* m, mco_data* d);
void mco_collision(const mco_model* m, mco_data* d) {
  collide(m, d);
  int k = d->nentry + d->ndrop; if (k > MCO_MAXCON) k = MCO_MAXCON;
  __atomic_fetch_add(&mco_entry_hist[k], 1, __ATOMIC_RELAXED);
}
static void collide(const mco_model* m, mco_data* d) {
  d->ncon = 0; d->nentry = 0; d->ndrop = 0;
  /* the primitive geoms, in geom order: ground - pads, ground - cube, table - pads, table - cube, pads - cube (the order the kernels emit them in) */
  for (int g1 = 0; g1 < m->ngeom; g1++) for (int g2 = g1 + 1; g2 < m->ngeom; g2++) {
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    if (t1 == MCO_GEOM_MESH || t2 == MCO_GEOM_MESH) continue;
    if (m->collide_scope_geom >= 0 && g1 != m->collide_scope_geom && g2 != m->collide_scope_geom
        && !(m->collide_extra[g1] && m->collide_extra[g2] && m->collide_extra[g1] != m->collide_extra[g2])) continue;
    if (filtered(m, g1, g2)) continue;
    if (t1 == MCO_GEOM_PLANE && t2 == MCO_GEOM_BOX) plane_box(m, d, g1, g2);
    else if (t1 == MCO_GEOM_BOX && t2 == MCO_GEOM_BOX) {
      const double* s1 = m->geom_size[g1]; const double* s2 = m->geom_size[g2];
      double r = sqrt(dot3(s1, s1)) + sqrt(dot3(s2, s2));
      double dp[3] = { d->geom_xpos[g2][0] - d->geom_xpos[g1][0], d->geom_xpos[g2][1] - d->geom_xpos[g1][1], d->geom_xpos[g2][2] - d->geom_xpos[g1][2] };
      if (dot3(dp, dp) > r * r) continue;                               /* bounding spheres */
      box_box(m, d, g1, g2);
    }
  }
  /* the mesh geoms, polytope by polytope (links 1-6, flange, gripper base, gear / finger links right and left, hinge links), each against
   * the ground plane, the static boxes (the table) and the scope geom (the cube), in that order.  The reference attaches every mesh
   * twice (a visual and a collision geom of equal pose, both colliding): the twins' identical contacts follow each other and form ONE
   * entry of the list. */
  if (!m->poly) return;
  for (int mi = 0; mi < MCO_NMESH; mi++) {
    int tw[4], ntw = 0;
    for (int g = 0; g < m->ngeom && ntw < 4; g++) if (m->geom_type[g] == MCO_GEOM_MESH && m->geom_poly[g] == mi) tw[ntw++] = g;
    if (!ntw) continue;
    for (int pass = 0; pass < 2; pass++) for (int go = 0; go < m->ngeom; go++) {
      if (pass == 0 ? (m->collide_extra[go] != 1) : (go != m->collide_scope_geom)) continue;
      if (m->geom_type[go] != MCO_GEOM_PLANE && m->geom_type[go] != MCO_GEOM_BOX) continue;
      int gm = tw[0];
      double pos[3], nrm[3], dist; int hit;
      if (m->geom_type[go] == MCO_GEOM_PLANE) hit = plane_polytope(m, d, go, gm, pos, nrm, &dist);
      else hit = poly_far_from_box(m, d, go, gm) ? 0 : box_polytope(m, d, go, gm, gm < go, pos, nrm, &dist);
      if (!hit) continue;
      int live = 0;
      for (int t = 0; t < ntw; t++) live += !filtered(m, go, tw[t]);
      if (!live) continue;
      if (d->nentry >= m->maxentry || d->ncon + live > MCO_MAXCON) { d->ndrop += live; continue; }
      int keep = d->nentry;
      for (int t = 0; t < ntw; t++) {
        if (filtered(m, go, tw[t])) continue;
        d->nentry = keep;                                                /* the twins share one entry */
        if (tw[t] < go) add_contact(m, d, tw[t], go, pos, nrm, dist); else add_contact(m, d, go, tw[t], pos, nrm, dist);
      }
    }
  }
}
