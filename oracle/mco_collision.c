/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see mco_physics.h).
 *
 * P4 `mj_collision` restated for the primitive geoms of the PickAndPlace scene: the cube, the table box, the two
 * finger-pad boxes and the ground plane (/root/reference/mycobotgym/envs/assets/mycobot280_main.xml:81,87,195-199,
 * 222-225,262).  The 28 mesh geoms (convex hulls in MuJoCo) are outside this build's scope and never
 * produce contacts here.  With `collide_scope_geom` set (the build's configuration) only pairs involving the cube are
 * tested: pads touching the table or each other would, in MuJoCo, be preceded by finger-mesh contacts that are out
 * of scope anyway, so modelling them without the meshes is not closer to the reference than leaving them out.
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
  if (!(dist < 0) || d->ncon >= MCO_MAXCON) return;
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
  double pa[3], pb[3];                  /* box centres (a mesh geom's bounding box is off its frame origin by obb_center, else zero) */
  for (int r = 0; r < 3; r++) {
    pa[r] = d->geom_xpos[ga][r] + Ra[3*r]*m->obb_center[ga][0] + Ra[3*r+1]*m->obb_center[ga][1] + Ra[3*r+2]*m->obb_center[ga][2];
    pb[r] = d->geom_xpos[gb][r] + Rb[3*r]*m->obb_center[gb][0] + Rb[3*r+1]*m->obb_center[gb][1] + Rb[3*r+2]*m->obb_center[gb][2];
  }
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

/* ------------------------------------------------------------------- static box / plane - support polytope of a mesh */
static void world_vertex(const mco_model* m, const mco_data* d, int g, int k, double* w) {
  const double* R = d->geom_xmat[g]; const double* p = d->geom_xpos[g]; const double* v = m->hull_vert[g][k];
  for (int r = 0; r < 3; r++) w[r] = p[r] + R[3*r]*v[0] + R[3*r+1]*v[1] + R[3*r+2]*v[2];
}
static void plane_polytope(const mco_model* m, mco_data* d, int gp, int gm) {
  double n[3]; col(d->geom_xmat[gp], 2, n);
  const double* pp = d->geom_xpos[gp];
  double best = 0, bw[3] = {0, 0, 0}; int found = 0;
  for (int k = 0; k < m->hull_nvert[gm]; k++) {
    double w[3]; world_vertex(m, d, gm, k, w);
    double rel[3] = { w[0] - pp[0], w[1] - pp[1], w[2] - pp[2] };
    double dist = dot3(rel, n);
    if (dist < 0 && (!found || dist < best)) { best = dist; memcpy(bw, w, sizeof(bw)); found = 1; }
  }
  if (!found) return;
  double pos[3] = { bw[0] - 0.5*best*n[0], bw[1] - 0.5*best*n[1], bw[2] - 0.5*best*n[2] };
  add_contact(m, d, gp, gm, pos, n, best);
}
/* Box <-> support polytope of a mesh.  Separating-axis test over the box's three face axes and the polytope's own 13 canonical axes
 * (its frame's axes, face diagonals and space diagonals: the directions its vertices are support points of -- together they bound the
 * polytope by its 26-DOP).  Without the 13 a polytope diagonally off an edge of the box counts as touching whenever its box-aligned
 * extent overlaps the box (round 2: a finger 2.4 cm from the cube's centre; a link beside the table's edge).  ONE contact:
 *   full = 0 (the static table): along the box FACE of least penetration, at the polytope's deepest vertex;
 *   full = 1 (the cube):         along the axis of least penetration among all 16; for a polytope axis the contact sits at the box's
 *                                deepest corner along it.
 * flip: the mesh is geom1 and the box geom2 (a gripper mesh against the cube, which comes later in geom order): the contact's normal
 * then points from the mesh to the box. */
static const int DIR13[13][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, -1, 0}, {1, 0, 1}, {1, 0, -1}, {0, 1, 1}, {0, 1, -1}, {1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {1, -1, -1}};
static void box_polytope(const mco_model* m, mco_data* d, int gb, int gm, int full, int flip) {
  const double* pb = d->geom_xpos[gb]; const double* Rb = d->geom_xmat[gb]; const double* h = m->geom_size[gb];
  double lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
  double wlo[3][3], whi[3][3];                      /* the vertices that realise the extremes (first occurrence) */
  for (int k = 0; k < m->hull_nvert[gm]; k++) {
    double w[3]; world_vertex(m, d, gm, k, w);
    double rel[3] = { w[0] - pb[0], w[1] - pb[1], w[2] - pb[2] };
    for (int a = 0; a < 3; a++) {
      double ax[3]; col(Rb, a, ax);
      double c = dot3(rel, ax);
      if (c < lo[a]) { lo[a] = c; memcpy(wlo[a], w, sizeof(w)); }
      if (c > hi[a]) { hi[a] = c; memcpy(whi[a], w, sizeof(w)); }
    }
  }
  for (int a = 0; a < 3; a++) if (lo[a] > h[a] || hi[a] < -h[a]) return;      /* a face axis separates */
  double depth = INFINITY; int axis = 0, sign = 1; double n[3] = { 0, 0, 0 };
  for (int a = 0; a < 3; a++) {
    double dp = h[a] - lo[a], dn = hi[a] + h[a];    /* push the polytope out through face +a / -a */
    if (dp < depth) { depth = dp; axis = a; sign = 1; }
    if (dn < depth) { depth = dn; axis = a; sign = -1; }
  }
  if (!m->rule[7]) {      /* study switch rule[7] = 1: the face axes alone (the first version of this test) */
    const double* Rm = d->geom_xmat[gm]; const double* pm = d->geom_xpos[gm];
    for (int k = 0; k < 13; k++) {
      double mn = INFINITY, mx = -INFINITY;
      for (int v = 0; v < m->hull_nvert[gm]; v++) {
        const double* hv = m->hull_vert[gm][v];
        double c = DIR13[k][0]*hv[0] + DIR13[k][1]*hv[1] + DIR13[k][2]*hv[2];
        if (c < mn) mn = c;
        if (c > mx) mx = c;
      }
      double w[3];
      for (int r = 0; r < 3; r++) w[r] = Rm[3*r]*DIR13[k][0] + Rm[3*r+1]*DIR13[k][1] + Rm[3*r+2]*DIR13[k][2];
      double rel = (pm[0] - pb[0])*w[0] + (pm[1] - pb[1])*w[1] + (pm[2] - pb[2])*w[2];      /* polytope origin - box centre, along w */
      double rad = 0;
      for (int a = 0; a < 3; a++) { double ax[3]; col(Rb, a, ax); rad += h[a] * fabs(dot3(ax, w)); }
      if (rel + mn > rad || rel + mx < -rad) return;
      if (full) {
        int l2 = DIR13[k][0]*DIR13[k][0] + DIR13[k][1]*DIR13[k][1] + DIR13[k][2]*DIR13[k][2];
        double il = 1.0 / (l2 == 1 ? 1.0 : (l2 == 2 ? 1.4142135623730951 : 1.7320508075688772));      /* the kernels' constants */
        double dp = (rad - (rel + mn)) * il, dn = ((rel + mx) + rad) * il;        /* push the polytope out along +w / -w */
        if (dp < depth) { depth = dp; axis = 3 + k; sign = 1; for (int r = 0; r < 3; r++) n[r] = w[r] * il; }
        if (dn < depth) { depth = dn; axis = 3 + k; sign = -1; for (int r = 0; r < 3; r++) n[r] = -w[r] * il; }
      }
    }
  }
  double pos[3];
  if (axis < 3) {
    col(Rb, axis, n); for (int k = 0; k < 3; k++) n[k] *= sign;               /* from the box to the mesh */
    const double* w = sign > 0 ? wlo[axis] : whi[axis];                        /* the vertex deepest inside */
    for (int k = 0; k < 3; k++) pos[k] = w[k] + 0.5*depth*n[k];
  } else {                                                                     /* the box's deepest corner along n, half a depth back */
    double xb[3] = { pb[0], pb[1], pb[2] };
    for (int a = 0; a < 3; a++) { double ax[3]; col(Rb, a, ax); double sg = dot3(n, ax) > 0 ? 1.0 : -1.0; for (int k = 0; k < 3; k++) xb[k] += sg * h[a] * ax[k]; }
    for (int k = 0; k < 3; k++) pos[k] = xb[k] - 0.5*depth*n[k];
  }
  if (flip) { double nn[3] = { -n[0], -n[1], -n[2] }; add_contact(m, d, gm, gb, pos, nn, -depth); }
  else add_contact(m, d, gb, gm, pos, n, -depth);
}

void mco_collision(const mco_model* m, mco_data* d) {
  d->ncon = 0;
  /* static primitive <-> arm-side mesh (support polytope): these pairs first, mesh by mesh (the order the kernels emit them in) */
  for (int g2 = 0; g2 < m->ngeom; g2++) {
    if (m->geom_type[g2] != MCO_GEOM_MESH || m->hull_nvert[g2] <= 0 || (m->collide_extra[g2] != 3 && m->collide_extra[g2] != 5)) continue;
    for (int g1 = 0; g1 < m->ngeom; g1++) {
      if (m->collide_extra[g1] != 1 || filtered(m, g1, g2)) continue;
      if (m->geom_type[g1] == MCO_GEOM_PLANE) plane_polytope(m, d, g1, g2);
      else if (m->geom_type[g1] == MCO_GEOM_BOX) box_polytope(m, d, g1, g2, 0, 0);
    }
  }
  for (int g1 = 0; g1 < m->ngeom; g1++) for (int g2 = g1 + 1; g2 < m->ngeom; g2++) {
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    if (t1 == MCO_GEOM_MESH || t2 == MCO_GEOM_MESH) continue;           /* convex-mesh collision: out of scope */
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
  /* finger-link meshes (collide_extra 4; 5 = a mesh that also collides with the static geoms above: the rule supports the gripper base
   * that way, the build does not enable it -- a fourth contact class costs the kernels' coupled solve 50 %) <-> the cube (SURVEY 8f-4,
   * second stage), after the primitive pairs, geom by geom (the reference attaches every mesh twice: the twin's contact follows at once) */
  if (m->collide_scope_geom >= 0 && m->geom_type[m->collide_scope_geom] == MCO_GEOM_BOX) {
    int gc = m->collide_scope_geom;
    for (int g = 0; g < m->ngeom; g++) {
      if (m->geom_type[g] != MCO_GEOM_MESH || m->hull_nvert[g] <= 0 || (m->collide_extra[g] != 4 && m->collide_extra[g] != 5) || filtered(m, g, gc)) continue;
      box_polytope(m, d, gc, g, 1, g < gc);
    }
  }
}
