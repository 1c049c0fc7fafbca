/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see mco_physics.h for scope and the "parity unpinned" note).
 *
 * Restates, stage by stage, what `mujoco.mj_step` does for the MyCobot model class
 * (call sites /root/reference/mycobotgym/envs/mycobot.py:170,189,193; `mj_forward` :213,229,453).
 * Stage names follow SURVEY.md section 8(a) rows P1-P12 and Appendix B [RECALL MuJoCo 2.3.2].
 * Written for clarity, not speed: dense matrices, generic loops over all bodies (no welding,
 * no specialisation) -- the HIP kernels are the specialised form and are checked against this.
 */
#include "mco_physics.h"

#include <math.h>
#include <stddef.h>
#include <string.h>

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999
#define PI 3.14159265358979323846

/* --------------------------------------------------------------------------- small vectors */
static void zero(double* a, int n) { for (int i = 0; i < n; i++) a[i] = 0.0; }
static void copy(double* r, const double* a, int n) { for (int i = 0; i < n; i++) r[i] = a[i]; }
static double dot3(const double* a, const double* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static double dotn(const double* a, const double* b, int n) { double s = 0; for (int i = 0; i < n; i++) s += a[i]*b[i]; return s; }
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double normalize3(double* v) {           /* mju_normalize3 */
  double n = sqrt(dot3(v, v));
  if (n < MINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; } else { v[0] /= n; v[1] /= n; v[2] /= n; }
  return n;
}
static void normalize4(double* q) {             /* mju_normalize4 */
  double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { for (int i = 0; i < 4; i++) q[i] /= n; }
}
static void mat_vec3(double* r, const double* M, const double* v) {      /* r = M v, M row-major 3x3 */
  double x = M[0]*v[0] + M[1]*v[1] + M[2]*v[2], y = M[3]*v[0] + M[4]*v[1] + M[5]*v[2],
         z = M[6]*v[0] + M[7]*v[1] + M[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void matT_vec3(double* r, const double* M, const double* v) {     /* r = M^T v */
  double x = M[0]*v[0] + M[3]*v[1] + M[6]*v[2], y = M[1]*v[0] + M[4]*v[1] + M[7]*v[2],
         z = M[2]*v[0] + M[5]*v[1] + M[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}

/* ------------------------------------------------------------------------------ quaternions */
void mco_mulquat(double* r, const double* a, const double* b) {          /* mju_mulQuat */
  double t[4] = { a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3],
                  a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2],
                  a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1],
                  a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0] };
  copy(r, t, 4);
}
void mco_negquat(double* r, const double* q) { r[0] = q[0]; r[1] = -q[1]; r[2] = -q[2]; r[3] = -q[3]; }
void mco_quat2mat(double* m, const double* q) {                          /* mju_quat2Mat */
  double q00 = q[0]*q[0], q01 = q[0]*q[1], q02 = q[0]*q[2], q03 = q[0]*q[3];
  double q11 = q[1]*q[1], q12 = q[1]*q[2], q13 = q[1]*q[3], q22 = q[2]*q[2], q23 = q[2]*q[3], q33 = q[3]*q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2*(q12 - q03); m[2] = 2*(q13 + q02); m[3] = 2*(q12 + q03);
  m[5] = 2*(q23 - q01); m[6] = 2*(q13 - q02); m[7] = 2*(q23 + q01);
}
static void rot_vec_quat(double* r, const double* v, const double* q) {  /* mju_rotVecQuat */
  double m[9]; mco_quat2mat(m, q); mat_vec3(r, m, v);
}
static void axisangle2quat(double* q, const double* axis, double angle) {
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
void mco_mat2quat(double* quat, const double* mat) {                     /* mju_mat2Quat [RECALL] */
  if (mat[0] + mat[4] + mat[8] > 0) {
    quat[0] = 0.5 * sqrt(1 + mat[0] + mat[4] + mat[8]);
    quat[1] = 0.25 * (mat[7] - mat[5]) / quat[0];
    quat[2] = 0.25 * (mat[2] - mat[6]) / quat[0];
    quat[3] = 0.25 * (mat[3] - mat[1]) / quat[0];
  } else if (mat[0] > mat[4] && mat[0] > mat[8]) {
    quat[1] = 0.5 * sqrt(1 + mat[0] - mat[4] - mat[8]);
    quat[0] = 0.25 * (mat[7] - mat[5]) / quat[1];
    quat[2] = 0.25 * (mat[1] + mat[3]) / quat[1];
    quat[3] = 0.25 * (mat[2] + mat[6]) / quat[1];
  } else if (mat[4] > mat[8]) {
    quat[2] = 0.5 * sqrt(1 - mat[0] + mat[4] - mat[8]);
    quat[0] = 0.25 * (mat[2] - mat[6]) / quat[2];
    quat[1] = 0.25 * (mat[1] + mat[3]) / quat[2];
    quat[3] = 0.25 * (mat[5] + mat[7]) / quat[2];
  } else {
    quat[3] = 0.5 * sqrt(1 - mat[0] - mat[4] + mat[8]);
    quat[0] = 0.25 * (mat[3] - mat[1]) / quat[3];
    quat[1] = 0.25 * (mat[2] + mat[6]) / quat[3];
    quat[2] = 0.25 * (mat[5] + mat[7]) / quat[3];
  }
  normalize4(quat);
}
void mco_quat2vel(double* res, const double* quat, double dt) {          /* mju_quat2Vel [RECALL] */
  double axis[3] = { quat[1], quat[2], quat[3] };
  double sin_a_2 = normalize3(axis);
  double speed = 2 * atan2(sin_a_2, quat[0]);
  if (speed > PI) speed -= 2 * PI;
  speed /= dt;
  res[0] = axis[0]*speed; res[1] = axis[1]*speed; res[2] = axis[2]*speed;
}
static void quat_integrate(double* quat, const double* vel, double scale) {  /* mju_quatIntegrate */
  double tmp[3] = { vel[0], vel[1], vel[2] }, qrot[4];
  double angle = scale * normalize3(tmp);
  axisangle2quat(qrot, tmp, angle);
  normalize4(quat);
  mco_mulquat(quat, quat, qrot);
}

/* --------------------------------------------------- spatial algebra (MuJoCo conventions) */
/* 6-vectors are [rotational; translational]; the 10-number inertia is
   [Ixx Iyy Izz Ixy Ixz Iyz | m*dx m*dy m*dz | m] about a reference point, d = com - ref. */
static void inert_com(double* res, const double* inert, const double* mat, const double* dif, double mass) {
  /* mju_inertCom */
  double tmp[9];   /* mat * diag(inert) * mat^T */
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
    tmp[3*i+j] = mat[3*i]*inert[0]*mat[3*j] + mat[3*i+1]*inert[1]*mat[3*j+1] + mat[3*i+2]*inert[2]*mat[3*j+2];
  res[0] = tmp[0] + mass*(dif[1]*dif[1] + dif[2]*dif[2]);
  res[1] = tmp[4] + mass*(dif[0]*dif[0] + dif[2]*dif[2]);
  res[2] = tmp[8] + mass*(dif[0]*dif[0] + dif[1]*dif[1]);
  res[3] = tmp[1] - mass*dif[0]*dif[1];
  res[4] = tmp[2] - mass*dif[0]*dif[2];
  res[5] = tmp[5] - mass*dif[1]*dif[2];
  res[6] = mass*dif[0]; res[7] = mass*dif[1]; res[8] = mass*dif[2]; res[9] = mass;
}
static void mul_inert_vec(double* res, const double* i, const double* v) {  /* mju_mulInertVec */
  res[0] = i[0]*v[0] + i[3]*v[1] + i[4]*v[2] - i[8]*v[4] + i[7]*v[5];
  res[1] = i[3]*v[0] + i[1]*v[1] + i[5]*v[2] + i[8]*v[3] - i[6]*v[5];
  res[2] = i[4]*v[0] + i[5]*v[1] + i[2]*v[2] - i[7]*v[3] + i[6]*v[4];
  res[3] = i[8]*v[1] - i[7]*v[2] + i[9]*v[3];
  res[4] = i[6]*v[2] - i[8]*v[0] + i[9]*v[4];
  res[5] = i[7]*v[0] - i[6]*v[1] + i[9]*v[5];
}
static void cross_motion(double* res, const double* vel, const double* v) { /* mju_crossMotion */
  double a[3], b[3];
  cross3(res, vel, v);
  cross3(a, vel, v + 3); cross3(b, vel + 3, v);
  res[3] = a[0] + b[0]; res[4] = a[1] + b[1]; res[5] = a[2] + b[2];
}
static void cross_force(double* res, const double* vel, const double* f) {  /* mju_crossForce */
  double a[3], b[3];
  cross3(a, vel, f); cross3(b, vel + 3, f + 3);
  res[0] = a[0] + b[0]; res[1] = a[1] + b[1]; res[2] = a[2] + b[2];
  cross3(res + 3, vel, f + 3);
}

/* ----------------------------------------------------------------- dense SPD linear algebra */
static int cholesky(int n, double A[MCO_MAXNV][MCO_MAXNV], double L[MCO_MAXNV][MCO_MAXNV]) {
  for (int i = 0; i < n; i++) {
    for (int j = 0; j <= i; j++) {
      double s = A[i][j];
      for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
      if (i == j) { if (s <= 0) return -1; L[i][i] = sqrt(s); }
      else L[i][j] = s / L[j][j];
    }
    for (int j = i + 1; j < n; j++) L[i][j] = 0;
  }
  return 0;
}
static void chol_solve(int n, double L[MCO_MAXNV][MCO_MAXNV], double* x) {
  for (int i = 0; i < n; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= L[i][k]*x[k]; x[i] = s / L[i][i]; }
  for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k][i]*x[k]; x[i] = s / L[i][i]; }
}

/* ============================================================ P1: mj_kinematics + mj_comPos */
static void kinematics(const mco_model* m, mco_data* d) {
  zero(d->xpos[0], 3); d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  mco_quat2mat(d->xmat[0], d->xquat[0]);
  copy(d->xipos[0], d->xpos[0], 3); copy(d->ximat[0], d->xmat[0], 9);
  for (int i = 1; i < m->nbody; i++) {
    double xpos[3], xquat[4], vec[3];
    int pid = m->body_parent[i];
    int jfirst = -1, jnum = 0;
    for (int j = 0; j < m->njnt; j++) if (m->jnt_body[j] == i) { if (jfirst < 0) jfirst = j; jnum++; }
    if (m->body_mocapid[i] >= 0) {                    /* mocap body: pose from data, quaternion normalised in place */
      int mid = m->body_mocapid[i];                   /* [RECALL mj_kinematics: "normalize all quaternions in qpos and mocap_quat"] */
      normalize4(d->mocap_quat[mid]);
      copy(xpos, d->mocap_pos[mid], 3); copy(xquat, d->mocap_quat[mid], 4);
    } else if (jnum == 1 && m->jnt_type[jfirst] == MCO_JNT_FREE) {
      int qa = m->jnt_qposadr[jfirst];
      copy(xpos, d->qpos + qa, 3);
      normalize4(d->qpos + qa + 3);                 /* MuJoCo normalises the stored quaternion */
      copy(xquat, d->qpos + qa + 3, 4);
      copy(d->xanchor[jfirst], xpos, 3);
      d->xaxis[jfirst][0] = 0; d->xaxis[jfirst][1] = 0; d->xaxis[jfirst][2] = 1;
    } else {
      mat_vec3(vec, d->xmat[pid], m->body_pos[i]);
      for (int k = 0; k < 3; k++) xpos[k] = d->xpos[pid][k] + vec[k];
      mco_mulquat(xquat, d->xquat[pid], m->body_quat[i]);
      for (int j = jfirst; j >= 0 && j < jfirst + jnum; j++) {
        double qloc[4];
        rot_vec_quat(vec, m->jnt_pos[j], xquat);
        for (int k = 0; k < 3; k++) d->xanchor[j][k] = vec[k] + xpos[k];
        rot_vec_quat(d->xaxis[j], m->jnt_axis[j], xquat);
        /* hinge: rotate about the local axis, then correct for an off-centre anchor */
        axisangle2quat(qloc, m->jnt_axis[j], d->qpos[m->jnt_qposadr[j]] - m->qpos0[m->jnt_qposadr[j]]);
        mco_mulquat(xquat, xquat, qloc);
        rot_vec_quat(vec, m->jnt_pos[j], xquat);
        for (int k = 0; k < 3; k++) xpos[k] = d->xanchor[j][k] - vec[k];
      }
    }
    normalize4(xquat);
    copy(d->xquat[i], xquat, 4); copy(d->xpos[i], xpos, 3);
    mco_quat2mat(d->xmat[i], xquat);
    {
      double q[4];
      mat_vec3(vec, d->xmat[i], m->body_ipos[i]);
      for (int k = 0; k < 3; k++) d->xipos[i][k] = xpos[k] + vec[k];
      mco_mulquat(q, xquat, m->body_iquat[i]);
      mco_quat2mat(d->ximat[i], q);
    }
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_body[g]; double vec[3], q[4];
    mat_vec3(vec, d->xmat[b], m->geom_pos[g]);
    for (int k = 0; k < 3; k++) d->geom_xpos[g][k] = d->xpos[b][k] + vec[k];
    mco_mulquat(q, d->xquat[b], m->geom_quat[g]); mco_quat2mat(d->geom_xmat[g], q);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_body[s]; double vec[3], q[4];
    mat_vec3(vec, d->xmat[b], m->site_pos[s]);
    for (int k = 0; k < 3; k++) d->site_xpos[s][k] = d->xpos[b][k] + vec[k];
    mco_mulquat(q, d->xquat[b], m->site_quat[s]); mco_quat2mat(d->site_xmat[s], q);
  }
}

static void com_pos(const mco_model* m, mco_data* d) {
  double mass_subtree[MCO_MAXBODY];
  for (int i = 0; i < m->nbody; i++) {
    for (int k = 0; k < 3; k++) d->subtree_com[i][k] = m->body_mass[i] * d->xipos[i][k];
    mass_subtree[i] = m->body_mass[i];
  }
  for (int i = m->nbody - 1; i > 0; i--) {
    int p = m->body_parent[i];
    for (int k = 0; k < 3; k++) d->subtree_com[p][k] += d->subtree_com[i][k];
    mass_subtree[p] += mass_subtree[i];
  }
  for (int i = 0; i < m->nbody; i++) {
    if (mass_subtree[i] < MINVAL) copy(d->subtree_com[i], d->xipos[i], 3);
    else for (int k = 0; k < 3; k++) d->subtree_com[i][k] /= mass_subtree[i];
  }
  /* cinert about the kinematic-tree root's subtree CoM */
  zero(d->cinert[0], 10);
  for (int i = 1; i < m->nbody; i++) {
    double dif[3];
    for (int k = 0; k < 3; k++) dif[k] = d->xipos[i][k] - d->subtree_com[m->body_rootid[i]][k];
    inert_com(d->cinert[i], m->body_inertia[i], d->ximat[i], dif, m->body_mass[i]);
  }
  /* cdof: motion axis of every dof, referenced to the same point */
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_body[j], da = m->jnt_dofadr[j];
    double off[3];
    for (int k = 0; k < 3; k++) off[k] = d->subtree_com[m->body_rootid[b]][k] - d->xanchor[j][k];
    if (m->jnt_type[j] == MCO_JNT_FREE) {
      for (int k = 0; k < 3; k++) { zero(d->cdof[da + k], 6); d->cdof[da + k][3 + k] = 1; }
      for (int k = 0; k < 3; k++) {          /* rotation about the body-local axes */
        double ax[3] = { d->xmat[b][k], d->xmat[b][3 + k], d->xmat[b][6 + k] };
        copy(d->cdof[da + 3 + k], ax, 3); cross3(d->cdof[da + 3 + k] + 3, ax, off);
      }
    } else {
      copy(d->cdof[da], d->xaxis[j], 3); cross3(d->cdof[da] + 3, d->xaxis[j], off);
    }
  }
}

/* ================================================================ P2: mj_tendon, mj_transmission */
static void tendon_transmission(const mco_model* m, mco_data* d) {
  for (int t = 0; t < m->ntendon; t++) {
    d->ten_length[t] = 0; zero(d->ten_J[t], m->nv);
    for (int k = 0; k < m->ten_num[t]; k++) {
      int j = m->ten_jnt[t][k];
      d->ten_length[t] += m->ten_coef[t][k] * d->qpos[m->jnt_qposadr[j]];
      d->ten_J[t][m->jnt_dofadr[j]] = m->ten_coef[t][k];
    }
  }
  for (int u = 0; u < m->nu; u++) {
    zero(d->act_moment[u], m->nv);
    if (m->act_trntype[u] == 0) {
      int j = m->act_trnid[u];
      d->act_length[u] = m->act_gear[u] * d->qpos[m->jnt_qposadr[j]];
      d->act_moment[u][m->jnt_dofadr[j]] = m->act_gear[u];
    } else {
      int t = m->act_trnid[u];
      d->act_length[u] = m->act_gear[u] * d->ten_length[t];
      for (int k = 0; k < m->nv; k++) d->act_moment[u][k] = m->act_gear[u] * d->ten_J[t][k];
    }
  }
}

/* ======================================================================= P3: mj_crb + mj_factorM */
static void crb_factor(const mco_model* m, mco_data* d) {
  int nv = m->nv;
  for (int i = 0; i < m->nbody; i++) copy(d->crb[i], d->cinert[i], 10);
  for (int i = m->nbody - 1; i > 0; i--) {
    int p = m->body_parent[i];
    if (p > 0) for (int k = 0; k < 10; k++) d->crb[p][k] += d->crb[i][k];
  }
  for (int i = 0; i < nv; i++) zero(d->qM[i], nv);
  for (int i = 0; i < nv; i++) {
    double buf[6];
    mul_inert_vec(buf, d->crb[m->dof_body[i]], d->cdof[i]);
    for (int j = i; j >= 0; j = m->dof_parent[j]) {
      d->qM[i][j] = d->qM[j][i] = dotn(d->cdof[j], buf, 6);
    }
    d->qM[i][i] += m->dof_armature[i];
  }
  cholesky(nv, d->qM, d->qL);
}

/* ================================================================================== mj_jac */
void mco_jac(const mco_model* m, const mco_data* d, double* jacp, double* jacr, const double point[3], int body) {
  int nv = m->nv;
  if (jacp) zero(jacp, 3 * nv);
  if (jacr) zero(jacr, 3 * nv);
  double off[3];
  for (int k = 0; k < 3; k++) off[k] = point[k] - d->subtree_com[m->body_rootid[body]][k];
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parent[body];
  if (body == 0) return;
  for (int i = m->body_dofadr[body] + m->body_dofnum[body] - 1; i >= 0; i = m->dof_parent[i]) {
    double tmp[3];
    cross3(tmp, d->cdof[i], off);
    for (int k = 0; k < 3; k++) {
      if (jacr) jacr[k * nv + i] = d->cdof[i][k];
      if (jacp) jacp[k * nv + i] = d->cdof[i][3 + k] + tmp[k];
    }
  }
}
void mco_jac_site(const mco_model* m, const mco_data* d, double* jacp, double* jacr, int site) {
  mco_jac(m, d, jacp, jacr, d->site_xpos[site], m->site_body[site]);
}

/* ============================================================================ P4: mj_collision */
/* implemented in mco_collision.c */
void mco_collision(const mco_model* m, mco_data* d);

/* =================================================== P5: mj_makeConstraint + mj_projectConstraint */
static int add_row(mco_data* d, int nv, const double* J, double pos, double margin, int type, int id) {
  int i = d->nefc;
  if (i >= MCO_MAXEFC) return -1;
  copy(d->efc_J[i], J, nv);
  d->efc_pos[i] = pos; d->efc_margin[i] = margin; d->efc_type[i] = type; d->efc_id[i] = id;
  d->nefc++;
  return i;
}

static void get_solparam(const mco_model* m, const mco_data* d, int i, double* solref, double* solimp) {
  int id = d->efc_id[i];
  switch (d->efc_type[i]) {
    case MCO_EFC_EQUALITY: copy(solref, m->eq_solref[id], 2); copy(solimp, m->eq_solimp[id], 5); break;
    case MCO_EFC_LIMIT:    copy(solref, m->jnt_solref[id], 2); copy(solimp, m->jnt_solimp[id], 5); break;
    default:               copy(solref, d->contact[id].solref, 2); copy(solimp, d->contact[id].solimp, 5); break;
  }
  /* refsafe: a positive time constant may not be below 2 timesteps */
  if (solref[0] > 0 && solref[0] < 2 * m->timestep) solref[0] = 2 * m->timestep;
  if (solimp[0] < MINIMP) solimp[0] = MINIMP; if (solimp[0] > MAXIMP) solimp[0] = MAXIMP;
  if (solimp[1] < MINIMP) solimp[1] = MINIMP; if (solimp[1] > MAXIMP) solimp[1] = MAXIMP;
  if (solimp[2] < 0) solimp[2] = 0;
  if (solimp[3] < MINIMP) solimp[3] = MINIMP; if (solimp[3] > MAXIMP) solimp[3] = MAXIMP;
  if (solimp[4] < 1) solimp[4] = 1;
}

static double get_impedance(const double* solimp, double pos, double margin) {
  if (solimp[0] == solimp[1] || solimp[2] <= MINVAL) return 0.5 * (solimp[0] + solimp[1]);
  double x = (pos - margin) / solimp[2];
  if (x < 0) x = -x;
  if (x >= 1) return solimp[1];
  if (x == 0) return solimp[0];
  double y, mid = solimp[3], pw = solimp[4];
  if (pw == 1) y = x;
  else if (x <= mid) y = pow(x, pw) / pow(mid, pw - 1);
  else y = 1 - pow(1 - x, pw) / pow(1 - mid, pw - 1);
  return solimp[0] + y * (solimp[1] - solimp[0]);
}

static void make_constraint(const mco_model* m, mco_data* d) {
  int nv = m->nv;
  double J[MCO_MAXNV], jp1[3 * MCO_MAXNV], jp2[3 * MCO_MAXNV];
  d->nefc = 0;
  /* ---- equalities (always active) */
  for (int e = 0; e < m->neq; e++) {
    if (m->eq_type[e] == MCO_EQ_CONNECT) {
      int b1 = m->eq_obj1[e], b2 = m->eq_obj2[e];
      double p1[3], p2[3], v[3];
      mat_vec3(v, d->xmat[b1], m->eq_data[e]);     for (int k = 0; k < 3; k++) p1[k] = d->xpos[b1][k] + v[k];
      mat_vec3(v, d->xmat[b2], m->eq_data[e] + 3); for (int k = 0; k < 3; k++) p2[k] = d->xpos[b2][k] + v[k];
      mco_jac(m, d, jp1, NULL, p1, b1); mco_jac(m, d, jp2, NULL, p2, b2);
      for (int r = 0; r < 3; r++) {
        for (int k = 0; k < nv; k++) J[k] = jp1[r * nv + k] - jp2[r * nv + k];
        add_row(d, nv, J, p1[r] - p2[r], 0, MCO_EFC_EQUALITY, e);
      }
    } else if (m->eq_type[e] == MCO_EQ_JOINT) {
      int j1 = m->eq_obj1[e], j2 = m->eq_obj2[e];
      const double* pc = m->eq_data[e];
      double q1 = d->qpos[m->jnt_qposadr[j1]] - m->qpos0[m->jnt_qposadr[j1]];
      double q2 = d->qpos[m->jnt_qposadr[j2]] - m->qpos0[m->jnt_qposadr[j2]];
      double poly = pc[0] + q2 * (pc[1] + q2 * (pc[2] + q2 * (pc[3] + q2 * pc[4])));
      double dpoly = pc[1] + q2 * (2 * pc[2] + q2 * (3 * pc[3] + q2 * 4 * pc[4]));
      zero(J, nv);
      J[m->jnt_dofadr[j1]] = 1; J[m->jnt_dofadr[j2]] = -dpoly;
      add_row(d, nv, J, q1 - poly, 0, MCO_EFC_EQUALITY, e);
    }
    else if (m->eq_type[e] == MCO_EQ_WELD) {
      /* mjEQ_WELD [RECALL MuJoCo 2.3.2 mj_instantiateEquality]: data = anchor(3, body2 frame), relpos(3, the same point in
       * body1's frame), relquat(4), torquescale.  Rows 0-2: p(body1) - p(body2); rows 3-5: torquescale * imag(neg(q2) q1 relquat);
       * Jacobian = jac(body1) - jac(body2), its rotational part mapped through 0.5 * neg(q2) (.) q1 relquat. */
      int b1 = m->eq_obj1[e], b2 = m->eq_obj2[e];
      const double* data = m->eq_data[e];
      double p1[3], p2[3], v[3], jr1[3 * MCO_MAXNV], jr2[3 * MCO_MAXNV];
      mat_vec3(v, d->xmat[b1], data + 3); for (int k = 0; k < 3; k++) p1[k] = d->xpos[b1][k] + v[k];
      mat_vec3(v, d->xmat[b2], data);     for (int k = 0; k < 3; k++) p2[k] = d->xpos[b2][k] + v[k];
      mco_jac(m, d, jp1, jr1, p1, b1); mco_jac(m, d, jp2, jr2, p2, b2);
      double quat[4], quat1[4], quat2[4], ts = data[10];
      mco_mulquat(quat, d->xquat[b1], data + 6);       /* q1 * relquat */
      mco_negquat(quat1, d->xquat[b2]);                /* neg(q2) */
      mco_mulquat(quat2, quat1, quat);
      double cpos[6] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2], ts * quat2[1], ts * quat2[2], ts * quat2[3]};
      double Jw[3][MCO_MAXNV];
      for (int k = 0; k < nv; k++) {
        double axis[4] = {0, jr1[0 * nv + k] - jr2[0 * nv + k], jr1[1 * nv + k] - jr2[1 * nv + k], jr1[2 * nv + k] - jr2[2 * nv + k]};
        double t[4], q3[4];
        mco_mulquat(t, quat1, axis);                   /* mju_mulQuatAxis */
        mco_mulquat(q3, t, quat);
        for (int r = 0; r < 3; r++) Jw[r][k] = 0.5 * q3[1 + r] * ts;
      }
      for (int r = 0; r < 3; r++) {
        for (int k = 0; k < nv; k++) J[k] = jp1[r * nv + k] - jp2[r * nv + k];
        add_row(d, nv, J, cpos[r], 0, MCO_EFC_EQUALITY, e);
      }
      for (int r = 0; r < 3; r++) add_row(d, nv, Jw[r], cpos[3 + r], 0, MCO_EFC_EQUALITY, e);
    }
  }
  d->ne = d->nefc;
  /* ---- joint limits: a row only while dist < margin (= 0) */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || m->jnt_type[j] != MCO_JNT_HINGE) continue;
    double q = d->qpos[m->jnt_qposadr[j]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[j][(side + 1) / 2] - q);
      if (dist < 0) {
        zero(J, nv); J[m->jnt_dofadr[j]] = -side;
        add_row(d, nv, J, dist, 0, MCO_EFC_LIMIT, j);
      }
    }
  }
  d->nl = d->nefc - d->ne;
  /* ---- contacts, pyramidal friction cones */
  for (int c = 0; c < d->ncon; c++) {
    mco_contact* con = &d->contact[c];
    int b1 = m->geom_body[con->geom1], b2 = m->geom_body[con->geom2];
    double jr1[3 * MCO_MAXNV], jr2[3 * MCO_MAXNV], Jc[6][MCO_MAXNV];
    mco_jac(m, d, jp1, jr1, con->pos, b1); mco_jac(m, d, jp2, jr2, con->pos, b2);
    /* rows of the contact-frame Jacobian: normal, 2 tangents, then rotations about the same axes */
    for (int a = 0; a < 3; a++) for (int k = 0; k < nv; k++) {
      double sp = 0, sr = 0;
      for (int r = 0; r < 3; r++) {
        sp += con->frame[3 * a + r] * (jp2[r * nv + k] - jp1[r * nv + k]);
        sr += con->frame[3 * a + r] * (jr2[r * nv + k] - jr1[r * nv + k]);
      }
      Jc[a][k] = sp; Jc[3 + a][k] = sr;
    }
    con->efc_address = d->nefc;
    if (con->dim == 1) {
      add_row(d, nv, Jc[0], con->dist, con->includemargin, MCO_EFC_CONTACT, c);
    } else {
      for (int k = 1; k < con->dim; k++) {
        for (int s = 0; s < 2; s++) {
          double sign = s ? -1.0 : 1.0;
          for (int q = 0; q < nv; q++) J[q] = Jc[0][q] + sign * con->friction[k - 1] * Jc[k][q];
          add_row(d, nv, J, con->dist, con->includemargin, MCO_EFC_CONTACT, c);
        }
      }
    }
  }
  /* ---- diagApprox from the compile-time inverse weights */
  for (int i = 0; i < d->nefc; i++) {
    int id = d->efc_id[i];
    if (d->efc_type[i] == MCO_EFC_EQUALITY) {
      if (m->eq_type[id] == MCO_EQ_CONNECT)
        d->efc_diagApprox[i] = m->body_invweight0[m->eq_obj1[id]][0] + m->body_invweight0[m->eq_obj2[id]][0];
      else if (m->eq_type[id] == MCO_EQ_WELD) {
        /* All six rows of a weld carry the same weight.  Pinned by the reference's own data: the mocap keyframe
         * (mycobot280_mocap.xml:6-9) sits at a wrist singularity where the weld cannot be met, and its residual is parallel to
         * the null direction of J^T -- the signature of an isotropic row stiffness; with the rotational inverse weight on rows
         * 3-5 (43x softer) the equilibrium residual is rotation-dominated instead and the arm leaves the keyframe
         * (tests/test_oracle_known_answers.py).  The common value = the translational weight is an assumption [unpinned]. */
        d->efc_diagApprox[i] = m->body_invweight0[m->eq_obj1[id]][0] + m->body_invweight0[m->eq_obj2[id]][0];
        if (m->rule[0] == 1) {            /* study switch: the rule as recalled from mj_diagApprox (rows 3-5 rotational) */
          int first = i; while (first > 0 && d->efc_type[first - 1] == MCO_EFC_EQUALITY && d->efc_id[first - 1] == id) first--;
          int part = (i - first) > 2;
          d->efc_diagApprox[i] = m->body_invweight0[m->eq_obj1[id]][part] + m->body_invweight0[m->eq_obj2[id]][part];
        }
      } else
        d->efc_diagApprox[i] = m->dof_invweight0[m->jnt_dofadr[m->eq_obj1[id]]] + m->dof_invweight0[m->jnt_dofadr[m->eq_obj2[id]]];
    } else if (d->efc_type[i] == MCO_EFC_LIMIT) {
      d->efc_diagApprox[i] = m->dof_invweight0[m->jnt_dofadr[id]];
    } else {
      const mco_contact* con = &d->contact[id];
      int b1 = m->geom_body[con->geom1], b2 = m->geom_body[con->geom2];
      double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
      double rot = m->body_invweight0[b1][1] + m->body_invweight0[b2][1];
      if (con->dim == 1) d->efc_diagApprox[i] = tran;
      else {
        int j = i - con->efc_address;
        double fri = con->friction[j / 2];
        d->efc_diagApprox[i] = tran + fri * fri * (j < 4 ? tran : rot);
        if (m->rule[2] == 1) d->efc_diagApprox[i] = tran;
      }
    }
  }
  /* ---- impedance -> R, D, KBIP */
  for (int i = 0; i < d->nefc; i++) {
    double solref[2], solimp[5], pos = d->efc_pos[i];
    get_solparam(m, d, i, solref, solimp);
    /* a connect's three rows share one impedance, evaluated at the norm of its residual */
    if (d->efc_type[i] == MCO_EFC_EQUALITY && m->eq_type[d->efc_id[i]] != MCO_EQ_JOINT) {
      int first = i; while (first > 0 && d->efc_type[first - 1] == MCO_EFC_EQUALITY && d->efc_id[first - 1] == d->efc_id[i]) first--;
      int size = m->eq_type[d->efc_id[i]] == MCO_EQ_WELD ? 6 : 3;     /* a weld's six rows share one impedance likewise */
      double ss = 0; for (int r = 0; r < size; r++) ss += d->efc_pos[first + r] * d->efc_pos[first + r];
      pos = sqrt(ss);
      if (size == 6 && m->rule[1] == 1) pos = d->efc_pos[i];
      if (size == 6 && m->rule[1] == 2) pos = d->efc_margin[i];
    }
    double imp = get_impedance(solimp, pos, d->efc_margin[i]);
    if (imp < MINIMP) imp = MINIMP; if (imp > MAXIMP) imp = MAXIMP;
    double R = (1 - imp) * d->efc_diagApprox[i] / imp;
    d->efc_R[i] = R > MINVAL ? R : MINVAL;
    double K, B, dmax = solimp[1];
    if (solref[0] > 0) {
      double kd = dmax * dmax * solref[0] * solref[0] * solref[1] * solref[1];
      double bd = dmax * solref[0];
      K = 1 / (kd > MINVAL ? kd : MINVAL); B = 2 / (bd > MINVAL ? bd : MINVAL);
    } else {
      K = -solref[0] / (dmax * dmax > MINVAL ? dmax * dmax : MINVAL);
      B = -solref[1] / (dmax > MINVAL ? dmax : MINVAL);
    }
    d->efc_KBIP[i][0] = K; d->efc_KBIP[i][1] = B; d->efc_KBIP[i][2] = imp; d->efc_KBIP[i][3] = 0;
  }
  /* pyramidal friction rows share Rpy = 2 mu^2 R(first row) [RECALL, impratio = 1] */
  for (int c = 0; c < d->ncon; c++) {
    const mco_contact* con = &d->contact[c];
    if (con->dim > 1) {
      int a = con->efc_address;
      double Rpy = 2 * con->friction[0] * con->friction[0] * d->efc_R[a];
      if (m->rule[3] == 1) continue;
      if (m->rule[3] == 2) Rpy *= 2;
      if (Rpy < MINVAL) Rpy = MINVAL;
      for (int j = 0; j < 2 * (con->dim - 1); j++) d->efc_R[a + j] = Rpy;
    }
  }
  for (int i = 0; i < d->nefc; i++) d->efc_D[i] = 1 / d->efc_R[i];
}

/* ============================================================================ P6: mj_fwdVelocity */
static void fwd_velocity(const mco_model* m, mco_data* d) {
  int nv = m->nv;
  for (int t = 0; t < m->ntendon; t++) d->ten_velocity[t] = dotn(d->ten_J[t], d->qvel, nv);
  for (int u = 0; u < m->nu; u++) d->act_velocity[u] = dotn(d->act_moment[u], d->qvel, nv);
  /* mj_comVel */
  zero(d->cvel[0], 6);
  for (int i = 1; i < m->nbody; i++) {
    double cvel[6];
    copy(cvel, d->cvel[m->body_parent[i]], 6);
    int da = m->body_dofadr[i], dn = m->body_dofnum[i];
    if (dn == 6) {                                   /* free joint */
      for (int k = 0; k < 3; k++) { zero(d->cdof_dot[da + k], 6); for (int r = 0; r < 6; r++) cvel[r] += d->cdof[da + k][r] * d->qvel[da + k]; }
      for (int k = 3; k < 6; k++) cross_motion(d->cdof_dot[da + k], cvel, d->cdof[da + k]);
      for (int k = 3; k < 6; k++) for (int r = 0; r < 6; r++) cvel[r] += d->cdof[da + k][r] * d->qvel[da + k];
    } else {
      for (int j = da; j < da + dn; j++) {
        cross_motion(d->cdof_dot[j], cvel, d->cdof[j]);
        for (int r = 0; r < 6; r++) cvel[r] += d->cdof[j][r] * d->qvel[j];
      }
    }
    copy(d->cvel[i], cvel, 6);
  }
  /* mj_passive: joint damping only (the model has no springs, no fluid, no gravcomp) */
  for (int j = 0; j < nv; j++) d->qfrc_passive[j] = -m->dof_damping[j] * d->qvel[j];
  /* mj_referenceConstraint */
  for (int i = 0; i < d->nefc; i++) {
    d->efc_vel[i] = dotn(d->efc_J[i], d->qvel, nv);
    d->efc_aref[i] = -d->efc_KBIP[i][1] * d->efc_vel[i]
                     - d->efc_KBIP[i][0] * d->efc_KBIP[i][2] * (d->efc_pos[i] - d->efc_margin[i]);
  }
  /* mj_rne(flg_acc = 0): Coriolis, centrifugal and gravity */
  double cacc[MCO_MAXBODY][6], cfrc[MCO_MAXBODY][6];
  zero(cacc[0], 6); for (int k = 0; k < 3; k++) cacc[0][3 + k] = -m->gravity[k];
  zero(cfrc[0], 6);
  for (int i = 1; i < m->nbody; i++) {
    double tmp[6], tmp1[6];
    copy(cacc[i], cacc[m->body_parent[i]], 6);
    for (int j = m->body_dofadr[i]; j >= 0 && j < m->body_dofadr[i] + m->body_dofnum[i]; j++)
      for (int r = 0; r < 6; r++) cacc[i][r] += d->cdof_dot[j][r] * d->qvel[j];
    mul_inert_vec(cfrc[i], d->cinert[i], cacc[i]);
    mul_inert_vec(tmp, d->cinert[i], d->cvel[i]);
    cross_force(tmp1, d->cvel[i], tmp);
    for (int r = 0; r < 6; r++) cfrc[i][r] += tmp1[r];
  }
  for (int i = m->nbody - 1; i > 0; i--) {
    int p = m->body_parent[i];
    if (p > 0) for (int r = 0; r < 6; r++) cfrc[p][r] += cfrc[i][r];
  }
  for (int j = 0; j < nv; j++) d->qfrc_bias[j] = dotn(d->cdof[j], cfrc[m->dof_body[j]], 6);
}

/* ========================================================================== P7: mj_fwdActuation */
static void fwd_actuation(const mco_model* m, mco_data* d) {
  int nv = m->nv;
  zero(d->qfrc_actuator, nv);
  for (int u = 0; u < m->nu; u++) {
    double c = d->ctrl[u];                         /* clamped on a local copy; data.ctrl is untouched */
    if (m->act_ctrllimited[u]) { if (c < m->act_ctrlrange[u][0]) c = m->act_ctrlrange[u][0]; if (c > m->act_ctrlrange[u][1]) c = m->act_ctrlrange[u][1]; }
    double f = m->act_gainprm[u][0] * c + m->act_biasprm[u][0]
             + m->act_biasprm[u][1] * d->act_length[u] + m->act_biasprm[u][2] * d->act_velocity[u];
    if (m->act_forcelimited[u]) { if (f < m->act_forcerange[u][0]) f = m->act_forcerange[u][0]; if (f > m->act_forcerange[u][1]) f = m->act_forcerange[u][1]; }
    d->act_force[u] = f;
    for (int k = 0; k < nv; k++) d->qfrc_actuator[k] += d->act_moment[u][k] * f;
  }
}

/* ======================================================================== P8: mj_fwdAcceleration */
static void fwd_acceleration(const mco_model* m, mco_data* d) {
  int nv = m->nv;
  for (int j = 0; j < nv; j++) d->qfrc_smooth[j] = d->qfrc_passive[j] - d->qfrc_bias[j] + d->qfrc_actuator[j];
  copy(d->qacc_smooth, d->qfrc_smooth, nv);
  chol_solve(nv, d->qL, d->qacc_smooth);
}

/* =========================================================== P9: mj_fwdConstraint (Newton, primal) */
static int row_active(const mco_data* d, int i, double r) { return d->efc_type[i] == MCO_EFC_EQUALITY || r < 0; }

static double solver_cost(const mco_model* m, const mco_data* d, const double* a, double* jar_out) {
  int nv = m->nv;
  double dif[MCO_MAXNV], Md[MCO_MAXNV], cost = 0;
  for (int j = 0; j < nv; j++) dif[j] = a[j] - d->qacc_smooth[j];
  for (int j = 0; j < nv; j++) { Md[j] = dotn(d->qM[j], dif, nv); cost += 0.5 * dif[j] * Md[j]; }
  for (int i = 0; i < d->nefc; i++) {
    double r = dotn(d->efc_J[i], a, nv) - d->efc_aref[i];
    if (jar_out) jar_out[i] = r;
    if (row_active(d, i, r)) cost += 0.5 * d->efc_D[i] * r * r;
  }
  return cost;
}

static void fwd_constraint(const mco_model* m, mco_data* d) {
  int nv = m->nv, nefc = d->nefc;
  zero(d->qfrc_constraint, nv);
  d->solver_iter = 0;
  if (nefc == 0) { copy(d->qacc, d->qacc_smooth, nv); return; }
  /* warm start: the cheaper of qacc_warmstart and qacc_smooth */
  double a[MCO_MAXNV], jar[MCO_MAXEFC];
  if (solver_cost(m, d, d->qacc_warmstart, NULL) < solver_cost(m, d, d->qacc_smooth, NULL)) copy(a, d->qacc_warmstart, nv);
  else copy(a, d->qacc_smooth, nv);
  /* Newton iterations with an exact line search on the piecewise-quadratic cost.  MuJoCo stops at
     tolerance 1e-8 (scaled); here the iteration runs until the active set is stationary, i.e. to
     the exact minimiser, which MuJoCo's answer approaches within its tolerance. */
  static const int MAXIT = 100;
  for (int it = 0; it < MAXIT; it++) {
    double grad[MCO_MAXNV], dif[MCO_MAXNV], p[MCO_MAXNV];
    double H[MCO_MAXNV][MCO_MAXNV], L[MCO_MAXNV][MCO_MAXNV];
    for (int i = 0; i < nefc; i++) jar[i] = dotn(d->efc_J[i], a, nv) - d->efc_aref[i];
    for (int j = 0; j < nv; j++) dif[j] = a[j] - d->qacc_smooth[j];
    for (int j = 0; j < nv; j++) { grad[j] = dotn(d->qM[j], dif, nv); for (int k = 0; k < nv; k++) H[j][k] = d->qM[j][k]; }
    for (int i = 0; i < nefc; i++) if (row_active(d, i, jar[i])) {
      double Dr = d->efc_D[i] * jar[i];
      for (int j = 0; j < nv; j++) {
        if (d->efc_J[i][j] == 0) continue;
        grad[j] += d->efc_J[i][j] * Dr;
        for (int k = 0; k < nv; k++) H[j][k] += d->efc_D[i] * d->efc_J[i][j] * d->efc_J[i][k];
      }
    }
    double gnorm = sqrt(dotn(grad, grad, nv));
    if (gnorm == 0) break;
    cholesky(nv, H, L);
    for (int j = 0; j < nv; j++) p[j] = -grad[j];
    chol_solve(nv, L, p);
    d->solver_iter = it + 1;
    /* exact line search: phi'(alpha) is piecewise linear and increasing; walk its breakpoints */
    double Mp[MCO_MAXNV], jp[MCO_MAXEFC];
    for (int j = 0; j < nv; j++) Mp[j] = dotn(d->qM[j], p, nv);
    double d0 = dotn(grad, p, nv);                /* phi'(0), includes the active constraint rows */
    double slope = dotn(p, Mp, nv);
    for (int i = 0; i < nefc; i++) {
      jp[i] = dotn(d->efc_J[i], p, nv);
      if (row_active(d, i, jar[i])) slope += d->efc_D[i] * jp[i] * jp[i];
    }
    /* breakpoints of the inequality rows, in increasing alpha */
    int order[MCO_MAXEFC], nb = 0; double bp[MCO_MAXEFC];
    for (int i = 0; i < nefc; i++) {
      if (d->efc_type[i] == MCO_EFC_EQUALITY || jp[i] == 0) continue;
      double al = -jar[i] / jp[i];
      if (al > 0) { bp[i] = al; order[nb++] = i; }
    }
    for (int x = 1; x < nb; x++) { int v = order[x], y = x - 1; while (y >= 0 && bp[order[y]] > bp[v]) { order[y + 1] = order[y]; y--; } order[y + 1] = v; }
    double alpha = 0, val = d0;                   /* phi'(alpha) along the walk */
    int changed = 0;
    for (int x = 0; x <= nb; x++) {
      double next = (x < nb) ? bp[order[x]] : INFINITY;
      /* root inside this linear piece? */
      if (slope > 0 && val + slope * (next - alpha) >= 0) { alpha = alpha - val / slope; val = 0; break; }
      if (x == nb) { alpha = next; break; }
      val += slope * (next - alpha); alpha = next;
      int i = order[x];
      /* row i toggles at this breakpoint: it was active iff jar<0 at smaller alpha */
      if (jar[i] < 0 || (jar[i] == 0 && jp[i] > 0)) slope -= d->efc_D[i] * jp[i] * jp[i];
      else slope += d->efc_D[i] * jp[i] * jp[i];
      changed = 1;
    }
    if (!isfinite(alpha)) alpha = 1;
    for (int j = 0; j < nv; j++) a[j] += alpha * p[j];
    /* a full Newton step that crossed no breakpoint lands on the exact minimiser of this piece */
    if (!changed && fabs(alpha - 1) < 1e-9) {
      /* one more gradient check is implicit in the next iteration for pieces that did change */
      int same = 1;
      for (int i = 0; i < nefc; i++) {
        double r = jar[i] + alpha * jp[i];
        if (row_active(d, i, jar[i]) != row_active(d, i, r)) { same = 0; break; }
      }
      if (same) break;
    }
  }
  copy(d->qacc, a, nv);
  for (int i = 0; i < nefc; i++) {
    double r = dotn(d->efc_J[i], a, nv) - d->efc_aref[i];
    d->efc_force[i] = row_active(d, i, r) ? -d->efc_D[i] * r : 0;
    for (int j = 0; j < nv; j++) d->qfrc_constraint[j] += d->efc_J[i][j] * d->efc_force[i];
  }
}

/* ====================================================================================== mj_forward */
void mco_forward(const mco_model* m, mco_data* d) {
  kinematics(m, d);
  com_pos(m, d);
  tendon_transmission(m, d);
  crb_factor(m, d);
  d->ncon = 0;
  if (m->enable_contact) mco_collision(m, d);
  make_constraint(m, d);
  fwd_velocity(m, d);
  fwd_actuation(m, d);
  fwd_acceleration(m, d);
  fwd_constraint(m, d);
}

/* ================================================================= P10: mj_Euler + mj_advance */
static void euler(const mco_model* m, mco_data* d) {
  int nv = m->nv;
  double qacc[MCO_MAXNV];
  int damped = 0;
  for (int j = 0; j < nv; j++) if (m->dof_damping[j] > 0) damped = 1;
  if (!damped) copy(qacc, d->qacc, nv);
  else {
    /* implicit in joint damping: (M + h diag(B)) qacc' = qfrc_smooth + qfrc_constraint */
    double H[MCO_MAXNV][MCO_MAXNV], L[MCO_MAXNV][MCO_MAXNV];
    for (int j = 0; j < nv; j++) { for (int k = 0; k < nv; k++) H[j][k] = d->qM[j][k]; H[j][j] += m->timestep * m->dof_damping[j]; }
    cholesky(nv, H, L);
    for (int j = 0; j < nv; j++) qacc[j] = d->qfrc_smooth[j] + d->qfrc_constraint[j];
    chol_solve(nv, L, qacc);
  }
  /* mj_advance */
  for (int j = 0; j < nv; j++) d->qvel[j] += m->timestep * qacc[j];
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == MCO_JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += m->timestep * d->qvel[da + k];
      quat_integrate(d->qpos + qa + 3, d->qvel + da + 3, m->timestep);
    } else d->qpos[qa] += m->timestep * d->qvel[da];
  }
  d->time += m->timestep;
  copy(d->qacc_warmstart, d->qacc, nv);
}

static int bad(double x) { return !(x == x) || x > 1e10 || x < -1e10; }   /* mj_checkPos/Vel/Acc */

void mco_step(const mco_model* m, mco_data* d) {
  int isbad = 0;
  for (int j = 0; j < m->nq; j++) isbad |= bad(d->qpos[j]);
  for (int j = 0; j < m->nv; j++) isbad |= bad(d->qvel[j]);
  if (isbad) { mco_reset_data(m, d); d->warning_badstate++; }   /* mj_checkPos/Vel -> mj_resetData */
  mco_forward(m, d);
  isbad = 0;
  for (int j = 0; j < m->nv; j++) isbad |= bad(d->qacc[j]);
  if (isbad) { mco_reset_data(m, d); d->warning_badstate++; mco_forward(m, d); }   /* mj_checkAcc */
  euler(m, d);
}

void mco_reset_data(const mco_model* m, mco_data* d) {
  int w = d->warning_badstate;
  memset(d, 0, sizeof(*d));
  d->warning_badstate = w;
  copy(d->qpos, m->qpos0, m->nq);
  for (int i = 1; i < m->nbody; i++) if (m->body_mocapid[i] >= 0) {     /* mj_resetData: mocap pose := body_pos / body_quat */
    copy(d->mocap_pos[m->body_mocapid[i]], m->body_pos[i], 3); copy(d->mocap_quat[m->body_mocapid[i]], m->body_quat[i], 4);
  }
}

double mco_energy(const mco_model* m, const mco_data* d, double* potential, double* kinetic) {
  double pe = 0, ke = 0, v[MCO_MAXNV];
  for (int i = 1; i < m->nbody; i++) pe -= m->body_mass[i] * dot3(m->gravity, d->xipos[i]);
  for (int j = 0; j < m->nv; j++) v[j] = dotn(d->qM[j], d->qvel, m->nv);
  ke = 0.5 * dotn(v, d->qvel, m->nv);
  if (potential) *potential = pe;
  if (kinetic) *kinetic = ke;
  return pe + ke;
}

/* =============================================================================== mj_setConst */
void mco_setconst(mco_model* m) {
  static mco_data dd;                              /* setconst is called once, single-threaded */
  mco_data* d = &dd;
  int nv = m->nv;
  memset(d, 0, sizeof(*d));
  copy(d->qpos, m->qpos0, m->nq);
  kinematics(m, d); com_pos(m, d); crb_factor(m, d);
  double tr = 0; for (int j = 0; j < nv; j++) tr += d->qM[j][j];
  m->meaninertia = nv ? tr / nv : 1;
  double jac[6 * MCO_MAXNV], col[MCO_MAXNV];
  for (int b = 0; b < m->nbody; b++) {
    m->body_invweight0[b][0] = m->body_invweight0[b][1] = 0;
    if (b == 0 || m->body_weldid[b] == 0) continue;
    mco_jac(m, d, jac, jac + 3 * nv, d->xipos[b], b);
    for (int part = 0; part < 2; part++) {
      double s = 0;
      for (int r = 0; r < 3; r++) {
        const double* row = jac + (3 * part + r) * nv;
        copy(col, row, nv); chol_solve(nv, d->qL, col);
        s += dotn(row, col, nv);
      }
      m->body_invweight0[b][part] = s / 3;
    }
  }
  for (int j = 0; j < m->njnt; j++) {
    int da = m->jnt_dofadr[j];
    int n = m->jnt_type[j] == MCO_JNT_FREE ? 6 : 1;
    double diag[6];
    for (int k = 0; k < n; k++) { zero(col, nv); col[da + k] = 1; chol_solve(nv, d->qL, col); diag[k] = col[da + k]; }
    if (n == 6) {
      double t = (diag[0] + diag[1] + diag[2]) / 3, r = (diag[3] + diag[4] + diag[5]) / 3;
      for (int k = 0; k < 3; k++) { m->dof_invweight0[da + k] = t; m->dof_invweight0[da + 3 + k] = r; }
    } else m->dof_invweight0[da] = diag[0];
  }
}

/* ===================================================================== ctypes-friendly accessors */
int mco_model_sizeof(void) { return (int)sizeof(mco_model); }
int mco_data_sizeof(void) { return (int)sizeof(mco_data); }

typedef struct { const char* name; size_t off; int count; int is_int; } field_t;
#define MF_I(f) { #f, offsetof(mco_model, f), (int)(sizeof(((mco_model*)0)->f) / sizeof(int)), 1 }
#define MF_D(f) { #f, offsetof(mco_model, f), (int)(sizeof(((mco_model*)0)->f) / sizeof(double)), 0 }
static const field_t model_fields[] = {
  MF_I(nbody), MF_I(njnt), MF_I(nq), MF_I(nv), MF_I(ngeom), MF_I(nsite), MF_I(nu), MF_I(neq), MF_I(ntendon),
  MF_I(nexclude), MF_I(enable_contact), MF_I(collide_scope_geom), MF_I(collide_extra), MF_I(geom_poly), MF_I(maxentry), MF_I(rule), MF_D(timestep), MF_D(gravity), MF_D(meaninertia),
  MF_I(body_parent), MF_I(body_rootid), MF_I(body_weldid), MF_I(body_dofadr), MF_I(body_dofnum), MF_I(body_mocapid),
  MF_D(body_pos), MF_D(body_quat), MF_D(body_ipos), MF_D(body_iquat), MF_D(body_mass), MF_D(body_inertia),
  MF_I(jnt_type), MF_I(jnt_body), MF_I(jnt_qposadr), MF_I(jnt_dofadr), MF_I(jnt_limited),
  MF_D(jnt_pos), MF_D(jnt_axis), MF_D(jnt_range), MF_D(jnt_solref), MF_D(jnt_solimp),
  MF_I(dof_body), MF_I(dof_jnt), MF_I(dof_parent), MF_D(dof_armature), MF_D(dof_damping), MF_D(qpos0),
  MF_I(geom_type), MF_I(geom_body), MF_I(geom_condim), MF_I(geom_contype), MF_I(geom_conaffinity),
  MF_D(geom_pos), MF_D(geom_quat), MF_D(geom_size), MF_D(geom_friction), MF_D(geom_solref), MF_D(geom_solimp),
  MF_I(site_body), MF_D(site_pos), MF_D(site_quat),
  MF_I(act_trntype), MF_I(act_trnid), MF_I(act_ctrllimited), MF_I(act_forcelimited),
  MF_D(act_gear), MF_D(act_gainprm), MF_D(act_biasprm), MF_D(act_ctrlrange), MF_D(act_forcerange),
  MF_I(ten_num), MF_I(ten_jnt), MF_D(ten_coef),
  MF_I(eq_type), MF_I(eq_obj1), MF_I(eq_obj2), MF_D(eq_data), MF_D(eq_solref), MF_D(eq_solimp),
  MF_I(exclude), MF_D(body_invweight0), MF_D(dof_invweight0),
};
#define DF_I(f) { #f, offsetof(mco_data, f), (int)(sizeof(((mco_data*)0)->f) / sizeof(int)), 1 }
#define DF_D(f) { #f, offsetof(mco_data, f), (int)(sizeof(((mco_data*)0)->f) / sizeof(double)), 0 }
static const field_t data_fields[] = {
  DF_D(qpos), DF_D(qvel), DF_D(ctrl), DF_D(qacc_warmstart), DF_D(time),
  DF_D(xpos), DF_D(xquat), DF_D(xmat), DF_D(xipos), DF_D(ximat), DF_D(xanchor), DF_D(xaxis),
  DF_D(geom_xpos), DF_D(geom_xmat), DF_D(site_xpos), DF_D(site_xmat), DF_D(subtree_com), DF_D(cdof),
  DF_D(qM), DF_D(qfrc_passive), DF_D(qfrc_bias), DF_D(act_force), DF_D(qfrc_actuator), DF_D(qfrc_smooth),
  DF_D(qacc_smooth), DF_I(ncon), DF_I(nefc), DF_I(ne), DF_I(nl), DF_I(efc_type), DF_I(efc_id), DF_D(efc_J),
  DF_D(efc_pos), DF_D(efc_diagApprox), DF_D(efc_R), DF_D(efc_D), DF_D(efc_KBIP), DF_D(efc_vel), DF_D(efc_aref),
  DF_D(efc_force), DF_D(qfrc_constraint), DF_D(qacc), DF_I(solver_iter), DF_I(warning_badstate), DF_I(nentry), DF_I(ndrop),
  DF_D(mocap_pos), DF_D(mocap_quat), DF_D(contact),      /* contact: raw, 28 doubles per mco_contact (the last two hold four ints) */
};

static const field_t* find(const field_t* tab, int n, const char* name) {
  for (int i = 0; i < n; i++) if (!strcmp(tab[i].name, name)) return &tab[i];
  return NULL;
}
#define NMODEL ((int)(sizeof(model_fields) / sizeof(field_t)))
#define NDATA ((int)(sizeof(data_fields) / sizeof(field_t)))

int mco_model_set_i(mco_model* m, const char* field, const int* v, int n) {
  const field_t* f = find(model_fields, NMODEL, field);
  if (!f || !f->is_int || n > f->count) return -1;
  memcpy((char*)m + f->off, v, sizeof(int) * n); return 0;
}
int mco_model_set_d(mco_model* m, const char* field, const double* v, int n) {
  const field_t* f = find(model_fields, NMODEL, field);
  if (!f || f->is_int || n > f->count) return -1;
  memcpy((char*)m + f->off, v, sizeof(double) * n); return 0;
}
void mco_model_set_poly(mco_model* m, const double* blob) { m->poly = blob; }
int mco_model_get_d(const mco_model* m, const char* field, double* v, int n) {
  const field_t* f = find(model_fields, NMODEL, field);
  if (!f || f->is_int || n > f->count) return -1;
  memcpy(v, (const char*)m + f->off, sizeof(double) * n); return 0;
}
int mco_data_get_d(const mco_data* d, const char* field, double* v, int n) {
  const field_t* f = find(data_fields, NDATA, field);
  if (!f || f->is_int || n > f->count) return -1;
  memcpy(v, (const char*)d + f->off, sizeof(double) * n); return 0;
}
int mco_data_set_d(mco_data* d, const char* field, const double* v, int n) {
  const field_t* f = find(data_fields, NDATA, field);
  if (!f || f->is_int || n > f->count) return -1;
  memcpy((char*)d + f->off, v, sizeof(double) * n); return 0;
}
int mco_data_get_i(const mco_data* d, const char* field, int* v, int n) {
  const field_t* f = find(data_fields, NDATA, field);
  if (!f || !f->is_int || n > f->count) return -1;
  memcpy(v, (const char*)d + f->off, sizeof(int) * n); return 0;
}
