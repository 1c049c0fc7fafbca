/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see mco_env.h for the reference lines restated here).
 */
#include "mco_env.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAX_CARTESIAN_DISPLACEMENT 0.2f   /* mycobot.py:22, multiplied in float32 (numpy f32 array * python float) */
#define MAX_ROTATION_DISPLACEMENT 0.5f    /* mycobot.py:23 */
#define IK_REGULARIZATION 0.3             /* utils.py:470 */
#define IK_ROT_DT 50.0                    /* utils.py:528 */
#define MAX_RESET_ATTEMPTS 1000           /* the reference's rejection loops are unbounded (mycobot.py:218,232) */

typedef struct {
  mco_data d;
  double goal[3], qpos_lag[MCO_MAXNQ], ep_return;
  int32_t elapsed, episode, ep_length;
  uint32_t draw;
} env_t;

struct mco_envs {
  mco_model model;          /* nominal model */
  mco_model* env_model;     /* per-env copies when domain randomisation is on, else NULL */
  mco_env_config cfg;
  env_t* env;
  int obs_dim, act_dim;
  double initial_gripper_xpos[3], grip_center, grip_range, dt;
  uint64_t seed;
};

/* ------------------------------------------------------------------------- Philox4x32-10 */
void mco_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* two uniforms in [0,1) with 53 random bits each; stream 0 = reset sampling */
static void rng_pair(const mco_envs* e, int i, env_t* v, uint32_t stream, double* u0, double* u1) {
  uint64_t gid = (uint64_t)(e->cfg.env_id_offset + i);
  uint32_t ctr[4] = { (uint32_t)gid, (uint32_t)v->episode, v->draw++, stream ^ ((uint32_t)(gid >> 32) << 8) };
  uint32_t key[2] = { (uint32_t)e->seed, (uint32_t)(e->seed >> 32) }, r[4];
  mco_philox4x32_10(ctr, key, r);
  *u0 = (double)((((uint64_t)r[0] << 32) | r[1]) >> 11) * (1.0 / 9007199254740992.0);
  *u1 = (double)((((uint64_t)r[2] << 32) | r[3]) >> 11) * (1.0 / 9007199254740992.0);
}

static const mco_model* model_of(const mco_envs* e, int i) { return e->env_model ? &e->env_model[i] : &e->model; }

/* ----------------------------------------------------------- rotations.* helpers (Appendix C.2) */
static void euler2quat(double* q, const double* e) {
  double ai = e[2] / 2, aj = -e[1] / 2, ak = e[0] / 2;
  double si = sin(ai), sj = sin(aj), sk = sin(ak), ci = cos(ai), cj = cos(aj), ck = cos(ak);
  double cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
  q[0] = cj * cc + sj * ss; q[3] = cj * sc - sj * cs; q[2] = -(cj * ss + sj * cc); q[1] = cj * cs - sj * sc;
}
static void mat2euler(double* e, const double* m) {
  double cy = sqrt(m[8] * m[8] + m[5] * m[5]);
  if (cy > 4 * 2.220446049250313e-16) {
    e[2] = -atan2(m[1], m[0]); e[1] = -atan2(-m[2], cy); e[0] = -atan2(m[5], m[8]);
  } else {
    e[2] = -atan2(-m[3], m[4]); e[1] = -atan2(-m[2], cy); e[0] = 0.0;
  }
}

/* ------------------------------------------------------------------------------- sampling */
static void sample_goal(const mco_envs* e, int i, env_t* v, double g[3]) {     /* mycobot.py:238-243 */
  double ux, uy, uc, uz;
  rng_pair(e, i, v, 0, &ux, &uy);
  rng_pair(e, i, v, 0, &uc, &uz);
  /* random.uniform(a, b) = a + (b - a) * random(); written as one fused multiply-add so that the CPU oracle and
     the GPU kernel round identically (the reference's own draws are not reproducible anyway, Appendix D-6) */
  g[0] = fma(0.12 - -0.12, ux, -0.12);
  g[1] = fma(0.06 - -0.06, uy, -0.06);
  g[2] = e->cfg.height_offset;
  if (e->cfg.target_in_the_air && uc < 0.5) g[2] = fma(0.1 - 0.0, uz, e->cfg.height_offset);
}

static void domain_randomise(mco_envs* e, int i, env_t* v);

/* --------------------------------------------------------------------------------- _get_obs */
static void get_obs(const mco_envs* e, int i, double* obs, double* achieved, double* desired) {
  const env_t* v = &e->env[i];
  const mco_model* m = model_of(e, i);
  const mco_data* d = &v->d;
  int nv = m->nv, k = 0;
  double jacp[3 * MCO_MAXNV], jacr[3 * MCO_MAXNV], grip_velp[3];
  const double* grip_pos = d->site_xpos[e->cfg.eef_site];
  mco_jac_site(m, d, jacp, NULL, e->cfg.eef_site);
  for (int r = 0; r < 3; r++) { double s = 0; for (int j = 0; j < nv; j++) s += jacp[r * nv + j] * d->qvel[j]; grip_velp[r] = s * e->dt; }
  int qa0 = m->jnt_qposadr[e->cfg.grip_jnt[0]], qa1 = m->jnt_qposadr[e->cfg.grip_jnt[1]];
  int da0 = m->jnt_dofadr[e->cfg.grip_jnt[0]], da1 = m->jnt_dofadr[e->cfg.grip_jnt[1]];
  for (int r = 0; r < 3; r++) obs[k++] = grip_pos[r];
  if (e->cfg.has_object) {
    int s = e->cfg.obj_site;
    const double* object_pos = d->site_xpos[s];
    double rot[3], velp[3], velr[3];
    mat2euler(rot, d->site_xmat[s]);
    mco_jac_site(m, d, jacp, jacr, s);
    for (int r = 0; r < 3; r++) {
      double sp = 0, sr = 0;
      for (int j = 0; j < nv; j++) { sp += jacp[r * nv + j] * d->qvel[j]; sr += jacr[r * nv + j] * d->qvel[j]; }
      velp[r] = sp * e->dt - grip_velp[r]; velr[r] = sr * e->dt;
    }
    for (int r = 0; r < 3; r++) obs[k++] = object_pos[r];
    for (int r = 0; r < 3; r++) obs[k++] = object_pos[r] - grip_pos[r];
    obs[k++] = d->qpos[qa0]; obs[k++] = d->qpos[qa1];
    for (int r = 0; r < 3; r++) obs[k++] = rot[r];
    for (int r = 0; r < 3; r++) obs[k++] = velp[r];
    for (int r = 0; r < 3; r++) obs[k++] = velr[r];
    for (int r = 0; r < 3; r++) achieved[r] = object_pos[r];
  } else {
    obs[k++] = d->qpos[qa0]; obs[k++] = d->qpos[qa1];
    for (int r = 0; r < 3; r++) achieved[r] = grip_pos[r];
  }
  for (int r = 0; r < 3; r++) obs[k++] = grip_velp[r];
  obs[k++] = d->qvel[da0] * e->dt; obs[k++] = d->qvel[da1] * e->dt;
  for (int r = 0; r < 3; r++) desired[r] = v->goal[r];
}

static double goal_distance(const double* a, const double* b) {
  double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return sqrt(dx * dx + dy * dy + dz * dz);
}

void mco_compute_reward(const double* achieved, const double* desired, int n, int reward_type,
                        double threshold, double* out) {                          /* mycobot.py:289-298 */
  for (int i = 0; i < n; i++) {
    double dist = goal_distance(achieved + 3 * i, desired + 3 * i);
    out[i] = reward_type == MCO_REWARD_SPARSE ? -(double)(float)(dist > threshold) : -dist;
  }
}

/* ------------------------------------------------ stage_rewards + check_contact (mycobot.py:402-448, utils.py:598-604) */
static int check_contact(const mco_data* d, int g1, int g2) {
  for (int c = 0; c < d->ncon; c++)
    if ((d->contact[c].geom1 == g1 && d->contact[c].geom2 == g2) || (d->contact[c].geom1 == g2 && d->contact[c].geom2 == g1)) return 1;
  return 0;
}
static double shaped_reward(const mco_envs* e, int i) {
  /* reach 0.2 (1 - tanh d_grip,obj); grasp 0.5 iff both pads touch the cube; lift 0.5 + 0.4 (1 - tanh d_obj,target).
     The `target0` site is only moved in _render_callback (mycobot.py:309-311), so without rendering it stays at its
     MJCF position (-0.15, 0, 0.21); that is what is restated here. */
  const mco_data* d = &e->env[i].d;
  const double target0[3] = { -0.15, 0.0, 0.21 };
  const double* grip = d->site_xpos[e->cfg.eef_site]; const double* obj = d->site_xpos[e->cfg.obj_site];
  double dx = grip[0] - obj[0], dy = grip[1] - obj[1], dz = grip[2] - obj[2];
  double r_reach = (1 - tanh(sqrt(dx * dx + dy * dy + dz * dz))) * 0.2;
  double r_grasp = (check_contact(d, e->cfg.pad_geom[0], e->cfg.obj_geom) && check_contact(d, e->cfg.pad_geom[1], e->cfg.obj_geom)) ? 0.5 : 0.0;
  double r_lift = 0.0;
  if (r_grasp > 0.0) {
    double tx = obj[0] - target0[0], ty = obj[1] - target0[1], tz = obj[2] - target0[2];
    r_lift = 0.5 + (1 - tanh(sqrt(tx * tx + ty * ty + tz * tz))) * (0.9 - 0.5);
  }
  double mx = r_reach > r_grasp ? r_reach : r_grasp;
  return (mx > r_lift ? mx : r_lift) * 100;
}

/* ------------------------------------------------------------------------------ reset_model */
static void reset_one(mco_envs* e, int i) {
  env_t* v = &e->env[i];
  const mco_model* m = model_of(e, i);
  mco_data* d = &v->d;
  v->draw = 0;
  if (e->cfg.dr_enable) { domain_randomise(e, i, v); m = model_of(e, i); }
  memcpy(d->qpos, e->cfg.init_qpos, sizeof(double) * m->nq);
  memcpy(d->qvel, e->cfg.init_qvel, sizeof(double) * m->nv);
  memcpy(d->ctrl, e->cfg.init_ctrl, sizeof(double) * m->nu);
  mco_forward(m, d);          /* note: time and qacc_warmstart persist (Appendix D-5) */
  double oxy[2] = { e->initial_gripper_xpos[0], e->initial_gripper_xpos[1] }, g[3];
  if (e->cfg.has_object) {
    int tries = 0;
    while (hypot(oxy[0] - e->initial_gripper_xpos[0], oxy[1] - e->initial_gripper_xpos[1]) < 0.1 && tries++ < MAX_RESET_ATTEMPTS) {
      sample_goal(e, i, v, g); oxy[0] = g[0]; oxy[1] = g[1];
    }
    int qa = m->jnt_qposadr[e->cfg.obj_jnt];
    d->qpos[qa] = oxy[0]; d->qpos[qa + 1] = oxy[1];
  }
  mco_forward(m, d);
  int tries = 0;
  sample_goal(e, i, v, v->goal);
  while (hypot(v->goal[0] - oxy[0], v->goal[1] - oxy[1]) < 0.1 && tries++ < MAX_RESET_ATTEMPTS) sample_goal(e, i, v, v->goal);
  memcpy(v->qpos_lag, d->qpos, sizeof(double) * m->nq);
  v->elapsed = 0; v->ep_return = 0; v->ep_length = 0;
  v->episode++;
}

static void domain_randomise(mco_envs* e, int i, env_t* v) {
  /* build-defined extension: per-reset scale of the cube's mass/inertia and of the
     sliding friction of the cube and pad geoms; invweight0 stays nominal (no mj_setConst). */
  mco_model* m = &e->env_model[i];
  double um, uf;
  uint32_t keep = v->draw; v->draw = 0;       /* stream 1 has its own draw counter: DR does not shift the goal draws */
  rng_pair(e, i, v, 1, &um, &uf);
  v->draw = keep;
  double ms = e->cfg.dr_mass_range[0] + (e->cfg.dr_mass_range[1] - e->cfg.dr_mass_range[0]) * um;
  double fs = e->cfg.dr_friction_range[0] + (e->cfg.dr_friction_range[1] - e->cfg.dr_friction_range[0]) * uf;
  if (e->cfg.has_object) {
    int b = e->model.jnt_body[e->cfg.obj_jnt];
    m->body_mass[b] = e->model.body_mass[b] * ms;
    for (int k = 0; k < 3; k++) m->body_inertia[b][k] = e->model.body_inertia[b][k] * ms;
    int gs[3] = { e->cfg.obj_geom, e->cfg.pad_geom[0], e->cfg.pad_geom[1] };
    for (int k = 0; k < 3; k++) if (gs[k] >= 0) m->geom_friction[gs[k]][0] = e->model.geom_friction[gs[k]][0] * fs;
  }
}

/* --------------------------------------------------------------- IKController.compute_qpos_delta */
static void ik_delta(const mco_envs* e, int i, const double* target_pos, const double* target_quat, double* dq) {
  const mco_model* m = model_of(e, i);
  const mco_data* d = &e->env[i].d;
  int nv = m->nv, s = e->cfg.eef_site;
  double J[6 * MCO_MAXNV], err[6], q[4], nq[4], eq[4];
  for (int r = 0; r < 3; r++) err[r] = target_pos[r] - d->site_xpos[s][r];
  mco_mat2quat(q, d->site_xmat[s]); mco_negquat(nq, q); mco_mulquat(eq, target_quat, nq);
  mco_quat2vel(err + 3, eq, IK_ROT_DT);
  mco_jac_site(m, d, J, J + 3 * nv, s);
  /* solve_DLS: (J^T J + tau I) x = J^T e; lstsq on this full-rank SPD system is its exact solve */
  double H[MCO_MAXNV][MCO_MAXNV], L[MCO_MAXNV][MCO_MAXNV];
  for (int a = 0; a < nv; a++) {
    for (int b = 0; b < nv; b++) { double t = 0; for (int r = 0; r < 6; r++) t += J[r * nv + a] * J[r * nv + b]; H[a][b] = t; }
    H[a][a] += IK_REGULARIZATION;
    double t = 0; for (int r = 0; r < 6; r++) t += J[r * nv + a] * err[r]; dq[a] = t;
  }
  for (int a = 0; a < nv; a++) for (int b = 0; b <= a; b++) {       /* Cholesky, lower */
    double t = H[a][b]; for (int k = 0; k < b; k++) t -= L[a][k] * L[b][k];
    L[a][b] = (a == b) ? sqrt(t) : t / L[b][b];
  }
  for (int a = 0; a < nv; a++) { double t = dq[a]; for (int k = 0; k < a; k++) t -= L[a][k] * dq[k]; dq[a] = t / L[a][a]; }
  for (int a = nv - 1; a >= 0; a--) { double t = dq[a]; for (int k = a + 1; k < nv; k++) t -= L[k][a] * dq[k]; dq[a] = t / L[a][a]; }
}

static void substeps(mco_envs* e, int i, int n) {
  env_t* v = &e->env[i];
  const mco_model* m = model_of(e, i);
  for (int s = 0; s < n; s++) {
    memcpy(v->qpos_lag, v->d.qpos, sizeof(double) * m->nq);
    mco_step(m, &v->d);
  }
}

/* ------------------------------------------------------------------------------------- step */
static void step_one(mco_envs* e, int i, const float* action) {                /* mycobot.py:132-193 */
  env_t* v = &e->env[i];
  mco_data* d = &v->d;
  float a[8];
  for (int k = 0; k < e->act_dim; k++) { float x = action[k]; a[k] = x < -1.f ? -1.f : (x > 1.f ? 1.f : x); }
  if (e->cfg.controller == MCO_CTRL_IK) {
    int s = e->cfg.eef_site;
    double target_pos[3], target_quat[4];
    for (int r = 0; r < 3; r++) target_pos[r] = d->site_xpos[s][r] + (double)(a[r] * MAX_CARTESIAN_DISPLACEMENT);
    if (e->cfg.fetch_env) { target_quat[0] = 0; target_quat[1] = -0.707; target_quat[2] = 0; target_quat[3] = 0.707; }
    else {
      double eul[3], qrot[4], cur[4];
      for (int r = 0; r < 3; r++) eul[r] = (double)(a[3 + r] * MAX_ROTATION_DISPLACEMENT);
      euler2quat(qrot, eul); mco_mat2quat(cur, d->site_xmat[s]); mco_mulquat(target_quat, qrot, cur);
    }
    double grip = e->grip_center + (double)a[e->act_dim - 1] * e->grip_range;
    for (int c = 0; c < e->cfg.control_steps; c++) {
      double dq[MCO_MAXNV];
      ik_delta(e, i, target_pos, target_quat, dq);
      for (int k = 0; k < 6; k++) d->ctrl[k] = d->ctrl[k] + dq[k];   /* accumulates, unclamped (D-3) */
      d->ctrl[6] = grip;
      substeps(e, i, e->cfg.frame_skip);
    }
  } else if (e->cfg.controller == MCO_CTRL_MOCAP) {                              /* mycobot.py:172-189 */
    /* mocap_action = (0.1 a[:3], quat - xquat[tcp]); mocap_set_action: reset_mocap2body_xpos (mocap pose := pose of the
       welded body, as of the last forward pass) then mocap_pos += dpos, mocap_quat += dquat  [RECALL gymnasium_robotics] */
    int tb = e->cfg.tcp_body;
    double quat[4] = {0.5, -0.5, -0.5, 0.5};                                      /* fetch: fixed orientation (:178-179) */
    if (!e->cfg.fetch_env) for (int r = 0; r < 4; r++) quat[r] = (double)a[3 + r];
    for (int r = 0; r < 3; r++) d->mocap_pos[0][r] = d->xpos[tb][r] + (double)(a[r] * 0.1f);      /* f32 product, as numpy */
    for (int r = 0; r < 4; r++) { double dq = quat[r] - d->xquat[tb][r]; d->mocap_quat[0][r] = d->xquat[tb][r] + dq; }
    d->ctrl[model_of(e, i)->nu - 1] = e->grip_center + (double)a[e->act_dim - 1] * e->grip_range;
    substeps(e, i, e->cfg.frame_skip);
  } else {
    /* joint: `data.ctrl += 0.05 a` is overwritten by do_simulation's `data.ctrl[:] = action` (D-2) */
    for (int k = 0; k < 7; k++) d->ctrl[k] = (double)a[k];
    substeps(e, i, e->cfg.frame_skip);
  }
}

void mco_envs_step(mco_envs* e, const float* actions, double* obs, double* achieved, double* desired,
                   double* reward, uint8_t* terminated, uint8_t* truncated, uint8_t* is_success,
                   double* final_obs, double* final_achieved, double* final_desired,
                   double* ep_return, int32_t* ep_length) {
  int n = e->cfg.n_envs, D = e->obs_dim, A = e->act_dim;
#pragma omp parallel for schedule(dynamic, 8) num_threads(e->cfg.n_threads > 0 ? e->cfg.n_threads : 1)
  for (int i = 0; i < n; i++) {
    env_t* v = &e->env[i];
    step_one(e, i, actions + (size_t)i * A);
    if (e->cfg.block_gripper) {                 /* _step_callback: set_joint_qpos(finger joints, 0) + mj_forward */
      const mco_model* mm = model_of(e, i);
      v->d.qpos[mm->jnt_qposadr[e->cfg.finger_jnt[0]]] = 0.0;
      v->d.qpos[mm->jnt_qposadr[e->cfg.finger_jnt[1]]] = 0.0;
      mco_forward(mm, &v->d);
      memcpy(v->qpos_lag, v->d.qpos, sizeof(double) * mm->nq);
    }
    double o[32], ag[3], dg[3], r;
    get_obs(e, i, o, ag, dg);
    double dist = goal_distance(ag, dg);
    int succ = dist < e->cfg.distance_threshold;
    if (e->cfg.reward_type == MCO_REWARD_SHAPING) r = shaped_reward(e, i);
    else mco_compute_reward(ag, dg, 1, e->cfg.reward_type, e->cfg.distance_threshold, &r);
    v->elapsed++; v->ep_length++; v->ep_return += r;
    int term = succ, trunc = succ || (v->elapsed >= e->cfg.max_episode_steps);   /* D-4 + TimeLimit */
    reward[i] = r; terminated[i] = (uint8_t)term; truncated[i] = (uint8_t)trunc; is_success[i] = (uint8_t)succ;
    if (ep_return) ep_return[i] = v->ep_return;
    if (ep_length) ep_length[i] = v->ep_length;
    if ((term || trunc) && e->cfg.auto_reset) {
      if (final_obs) memcpy(final_obs + (size_t)i * D, o, sizeof(double) * D);
      if (final_achieved) memcpy(final_achieved + 3 * (size_t)i, ag, sizeof(double) * 3);
      if (final_desired) memcpy(final_desired + 3 * (size_t)i, dg, sizeof(double) * 3);
      reset_one(e, i);
      get_obs(e, i, o, ag, dg);
    }
    memcpy(obs + (size_t)i * D, o, sizeof(double) * D);
    memcpy(achieved + 3 * (size_t)i, ag, sizeof(double) * 3);
    memcpy(desired + 3 * (size_t)i, dg, sizeof(double) * 3);
  }
}

void mco_envs_reset(mco_envs* e, const uint8_t* mask, int reseed, uint64_t seed,
                    double* obs, double* achieved, double* desired) {
  int n = e->cfg.n_envs, D = e->obs_dim;
  if (reseed) e->seed = seed;
#pragma omp parallel for schedule(dynamic, 8) num_threads(e->cfg.n_threads > 0 ? e->cfg.n_threads : 1)
  for (int i = 0; i < n; i++) {
    if (mask && !mask[i]) continue;
    if (reseed) e->env[i].episode = 0;
    reset_one(e, i);
  }
  if (obs) for (int i = 0; i < n; i++) get_obs(e, i, obs + (size_t)i * D, achieved + 3 * (size_t)i, desired + 3 * (size_t)i);
}

/* --------------------------------------------------------------------------------- lifecycle */
int mco_env_config_sizeof(void) { return (int)sizeof(mco_env_config); }
int mco_envs_obs_dim(const mco_envs* e) { return e->obs_dim; }
int mco_envs_action_dim(const mco_envs* e) { return e->act_dim; }
void mco_envs_initial_gripper_xpos(const mco_envs* e, double out[3]) { memcpy(out, e->initial_gripper_xpos, 3 * sizeof(double)); }
mco_data* mco_envs_data(mco_envs* e, int i) { return &e->env[i].d; }

mco_envs* mco_envs_create(const mco_model* model, const mco_env_config* cfg) {
  mco_envs* e = (mco_envs*)calloc(1, sizeof(mco_envs));
  e->model = *model; e->cfg = *cfg; e->seed = cfg->seed;
  e->obs_dim = cfg->has_object ? 25 : 10;
  e->act_dim = (cfg->controller == MCO_CTRL_MOCAP) ? (cfg->fetch_env ? 4 : 8)
             : (cfg->controller == MCO_CTRL_IK && cfg->fetch_env) ? 4 : 7;       /* mycobot.py:84-103 */
  e->dt = cfg->frame_skip * model->timestep;                                      /* mycobot.py:346 */
  int last = model->nu - 1;                                                       /* mycobot.py:113-115 */
  e->grip_range = (model->act_ctrlrange[last][1] - model->act_ctrlrange[last][0]) / 2.0;
  e->grip_center = (model->act_ctrlrange[last][1] + model->act_ctrlrange[last][0]) / 2.0;
  e->env = (env_t*)calloc((size_t)cfg->n_envs, sizeof(env_t));
  if (cfg->dr_enable) {
    e->env_model = (mco_model*)malloc(sizeof(mco_model) * (size_t)cfg->n_envs);
    for (int i = 0; i < cfg->n_envs; i++) e->env_model[i] = *model;
  }
  /* _env_setup (mycobot.py:450-472): forward at the initial state, cache the gripper position */
  mco_data* d = &e->env[0].d;
  memcpy(d->qpos, cfg->init_qpos, sizeof(double) * model->nq);
  memcpy(d->mocap_pos[0], cfg->init_mocap, sizeof(double) * 3); memcpy(d->mocap_quat[0], cfg->init_mocap + 3, sizeof(double) * 4);
  mco_forward(model, d);
  memcpy(e->initial_gripper_xpos, d->site_xpos[cfg->eef_site], 3 * sizeof(double));
  for (int i = 0; i < cfg->n_envs; i++) {
    memset(&e->env[i], 0, sizeof(env_t));
    memcpy(e->env[i].d.qpos, cfg->init_qpos, sizeof(double) * model->nq);
    memcpy(e->env[i].qpos_lag, cfg->init_qpos, sizeof(double) * model->nq);
    memcpy(e->env[i].d.mocap_pos[0], cfg->init_mocap, sizeof(double) * 3);
    memcpy(e->env[i].d.mocap_quat[0], cfg->init_mocap + 3, sizeof(double) * 4);
  }
  return e;
}

void mco_envs_destroy(mco_envs* e) { if (!e) return; free(e->env); free(e->env_model); free(e); }

void mco_envs_get_state(const mco_envs* e, double* qpos, double* qvel, double* ctrl, double* warm,
                        double* qpos_lag, double* goal, int32_t* elapsed, int32_t* episode) {
  int nq = e->model.nq, nv = e->model.nv, nu = e->model.nu;
  for (int i = 0; i < e->cfg.n_envs; i++) {
    const env_t* v = &e->env[i];
    if (qpos) memcpy(qpos + (size_t)i * nq, v->d.qpos, sizeof(double) * nq);
    if (qvel) memcpy(qvel + (size_t)i * nv, v->d.qvel, sizeof(double) * nv);
    if (ctrl) memcpy(ctrl + (size_t)i * nu, v->d.ctrl, sizeof(double) * nu);
    if (warm) memcpy(warm + (size_t)i * nv, v->d.qacc_warmstart, sizeof(double) * nv);
    if (qpos_lag) memcpy(qpos_lag + (size_t)i * nq, v->qpos_lag, sizeof(double) * nq);
    if (goal) memcpy(goal + (size_t)i * 3, v->goal, sizeof(double) * 3);
    if (elapsed) elapsed[i] = v->elapsed;
    if (episode) episode[i] = v->episode;
  }
}

void mco_envs_set_state(mco_envs* e, const double* qpos, const double* qvel, const double* ctrl,
                        const double* warm, const double* qpos_lag, const double* goal,
                        const int32_t* elapsed, const int32_t* episode) {
  int nq = e->model.nq, nv = e->model.nv, nu = e->model.nu;
#pragma omp parallel for schedule(static) num_threads(e->cfg.n_threads > 0 ? e->cfg.n_threads : 1)
  for (int i = 0; i < e->cfg.n_envs; i++) {
    env_t* v = &e->env[i];
    const mco_model* m = model_of(e, i);
    if (ctrl) memcpy(v->d.ctrl, ctrl + (size_t)i * nu, sizeof(double) * nu);
    if (qvel) memcpy(v->d.qvel, qvel + (size_t)i * nv, sizeof(double) * nv);
    if (warm) memcpy(v->d.qacc_warmstart, warm + (size_t)i * nv, sizeof(double) * nv);
    /* derived arrays (site poses, Jacobians) are those of the last forward pass: rebuild them at
       qpos_lag, then install the actual positions */
    const double* ql = qpos_lag ? qpos_lag + (size_t)i * nq : (qpos ? qpos + (size_t)i * nq : v->d.qpos);
    double keep[MCO_MAXNV];
    memcpy(keep, v->d.qacc_warmstart, sizeof(double) * nv);
    memcpy(v->d.qpos, ql, sizeof(double) * nq);
    memcpy(v->qpos_lag, ql, sizeof(double) * nq);
    mco_forward(m, &v->d);
    memcpy(v->d.qacc_warmstart, keep, sizeof(double) * nv);
    if (qpos) memcpy(v->d.qpos, qpos + (size_t)i * nq, sizeof(double) * nq);
    if (goal) memcpy(v->goal, goal + (size_t)i * 3, sizeof(double) * 3);
    if (elapsed) { v->elapsed = elapsed[i]; }
    if (episode) v->episode = episode[i];
  }
}
