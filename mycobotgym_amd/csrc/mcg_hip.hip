// mcg_hip.hip -- kernels and C ABI of the MI355X rollout engine (see include/mcg.h for the boundary).
//
// One environment per lane, 64-lane workgroups, struct-of-arrays state in HBM ([field][N], N fastest, so every
// state load/store of a wave is one contiguous 512-byte row).  A whole env.step() -- controller, all physics
// sub-steps, observation, reward, termination, TimeLimit, auto-reset with Philox sampling -- is ONE launch: state
// is read once and written once per env-step (B = 2*S + A + O bytes per env-step).
// gfx950 only; no CPU fallback exists on purpose.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "mcg.h"
#include "mcg_dynamics.hpp"
#include "mcg_cube.hpp"
#include "mcg_coop.hpp"
#include "model_gen.h"
#include "polytopes_gen.h"

using namespace mcg;

namespace {

thread_local char g_err[512] = "";
int fail(int code, const char* fmt, const char* a = "") { snprintf(g_err, sizeof(g_err), fmt, a); return code; }
#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(MCG_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); } while (0)

// ------------------------------------------------------------------------------------------- device-side views
struct Cfg {
  int n, has_object, controller, fetch, reward_type, frame_skip, control_steps, max_episode_steps;
  int target_in_the_air, auto_reset, nq, nv, obs_dim, act_dim, dr_enable, block_gripper;
  int coop_pair;         // PickAndPlace: the cooperative phase solves two environments per wave (default; MCG_COOP_PAIR=0: one per wave, the first implementation)
  int hidden;            // Reach with reward_shaping: the cube stays in the physics as a hidden free body (mycobot.py:475-481)
  double dr_mass[2], dr_fric[2], qpos0_cube[7];
  double distance_threshold, height_offset, igx[3], dt, grip_center, grip_range;
  double init_qpos[19], init_qvel[18], init_ctrl[7];
  unsigned long long seed;
  long long env_id_offset;
  unsigned long long* cnt;   // device: mcg_counters (reset-cap hits, bad-state resets, contacts dropped by the cap, flagged env-sub-steps)
};

struct View {           // SoA state: field f of env i at d[f * n + i]
  double* d; int32_t* i32; int n, nq, nv;
  __device__ double& qpos(int k, int i) const { return d[(size_t)k * n + i]; }
  __device__ double& qvel(int k, int i) const { return d[(size_t)(nq + k) * n + i]; }
  __device__ double& ctrl(int k, int i) const { return d[(size_t)(nq + nv + k) * n + i]; }
  __device__ double& warm(int k, int i) const { return d[(size_t)(nq + nv + 7 + k) * n + i]; }
  __device__ double& qlag(int k, int i) const { return d[(size_t)(nq + 2 * nv + 7 + k) * n + i]; }
  __device__ double& goal(int k, int i) const { return d[(size_t)(2 * nq + 2 * nv + 7 + k) * n + i]; }
  __device__ double& epret(int i) const { return d[(size_t)(2 * nq + 2 * nv + 10) * n + i]; }
  __device__ double& dr(int k, int i) const { return d[(size_t)(2 * nq + 2 * nv + 11 + k) * n + i]; }
  __device__ int32_t& elapsed(int i) const { return i32[i]; }
  __device__ int32_t& episode(int i) const { return i32[n + i]; }
  __device__ int32_t& eplen(int i) const { return i32[2 * n + i]; }
};
inline int state_doubles(int nq, int nv) { return 2 * nq + 2 * nv + 13; }

struct Env {            // one lane's working set
  Robot R;
  real qlag6[6], goal[3], epret;
  int32_t elapsed, episode, eplen;
};

// mcg_counters: one atomic per event, behind a wave-uniform guard (events are rare)
MCG_DEV void count_event(const Cfg& C, int slot, bool ev) {
  if (__any(ev)) { if (ev && C.cnt) atomicAdd(C.cnt + slot, 1ull); }
}

// ------------------------------------------------------------------------------------------------- sampling
MCG_DEV void rng_pair(const Cfg& C, int i, int32_t episode, uint32_t draw, uint32_t stream, real& u0, real& u1) {
  unsigned long long gid = (unsigned long long)(C.env_id_offset + i);
  uint32_t r[4];
  philox4x32_10((uint32_t)gid, (uint32_t)episode, draw, stream ^ ((uint32_t)(gid >> 32) << 8),
                (uint32_t)C.seed, (uint32_t)(C.seed >> 32), r);
  u0 = (real)((((unsigned long long)r[0] << 32) | r[1]) >> 11) * (1.0 / 9007199254740992.0);
  u1 = (real)((((unsigned long long)r[2] << 32) | r[3]) >> 11) * (1.0 / 9007199254740992.0);
}

// _sample_goal (mycobot.py:238-243) with generate_random_point_inside_rectangle (utils.py:14-21); uses draws d, d+1
MCG_DEV void sample_goal(const Cfg& C, int i, int32_t episode, uint32_t draw, real* g) {
  real ux, uy, uc, uz;
  rng_pair(C, i, episode, draw, 0, ux, uy);
  rng_pair(C, i, episode, draw + 1, 0, uc, uz);
  // a + (b - a) * u as one explicit fma: rounds identically on the CPU oracle and here
  g[0] = fma(0.12 - -0.12, ux, -0.12);
  g[1] = fma(0.06 - -0.06, uy, -0.06);
  const real air = fma(0.1 - 0.0, uz, C.height_offset);
  g[2] = sel((C.target_in_the_air && uc < 0.5), air, C.height_offset);
}

// reset_model (mycobot.py:207-236), Reach: the object position stays the initial gripper xy.
// The rejection loop is wave-uniform (__any) with per-lane selects -- see the compiler hazard note in mcg_dynamics.hpp -- and
// runs until the lanes that DO reset have their goal: with desynchronised episodes that is one or two lanes of a wave per step
// (a quarter of the draws are accepted), not the slowest of 64.
MCG_DEV void reset_env(const Cfg& C, int i, Env& E, bool doit) {
  const real ox = C.igx[0], oy = C.igx[1];
  real goal[3] = {0, 0, 0};
  uint32_t draw = 0;
  bool need = doit, capped = false;
  int tries = 0;
  do {
    real g[3];
    sample_goal(C, i, E.episode, draw, g);
    const bool rej = sqrt((g[0] - ox) * (g[0] - ox) + (g[1] - oy) * (g[1] - oy)) < 0.1;
    for (int k = 0; k < 3; k++) goal[k] = sel(need, g[k], goal[k]);
    draw += sel(need, 2u, 0u);
    capped = capped || (need && rej && !(tries < 1000));       // the reference's loop is unbounded (mycobot.py:232-233): count the give-ups
    need = need && rej && (tries < 1000);
    tries++;
  } while (__any(need));
  count_event(C, 0, capped);
  for (int k = 0; k < NB; k++) { E.R.q[k] = sel(doit, C.init_qpos[k], E.R.q[k]); E.R.qd[k] = sel(doit, C.init_qvel[k], E.R.qd[k]); }
  for (int k = 0; k < 7; k++) E.R.ctrl[k] = sel(doit, C.init_ctrl[k], E.R.ctrl[k]);
  for (int k = 0; k < 6; k++) E.qlag6[k] = sel(doit, C.init_qpos[k], E.qlag6[k]);
  for (int k = 0; k < 3; k++) E.goal[k] = sel(doit, goal[k], E.goal[k]);
  E.elapsed = sel(doit, 0, E.elapsed); E.epret = sel(doit, 0.0, E.epret); E.eplen = sel(doit, 0, E.eplen);
  E.episode += sel(doit, 1, 0);
}

// mj_checkPos / mj_checkVel [RECALL]: a non-finite or huge (> 1e10) coordinate makes MuJoCo call mj_resetData (qpos0,
// zero velocity / ctrl / warm start) and carry on.  The check runs on the state a step kernel loads (a caller may have set it) and
// on every sub-step's new state (robot_substep; the cube wave checks the cube when a sub-step starts).
MCG_DEV bool guard_robot(Robot& R, real* qlag6) {
  bool bad = false;
  for (int k = 0; k < NB; k++) bad = bad || bad_value(R.q[k]) || bad_value(R.qd[k]) || bad_value(R.warm[k]);
  for (int k = 0; k < NB; k++) { R.q[k] = sel(bad, 0.0, R.q[k]); R.qd[k] = sel(bad, 0.0, R.qd[k]); R.warm[k] = sel(bad, 0.0, R.warm[k]); }
  for (int k = 0; k < 7; k++) R.ctrl[k] = sel(bad, 0.0, R.ctrl[k]);
  for (int k = 0; k < 6; k++) qlag6[k] = sel(bad, 0.0, qlag6[k]);
  return bad;
}


// _get_obs / generate_mujoco_observations for Reach (mycobot.py:245-283, 342-388): 10 numbers
MCG_DEV void observe_reach(const Cfg& C, ModelPtr P, const Env& E, real* obs, real* ag) {
  EefPose X;
  eef_forward(P, E.qlag6, X, true);
  for (int k = 0; k < 3; k++) {
    real v = 0;
    for (int j = 0; j < 6; j++) v += X.jacp[k][j] * E.R.qd[j];
    obs[k] = X.pos[k]; obs[5 + k] = v * C.dt; ag[k] = X.pos[k];
  }
  obs[3] = E.R.q[6]; obs[4] = E.R.q[8];
  obs[8] = E.R.qd[6] * C.dt; obs[9] = E.R.qd[8] * C.dt;
}

// The episode bookkeeping (goal, return, counters) is only needed after the sub-steps: the step kernels load it THERE, so that it
// is not carried -- spilled to scratch and reloaded -- across the whole sub-step loop (scratch lines end up as HBM traffic).
MCG_DEV void load_robot(const View& V, int i, Env& E) {
  for (int k = 0; k < NB; k++) { E.R.q[k] = V.qpos(k, i); E.R.qd[k] = V.qvel(k, i); E.R.warm[k] = V.warm(k, i); }
  for (int k = 0; k < 7; k++) E.R.ctrl[k] = V.ctrl(k, i);
  for (int k = 0; k < 6; k++) E.qlag6[k] = V.qlag(k, i);
}
MCG_DEV void load_episode(const View& V, int i, Env& E) {
  for (int k = 0; k < 3; k++) E.goal[k] = V.goal(k, i);
  E.epret = V.epret(i); E.elapsed = V.elapsed(i); E.episode = V.episode(i); E.eplen = V.eplen(i);
}
MCG_DEV void load_env(const View& V, int i, Env& E) { load_robot(V, i, E); load_episode(V, i, E); }
MCG_DEV void store_env(const View& V, int i, const Env& E) {
  for (int k = 0; k < NB; k++) { V.qpos(k, i) = E.R.q[k]; V.qvel(k, i) = E.R.qd[k]; V.warm(k, i) = E.R.warm[k]; }
  for (int k = 0; k < 7; k++) V.ctrl(k, i) = E.R.ctrl[k];
  for (int k = 0; k < 6; k++) V.qlag(k, i) = E.qlag6[k];
  for (int k = 0; k < 3; k++) V.goal(k, i) = E.goal[k];
  V.epret(i) = E.epret; V.elapsed(i) = E.elapsed; V.episode(i) = E.episode; V.eplen(i) = E.eplen;
}

MCG_DEV void write_obs(const mcg_step_out& O, int i, int D, const real* obs, const real* ag, const real* goal) {
  if (O.obs) for (int k = 0; k < D; k++) O.obs[(size_t)i * D + k] = obs[k];
  if (O.achieved_goal) for (int k = 0; k < 3; k++) O.achieved_goal[(size_t)i * 3 + k] = ag[k];
  if (O.desired_goal) for (int k = 0; k < 3; k++) O.desired_goal[(size_t)i * 3 + k] = goal[k];
}

// ------------------------------------------------------------------------------------------------ step kernel
// MyCobotEnv.step (mycobot.py:132-205) for Reach, controller = joint | IK.
// mocap branch of step (mycobot.py:172-189) with gymnasium_robotics' mocap_set_action [RECALL]: the mocap pose is reset to the
// pose of the welded body as of the last forward pass (lagged angles), then moved by (0.1 a[:3], quat - xquat_tcp); the stored
// quaternion is normalised by the next mj_kinematics.  quat = a[3:7], or the fixed fetch orientation.
MCG_DEV void mocap_target(const Cfg& C, ModelPtr P, const real* qlag6, const float* act, Weld& W) {
  TcpPose X; tcp_forward(P, qlag6, X, false);
  for (int k = 0; k < 3; k++) W.pos[k] = X.pos[k] + (real)(act[k] * 0.1f);           // f32 product, as numpy computes it
  const real fq[4] = {0.5, -0.5, -0.5, 0.5};
  for (int k = 0; k < 4; k++) {
    const real qt = C.fetch ? fq[k] : (real)act[3 + k];
    const real dq = qt - X.quat[k];
    W.quat[k] = X.quat[k] + dq;
  }
  normalize4(W.quat);
}

// SPLIT: three waves per 64 environments (see SplitMain in mcg_dynamics.hpp): launched when the grid has at most one workgroup
// per CU, where the extra waves run on SIMDs that would idle.  144 KB of LDS per workgroup.
template <int CONTROLLER, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 192 : 64) void step_reach_kernel(Cfg C, View V, const mcg_model* __restrict__ Pg,
                                                                      const float* __restrict__ actions, mcg_step_out O) {
  typedef std::conditional_t<SPLIT, SplitMain, NoSplit> Split;
  __shared__ real lds[SPLIT ? LDS_SLOTS_SPLIT : LDS_SLOTS][64];
  const int lane = threadIdx.x & 63;
  const LaneScratch MS(&lds[0][lane]);
  const ModelPtr P = as_model_ptr(Pg);
  const int i = blockIdx.x * 64 + lane;
  if (i >= C.n) return;                          // the same lanes leave in both waves: barriers stay matched
  if constexpr (SPLIT) {
    if (threadIdx.x >= 64) {                     // helper wave: M and the Euler factor; RNE wave: passive - bias forces
      if constexpr (CONTROLLER == MCG_CTRL_IK) {
        // The IK controller (S2 / S3: mycobot.py:134-170, utils.py:499-556) runs on the RNE wave: the main wave carries the robot's whole
        // state through the sub-step loop, and the controller's Jacobians and 6 x 6 solve on top of it were that kernel's scratch frame
        // (612 B a lane; the joint kernel's is 12).  Per control step: C1 -- the lagged q of the last sub-step is in LDS -- the RNE wave
        // solves and leaves the six increments in the LDS_IKT slots -- C2 -- the main wave adds them to its ctrl.  The target pose is
        // formed once from the first pose and stays in the RNE wave's registers; the helper wave only passes the two barriers.
        if (threadIdx.x < 128) {
          for (int c = 0; c < C.control_steps; c++) { __syncthreads(); __syncthreads(); for (int s = 0; s < C.frame_skip; s++) helper_substep(P, MS); }
        } else {
          float act[8];
          _Pragma("unroll") for (int k = 0; k < 8; k++) {
            const float x = (k < C.act_dim) ? actions[(size_t)i * C.act_dim + (k < C.act_dim ? k : 0)] : 0.f;
            act[k] = fminf(fmaxf(x, -1.f), 1.f);
          }
          real tpos[3], tquat[4];
          for (int c = 0; c < C.control_steps; c++) {
            __syncthreads();                                        // C1
            real ql[6]; EefPose X;
            for (int k = 0; k < 6; k++) ql[k] = MS.ld(LDS_QLAG + k);
            eef_forward(P, ql, X, true);
            if (c == 0) {
              for (int k = 0; k < 3; k++) tpos[k] = X.pos[k] + (real)(act[k] * 0.2f);      // f32 product, as numpy computes it
              if (C.fetch) { tquat[0] = 0; tquat[1] = -0.707; tquat[2] = 0; tquat[3] = 0.707; }
              else {
                real e[3], qr[4], cur[4];
                for (int k = 0; k < 3; k++) e[k] = (real)(act[3 + k] * 0.5f);
                euler2quat(e, qr); mat2quat(X.mat, cur); mulquat(qr, cur, tquat);
              }
            }
            real dq[6];
            ik_delta(X, tpos, tquat, dq);
            for (int k = 0; k < 6; k++) MS.st(LDS_IKT + k, dq[k]);
            __syncthreads();                                        // C2
            for (int s = 0; s < C.frame_skip; s++) rne_substep(P, MS);
          }
        }
      } else {
        const int total = C.frame_skip;
        if (threadIdx.x < 128) { for (int s = 0; s < total; s++) helper_substep(P, MS); }
        else { for (int s = 0; s < total; s++) rne_substep(P, MS); }
      }
      return;
    }
  }
  MCG_TICK_INIT();
  Env E;
  load_robot(V, i, E);
  const bool bad0 = guard_robot(E.R, E.qlag6);      // the state as loaded (a caller may have set it): mj_step's first check
  bool hadbad = bad0;
  if constexpr (SPLIT) static_for<NB>([&](auto I) { constexpr int k = I; MS.st(LDS_QB + k, E.R.q[k]); MS.st(LDS_QDB + k, E.R.qd[k]); MS.st(LDS_WARM + k, E.R.warm[k]); if constexpr (k < 6) MS.st(LDS_QLAG + k, E.qlag6[k]); });   // q(0), qd(0) for the other waves; warm start and lagged q parked
  MCG_TICK(ST_LOAD);
  float act[8];
  _Pragma("unroll") for (int k = 0; k < 8; k++) {   // act_dim is 7, 4 (fetch) or 8 (mocap): static indices keep the array in registers
    const float x = (k < C.act_dim) ? actions[(size_t)i * C.act_dim + (k < C.act_dim ? k : 0)] : 0.f;
    act[k] = fminf(fmaxf(x, -1.f), 1.f);
  }
  const float act_last = C.act_dim == 8 ? act[7] : (C.act_dim == 7 ? act[6] : act[3]);     // the gripper command

  if constexpr (CONTROLLER == MCG_CTRL_IK && SPLIT) {
    const real grip = C.grip_center + (real)act_last * C.grip_range;
    for (int c = 0; c < C.control_steps; c++) {
      __syncthreads();                                              // C1: the lagged q is in LDS (from load_robot, or the last sub-step)
      __syncthreads();                                              // C2: the RNE wave's increments are
      for (int k = 0; k < 6; k++) E.R.ctrl[k] += MS.ld(LDS_IKT + k);
      E.R.ctrl[6] = grip;
      if (c == 0) for (int k = 0; k < 7; k++) E.R.ctrl[k] = sel(bad0, 0.0, E.R.ctrl[k]);      // mj_resetData inside the first mj_step, after data.ctrl was written
      MCG_TICK(ST_CTRL);
      for (int s = 0; s < C.frame_skip; s++) hadbad |= robot_substep<LaneScratch, NoCoupling, NoWeld, Split>(P, E.R, E.qlag6, MS);
    }
  } else if constexpr (CONTROLLER == MCG_CTRL_IK) {
    EefPose X;
    eef_forward(P, E.qlag6, X, true);
    real tpos[3], tquat[4];
    for (int k = 0; k < 3; k++) tpos[k] = X.pos[k] + (real)(act[k] * 0.2f);      // f32 product, as numpy computes it
    if (C.fetch) { tquat[0] = 0; tquat[1] = -0.707; tquat[2] = 0; tquat[3] = 0.707; }
    else {
      real e[3], qr[4], cur[4];
      for (int k = 0; k < 3; k++) e[k] = (real)(act[3 + k] * 0.5f);
      euler2quat(e, qr); mat2quat(X.mat, cur); mulquat(qr, cur, tquat);
    }
    const real grip = C.grip_center + (real)act_last * C.grip_range;
    for (int c = 0; c < C.control_steps; c++) {
      if (c > 0) eef_forward(P, E.qlag6, X, true);
      real dq[6];
      ik_delta(X, tpos, tquat, dq);
      for (int k = 0; k < 6; k++) E.R.ctrl[k] += dq[k];
      E.R.ctrl[6] = grip;
      if (c == 0) for (int k = 0; k < 7; k++) E.R.ctrl[k] = sel(bad0, 0.0, E.R.ctrl[k]);      // mj_resetData inside the first mj_step, after data.ctrl was written
      MCG_TICK(ST_CTRL);
      for (int s = 0; s < C.frame_skip; s++) hadbad |= robot_substep<LaneScratch, NoCoupling, NoWeld, Split>(P, E.R, E.qlag6, MS);
    }
  } else if constexpr (CONTROLLER == MCG_CTRL_MOCAP) {
    Weld W; mocap_target(C, P, E.qlag6, act, W);
    E.R.ctrl[6] = sel(bad0, 0.0, C.grip_center + (real)act_last * C.grip_range);
    MCG_TICK(ST_CTRL);
    for (int s = 0; s < C.frame_skip; s++) hadbad |= robot_substep<LaneScratch, NoCoupling, Weld, Split>(P, E.R, E.qlag6, MS, nullptr, &W);
  } else {
    for (int k = 0; k < 7; k++) E.R.ctrl[k] = sel(bad0, 0.0, (real)act[k]);      // (mj_resetData inside mj_step zeroes the ctrl just written)
    MCG_TICK(ST_CTRL);
    for (int s = 0; s < C.frame_skip; s++) hadbad |= robot_substep<LaneScratch, NoCoupling, NoWeld, Split>(P, E.R, E.qlag6, MS);
  }

  if constexpr (SPLIT) static_for<NB>([&](auto I) { constexpr int k = I; E.R.warm[k] = MS.ld(LDS_WARM + k); if constexpr (k < 6) E.qlag6[k] = MS.ld(LDS_QLAG + k); });     // warm start and lagged q back from LDS
  hadbad |= guard_robot(E.R, E.qlag6);
  if (__any(hadbad)) {                              // a reset in the last sub-step: the positions of "the last forward pass" are the reset ones
    bool zero = true;
    for (int k = 0; k < NB; k++) zero = zero && E.R.q[k] == 0.0 && E.R.qd[k] == 0.0;
    for (int k = 0; k < 6; k++) E.qlag6[k] = sel(hadbad && zero, 0.0, E.qlag6[k]);
  }
  count_event(C, 1, hadbad);
  // The row addresses of the state arrays must be RECOMPUTED here, not carried: the compiler otherwise keeps the 49 addresses it formed
  // for load_robot alive across the whole sub-step loop for store_env -- spilled, they were the kernel's entire scratch frame (396 B
  // per lane) and, as scratch lines, half of its HBM traffic.  An opaque copy of the env index cuts the common subexpressions.
  int i_tail = i; asm volatile("" : "+v"(i_tail));
  load_episode(V, i_tail, E);
  if (C.block_gripper) {       // _step_callback (mycobot.py:300-306): finger joints := 0, then mj_forward removes the lag
    E.R.q[7] = 0; E.R.q[9] = 0;
    for (int k = 0; k < 6; k++) E.qlag6[k] = E.R.q[k];
  }
  real obs[10], ag[3];
  observe_reach(C, P, E, obs, ag);
  real dx = ag[0] - E.goal[0], dy = ag[1] - E.goal[1], dz = ag[2] - E.goal[2];
  const real dist = sqrt(dx * dx + dy * dy + dz * dz);                       // goal_distance, utils.py:24-26
  const bool succ = dist < C.distance_threshold;                            // _is_success, mycobot.py:285-287
  const real rew = sel(C.reward_type == MCG_REWARD_SPARSE, -(real)(float)(dist > C.distance_threshold), -dist);
  E.elapsed++; E.eplen++; E.epret += rew;
  const bool term = succ;                                                   // compute_terminated, :390-394
  const bool trunc = succ || (E.elapsed >= C.max_episode_steps);            // compute_truncated :396-400 | TimeLimit
  if (O.reward) O.reward[i] = rew;
  if (O.terminated) O.terminated[i] = term;
  if (O.truncated) O.truncated[i] = trunc;
  if (O.is_success) O.is_success[i] = succ;
  if (O.ep_return) O.ep_return[i] = E.epret;
  if (O.ep_length) O.ep_length[i] = E.eplen;
  const bool done = (term || trunc) && C.auto_reset;
  if (__any(done)) {                       // wave-uniform; per-lane effects are predicated on `done`
    if (done) {                            // plain stores of live registers only
      if (O.final_obs) for (int k = 0; k < 10; k++) O.final_obs[(size_t)i * 10 + k] = obs[k];
      if (O.final_achieved) for (int k = 0; k < 3; k++) O.final_achieved[(size_t)i * 3 + k] = ag[k];
      if (O.final_desired) for (int k = 0; k < 3; k++) O.final_desired[(size_t)i * 3 + k] = E.goal[k];
    }
    reset_env(C, i, E, done);
    real obs2[10], ag2[3];
    observe_reach(C, P, E, obs2, ag2);
    for (int k = 0; k < 10; k++) obs[k] = sel(done, obs2[k], obs[k]);
    for (int k = 0; k < 3; k++) ag[k] = sel(done, ag2[k], ag[k]);
  }
  write_obs(O, i, 10, obs, ag, E.goal);
  store_env(V, i_tail, E);
  MCG_TICK(ST_POST);
  MCG_TICK_FLUSH();
}

__global__ __launch_bounds__(64) void reset_reach_kernel(Cfg C, View V, const mcg_model* __restrict__ Pg,
                                                         const uint8_t* __restrict__ mask, int reseed, mcg_step_out O) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= C.n) return;
  const ModelPtr P = as_model_ptr(Pg);
  Env E;
  load_env(V, i, E);
  const bool doit = !mask || mask[i];
  E.episode = sel((doit && reseed), 0, E.episode);
  reset_env(C, i, E, doit);
  store_env(V, i, E);
  real obs[10], ag[3];
  observe_reach(C, P, E, obs, ag);
  write_obs(O, i, 10, obs, ag, E.goal);
}


// ===================================================================================== PickAndPlace (has_object)
struct EnvP {
  Robot R; Cube Cb;
  real qlag6[6], qlag7[7], goal[3], epret, dr[2];
  int32_t elapsed, episode, eplen;
  bool touch;       // both finger pads touched the cube in the last forward pass (stage_rewards' grasp test)
};

MCG_DEV void load_envp(const View& V, int i, EnvP& E) {
  for (int k = 0; k < NB; k++) { E.R.q[k] = V.qpos(k, i); E.R.qd[k] = V.qvel(k, i); E.R.warm[k] = V.warm(k, i); }
  for (int k = 0; k < 7; k++) E.R.ctrl[k] = V.ctrl(k, i);
  for (int k = 0; k < 3; k++) E.Cb.pos[k] = V.qpos(12 + k, i);
  for (int k = 0; k < 4; k++) E.Cb.quat[k] = V.qpos(15 + k, i);
  for (int k = 0; k < 6; k++) { E.Cb.vel[k] = V.qvel(12 + k, i); E.Cb.warm[k] = V.warm(12 + k, i); E.qlag6[k] = V.qlag(k, i); }
  for (int k = 0; k < 7; k++) E.qlag7[k] = V.qlag(12 + k, i);
  E.dr[0] = V.dr(0, i); E.dr[1] = V.dr(1, i);
}
MCG_DEV void load_episodep(const View& V, int i, EnvP& E) {        // after the sub-steps, like load_episode
  for (int k = 0; k < 3; k++) E.goal[k] = V.goal(k, i);
  E.epret = V.epret(i); E.elapsed = V.elapsed(i); E.episode = V.episode(i); E.eplen = V.eplen(i);
}
MCG_DEV void store_envp(const View& V, int i, const EnvP& E) {
  for (int k = 0; k < NB; k++) { V.qpos(k, i) = E.R.q[k]; V.qvel(k, i) = E.R.qd[k]; V.warm(k, i) = E.R.warm[k]; }
  for (int k = 0; k < 7; k++) V.ctrl(k, i) = E.R.ctrl[k];
  for (int k = 0; k < 3; k++) V.qpos(12 + k, i) = E.Cb.pos[k];
  for (int k = 0; k < 4; k++) V.qpos(15 + k, i) = E.Cb.quat[k];
  for (int k = 0; k < 6; k++) { V.qvel(12 + k, i) = E.Cb.vel[k]; V.warm(12 + k, i) = E.Cb.warm[k]; V.qlag(k, i) = E.qlag6[k]; }
  for (int k = 0; k < 7; k++) V.qlag(12 + k, i) = E.qlag7[k];
  for (int k = 0; k < 3; k++) V.goal(k, i) = E.goal[k];
  V.dr(0, i) = E.dr[0]; V.dr(1, i) = E.dr[1];
  V.epret(i) = E.epret; V.elapsed(i) = E.elapsed; V.episode(i) = E.episode; V.eplen(i) = E.eplen;
}

// reset_model with an object (mycobot.py:207-236): cube xy resampled until >= 0.1 from the initial gripper xy,
// goal until >= 0.1 from the cube; per-reset domain randomisation (build-defined) on its own stream.
MCG_DEV void reset_envp(const Cfg& C, int i, EnvP& E, bool doit) {
  real drs[2] = {1.0, 1.0};
  if (C.dr_enable) {
    real um, uf; rng_pair(C, i, E.episode, 0, 1, um, uf);
    drs[0] = C.dr_mass[0] + (C.dr_mass[1] - C.dr_mass[0]) * um;
    drs[1] = C.dr_fric[0] + (C.dr_fric[1] - C.dr_fric[0]) * uf;
  }
  real oxy[2] = {C.igx[0], C.igx[1]}, goal[3] = {0, 0, 0};
  uint32_t draw = 0;
  bool need = doit && !C.hidden, capped = false; int tries = 0;     // hidden cube (Reach): reset_model places nothing (mycobot.py:216: `if self.has_object`)
  do {                                              // object position (mycobot.py:217-219)
    real g[3]; sample_goal(C, i, E.episode, draw, g);
    const bool rej = sqrt((g[0] - C.igx[0]) * (g[0] - C.igx[0]) + (g[1] - C.igx[1]) * (g[1] - C.igx[1])) < 0.1;
    oxy[0] = sel(need, g[0], oxy[0]); oxy[1] = sel(need, g[1], oxy[1]);
    draw += sel(need, 2u, 0u);
    capped = capped || (need && rej && !(tries + 1 < 1000));
    need = need && rej && (tries + 1 < 1000);
    tries++;
  } while (__any(need));
  const real cxy[2] = {sel(C.hidden, C.init_qpos[12], oxy[0]), sel(C.hidden, C.init_qpos[13], oxy[1])};
  need = doit; tries = 0;
  do {                                              // goal (mycobot.py:231-233)
    real g[3]; sample_goal(C, i, E.episode, draw, g);
    const bool rej = sqrt((g[0] - oxy[0]) * (g[0] - oxy[0]) + (g[1] - oxy[1]) * (g[1] - oxy[1])) < 0.1;
    for (int k = 0; k < 3; k++) goal[k] = sel(need, g[k], goal[k]);
    draw += sel(need, 2u, 0u);
    capped = capped || (need && rej && !(tries < 1000));
    need = need && rej && (tries < 1000);
    tries++;
  } while (__any(need));
  count_event(C, 0, capped);
  for (int k = 0; k < NB; k++) { E.R.q[k] = sel(doit, C.init_qpos[k], E.R.q[k]); E.R.qd[k] = sel(doit, C.init_qvel[k], E.R.qd[k]); }
  for (int k = 0; k < 7; k++) E.R.ctrl[k] = sel(doit, C.init_ctrl[k], E.R.ctrl[k]);
  for (int k = 0; k < 6; k++) { E.qlag6[k] = sel(doit, C.init_qpos[k], E.qlag6[k]); E.Cb.vel[k] = sel(doit, C.init_qvel[12 + k], E.Cb.vel[k]); }
  E.Cb.pos[0] = sel(doit, cxy[0], E.Cb.pos[0]); E.Cb.pos[1] = sel(doit, cxy[1], E.Cb.pos[1]); E.Cb.pos[2] = sel(doit, C.init_qpos[14], E.Cb.pos[2]);
  {   // mj_forward normalises the stored quaternion
    real q[4] = {C.init_qpos[15], C.init_qpos[16], C.init_qpos[17], C.init_qpos[18]};
    const real nq = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    for (int k = 0; k < 4; k++) E.Cb.quat[k] = sel(doit, q[k] / nq, E.Cb.quat[k]);
  }
  for (int k = 0; k < 3; k++) E.qlag7[k] = sel(doit, E.Cb.pos[k], E.qlag7[k]);
  for (int k = 0; k < 4; k++) E.qlag7[3 + k] = sel(doit, E.Cb.quat[k], E.qlag7[3 + k]);
  for (int k = 0; k < 3; k++) E.goal[k] = sel(doit, goal[k], E.goal[k]);
  E.dr[0] = sel(doit, drs[0], E.dr[0]); E.dr[1] = sel(doit, drs[1], E.dr[1]);
  E.elapsed = sel(doit, 0, E.elapsed); E.epret = sel(doit, 0.0, E.epret); E.eplen = sel(doit, 0, E.eplen);
  E.episode += sel(doit, 1, 0);
}

// _get_obs with an object (mycobot.py:245-283, 342-388): 25 numbers, Appendix A.6 order
MCG_DEV void observe_pnp(const Cfg& C, ModelPtr P, const EnvP& E, real* obs, real* ag) {
  EefPose X;
  eef_forward(P, E.qlag6, X, true);
  real gv[3];
  for (int k = 0; k < 3; k++) { real v = 0; for (int j = 0; j < 6; j++) v += X.jacp[k][j] * E.R.qd[j]; gv[k] = v * C.dt; }
  real Rl[9], eul[3];
  quat_to_mat(E.qlag7 + 3, Rl); mat2euler(Rl, eul);
  for (int k = 0; k < 3; k++) {
    obs[k] = X.pos[k]; obs[3 + k] = E.qlag7[k]; obs[6 + k] = E.qlag7[k] - X.pos[k];
    obs[11 + k] = eul[k];
    obs[14 + k] = E.Cb.vel[k] * C.dt - gv[k];                                         // site at the cube origin: jacp = identity
    obs[17 + k] = (Rl[3*k]*E.Cb.vel[3] + Rl[3*k+1]*E.Cb.vel[4] + Rl[3*k+2]*E.Cb.vel[5]) * C.dt;   // jacr = lagged body axes
    obs[20 + k] = gv[k];
    ag[k] = E.qlag7[k];
  }
  obs[9] = E.R.q[6]; obs[10] = E.R.q[8];
  obs[23] = E.R.qd[6] * C.dt; obs[24] = E.R.qd[8] * C.dt;
}

// Reach with a hidden cube: the observation is Reach's 10 numbers (grip_pos, gripper_state, grip_velp, gripper_vel -- the
// object entries are empty when has_object is false, mycobot.py:258-259, 368-371) and the achieved goal is the gripper's position.
MCG_DEV void hide_object(real* obs, real* ag) {
  const real o[10] = {obs[0], obs[1], obs[2], obs[9], obs[10], obs[20], obs[21], obs[22], obs[23], obs[24]};
  for (int k = 0; k < 10; k++) obs[k] = o[k];
  for (int k = 0; k < 3; k++) ag[k] = o[k];
}

// ------------------------------------------------------------------------------- four-wave PickAndPlace
// Without a contact that reaches the robot the cube's sub-step (collision, 6x6 Newton, integration) and the robot's are independent.
// The workgroup has four waves over the same 32 environments: ROBOT (the robot pipeline, speculatively: results held back), CUBE
// (owns the cube for the whole env-step), M (composite rigid bodies) and RNE (bias forces), as in the Reach kernel.  Barriers per
// sub-step:
//   S1  q(t), qd(t) of the robot are published         (the cube wave needs the joint frames for the collision pass)
//   S1b the lane-parallel part of the collision pass is done: the cube wave has the primitive geoms' contacts in the list and its share
//       of the mesh geoms' candidate masks, the M / RNE waves theirs (the arm meshes against the table / the ground); M, bias forces
//       are in LDS.  Then the MESH PHASE (mcg_mesh.hpp): the cube, M and RNE waves, 64 lanes each, take the environments that have
//       candidates and run the exact narrow phase one pair at a time, appending to the lists
//   S1c the lists are complete: the three waves share the per-contact solver numbers; the cube wave sets each lane's FLAG
//   S2  M, passive - bias, the collision results and the flags are published
//   S4  both sides are done; every wave reads the same flags
//   S5  only when some lane is flagged: the cooperative coupled solves are done
// Routing is PER ENVIRONMENT.  An unflagged lane commits: the robot wave its speculative sub-step, the cube wave its cube-alone solve.
// A flagged lane's cube is left as it was and handed over in LDS, the robot wave parks the inputs of the coupled solve next to it
// (PubHook, right after S2), and between S4 and S5 the four waves -- all idle at that point -- solve the flagged environments, one
// environment per 32 lanes (mcg_coop.hpp).  The robot wave then redoes the Euler step of its flagged lanes with the coupled
// acceleration, the cube wave advances their cubes with theirs.
// mj_checkPos / mj_checkVel / mj_checkAcc (every mj_step: mycobot.py:170,189,193) ride on values the waves exchange anyway: the robot
// wave leaves its verdict on the state it has just produced in XCH_T1 (the cube wave reads it with q(t) after S1 and resets the cube),
// the cube wave adds its verdict on the cube as bit 3 of the flag it publishes before S2 (the robot wave reads the flag after S4 and
// resets the robot's new state).  mj_resetData resets BOTH bodies; the other body follows one sub-step late here.
// Exchange slots: mcg_coop.hpp.
constexpr int XCH_FS = PNP_SLOTS;                 // + 12 slots, then the mesh phase's twelve (MP_CUBE ..), then four of the parked limit rows
constexpr int PNP_SLOTS_DUAL = PNP_SLOTS_LDS;
static_assert(PNP_SLOTS_DUAL * PNP_LANES * 8 <= 160 * 1024, "LDS of a CU");
// robot-side split of the four-wave PickAndPlace kernel: M from the helper wave, passive - bias from the RNE wave, the constraint
// part of H_eq assembled before barrier S2; the Euler step stays with M a (no room for the factor in LDS)
struct SplitPnp { static constexpr bool enabled = true, rne_remote = true, factor_remote = false, early_heq = true, warm_lds = false, mesh_split = true;
                  static constexpr int QB = XCH_Q, QDB = XCH_QD, FS = XCH_FS, WARM = 0, QLAG = 0, MASK0 = MP_MASK; };

MCG_DEV bool flag_coupled(real f) { return (((int)f) & 2) != 0; }       // XCH_FLAG: bit 1 = the environment goes to the cooperative solve,
MCG_DEV bool flag_cube_bad(real f) { return (((int)f) & 8) != 0; }      // bit 3 = the cube failed mj_checkPos / mj_checkVel / mj_checkAcc

// the robot wave's hook into robot_substep: park a lane's Newton inputs for the cooperative solve (slots: mcg_coop.hpp).  flags_known: the
// cube wave has published the flags (after S2): only the flagged lanes are parked; else (the robot wave runs ahead of the mesh phase)
// every lane is
struct PubHook {
  static constexpr bool enabled = false, publishes = true;
  const PnpScratch S;
  MCG_DEV void publish(const real* g0, const real* Dl, const real* arefl, const real* sgl, const real* qd, const real* warm, bool flags_known) const {
    const bool flag = !flags_known || flag_coupled(S.ld(XCH_FLAG));
    if (__any(flag)) {                                               // wave-uniform
      if (flag) {                                                    // plain LDS stores of live registers
        static_for<NB>([&](auto I) { constexpr int k = I; S.st(PUB_G0 + k, g0[k]); S.st(PUB_WARM + k, warm[k]); });      // (qd(t) is in its exchange slots)
        static_for<10>([&](auto I) { constexpr int j = I; S.st(PUB_SD + j, sgl[j] * Dl[j]); S.st(pub_aref(j), arefl[j]); });
      }
    }
  }
};

MCG_DEV void cube_to_lds(const PnpScratch MS, const Cube& Cb) {
  for (int k = 0; k < 3; k++) MS.st(XCH_CB + k, Cb.pos[k]);
  for (int k = 0; k < 4; k++) MS.st(XCH_CB + 3 + k, Cb.quat[k]);
  for (int k = 0; k < 6; k++) { MS.st(XCH_CB + 7 + k, Cb.vel[k]); MS.st(XCH_CB + 13 + k, Cb.warm[k]); }
}
MCG_DEV void cube_from_lds(const PnpScratch MS, Cube& Cb) {
  for (int k = 0; k < 3; k++) Cb.pos[k] = MS.ld(XCH_CB + k);
  for (int k = 0; k < 4; k++) Cb.quat[k] = MS.ld(XCH_CB + 3 + k);
  for (int k = 0; k < 6; k++) { Cb.vel[k] = MS.ld(XCH_CB + 7 + k); Cb.warm[k] = MS.ld(XCH_CB + 13 + k); }
}
// the flagged lanes, the same in all four waves; nvalid: the workgroup's real environments (a ragged last workgroup's surplus lanes
// shadow the last environment: same arithmetic, but they are never handed out, counted or stored)
MCG_DEV unsigned flagged_lanes(const PnpScratch MS, int nvalid) {
  return (unsigned)__ballot(flag_coupled(MS.ld(XCH_FLAG))) & (nvalid >= 32 ? 0xFFFFFFFFu : ((1u << nvalid) - 1u));
}

// the cube wave's whole env-step.  `lower`: lanes 0-31 carry the 32 environments; lanes 32-63 are alive for the mesh phase and the
// cooperative phase only
struct CubeWaveArgs { real qpos0_cube[7]; unsigned long long* cnt; int coop_pair; int nvalid; };
// (inlined into the kernel, the two phases inlined into it: the wave's own state across them is the cube system)
MCG_DEV void cube_wave(const CubeWaveArgs& C, const View& V, ModelPtr P, const real* __restrict__ poly, const PnpScratch MS, unsigned lds0, int i, int total, bool lower) {
  Cube Cb; real dr[2], qlag7[7];
  bool touch = false;
  const bool valid = (int)(threadIdx.x & (PNP_LANES - 1)) < C.nvalid;
  if (lower) {
    for (int k = 0; k < 3; k++) Cb.pos[k] = V.qpos(12 + k, i);
    for (int k = 0; k < 4; k++) Cb.quat[k] = V.qpos(15 + k, i);
    for (int k = 0; k < 6; k++) { Cb.vel[k] = V.qvel(12 + k, i); Cb.warm[k] = V.warm(12 + k, i); }
    for (int k = 0; k < 7; k++) qlag7[k] = V.qlag(12 + k, i);
    dr[0] = V.dr(0, i); dr[1] = V.dr(1, i);
    for (int k = 0; k < 3; k++) MS.st(MP_CUBE + k, Cb.pos[k]);        // for the M / RNE waves' broad phase (read after S1)
  }
  MCG_TICK2_INIT();
  for (int s = 0; s < total; s++) {
    CubeSys<PnpScratch> CS(MS, Cb, dr);
    int kind = 0;
    bool cbad = false;
    if (lower) {
      __syncthreads();                                              // S1 (a wave passes a barrier once, whatever its lanes do)
      MCG_TICK2(ST_W2_WAIT1);
      real q12[NB];
      static_for<NB>([&](auto I) { constexpr int k = I; q12[k] = MS.ld(XCH_Q + k); });
      const bool rbad = MS.ld(XCH_T1) != 0.0;                       // the robot wave's verdict on the state it published (same batch of LDS reads)
      {   // mj_checkPos / mj_checkVel / mj_checkAcc for the cube, every sub-step: the state this sub-step starts from (the acceleration the
          // last one ended with is its warm start); the robot wave's verdict resets the cube with it (mj_resetData resets both bodies)
        for (int k = 0; k < 3; k++) cbad = cbad || bad_value(Cb.pos[k]);
        for (int k = 0; k < 4; k++) cbad = cbad || bad_value(Cb.quat[k]);
        for (int k = 0; k < 6; k++) cbad = cbad || bad_value(Cb.vel[k]) || bad_value(Cb.warm[k]);
        const bool reset = cbad || rbad;
        if (__any(reset)) {                                         // wave-uniform; rare
          for (int k = 0; k < 3; k++) Cb.pos[k] = sel(reset, C.qpos0_cube[k], Cb.pos[k]);
          for (int k = 0; k < 4; k++) Cb.quat[k] = sel(reset, C.qpos0_cube[3 + k], Cb.quat[k]);
          for (int k = 0; k < 6; k++) { Cb.vel[k] = sel(reset, 0.0, Cb.vel[k]); Cb.warm[k] = sel(reset, 0.0, Cb.warm[k]); }
          CS.Cb = Cb;
          if (reset) { for (int k = 0; k < 3; k++) MS.st(MP_CUBE + k, Cb.pos[k]); }      // (the M / RNE waves' broad phase of THIS sub-step may still see the old position)
          if (__any(cbad)) { if (cbad && valid && C.cnt) atomicAdd(C.cnt + 1, 1ull); }
        }
      }
      CS.cnt = valid ? C.cnt : nullptr;
      CS.template collide_primitives<true, false>(P, q12);          // the primitive geoms' contacts; candidate pairs of the mesh geoms
      MCG_TICK2(ST_W2_COLLIDE);
    }
    __syncthreads();                                                // S1b
    MCG_TICK2(ST_CUBE_FIN);
    mesh_phase(P, poly, (LdsPtr)(uintptr_t)lds0, 0, 3);
    MCG_TICK2(ST_COLLIDE);
    __syncthreads();                                                // S1c: every list is complete
    MCG_TICK2(ST_X_S1C);
    if (lower) {
      CS.collect_list();
      touch = CS.touch[0] && CS.touch[1];
      // 0: nothing reaches the robot | 2: a contact does (a pad or a mesh on the cube, the table or the ground), or the cube's own list is
      // longer than its solve's row slots: the environment's 18 dofs go to the cooperative solve
      kind = (CS.any_pad || CS.ncon > ROW_SLOTS / 12) ? 2 : 0;
      MS.st(XCH_FLAG, (real)(kind + (cbad ? 8 : 0)));
      const unsigned nflag = (unsigned)__popc((unsigned)__ballot(kind != 0 && valid));      // (the ballot OUTSIDE the one-lane branch: rounds 2-3 took it inside
      if (nflag != 0u || __any(kind != 0)) {                                                 // and counted lane 0's flag only, 1/32 of the coupled env-sub-steps)
        if ((threadIdx.x & 63) == 0 && C.cnt && nflag) atomicAdd(C.cnt + 3, (unsigned long long)nflag);
        if (kind != 0) { MS.st(XCH_NCON, (real)CS.ncon); MS.st(XCH_DR, dr[0]); MS.st(XCH_DR + 1, dr[1]); cube_to_lds(MS, CS.Cb); }      // hand the (normalised, not advanced) cube over
      }
      solver_numbers_share(P, MS, dr[1], 0);
      MCG_TICK2(ST_X_NUMBERS);
      __syncthreads();                                              // S2 (the robot side's "M and bias ready")
      MCG_TICK2(ST_X_S2);
      CS.solve_alone(kind == 2);                                    // flagged lanes walk an empty list, store nothing
      if (!__any(kind == 2)) { CS.finish(qlag7); Cb = CS.Cb; for (int k = 0; k < 3; k++) MS.st(MP_CUBE + k, Cb.pos[k]); }
      MCG_TICK2(ST_W2_CUBE);
    }
    __syncthreads();                                                // S4: end of the lane-parallel part
    MCG_TICK2(ST_W2_WAIT2);
    const unsigned mask = flagged_lanes(MS, C.nvalid);              // all 64 lanes from here (the upper half reads the lower half's columns)
    if (mask != 0u) {
      coop_phase_body(P, (LdsPtr)(uintptr_t)lds0, __builtin_amdgcn_readfirstlane(mask), 1, C.coop_pair != 0 ? 1 : 0);
      MCG_TICK2(ST_COUPLED);
      __syncthreads();                                              // S5
      MCG_TICK2(ST_CO_IDLE);
      if (lower) {
        _Pragma("unroll") for (int k = 0; k < 6; k++) CS.a_c[k] = sel(kind == 2, MS.ld(XCH_CB + 13 + k), CS.a_c[k]);
        CS.finish(qlag7); Cb = CS.Cb;
        for (int k = 0; k < 3; k++) MS.st(MP_CUBE + k, Cb.pos[k]);
      }
    }
  }
  if (lower) {
    cube_to_lds(MS, Cb);
    for (int k = 0; k < 7; k++) MS.st(XCH_QL7 + k, qlag7[k]);
    MS.st(XCH_T0, touch ? 1.0 : 0.0);
  }
  __syncthreads();                                                  // end of the env-step: the robot wave takes the cube
}

// the robot wave's sub-step
template <class WLD>
MCG_DEV bool pnp_substep_robot(const Cfg& C, ModelPtr P, EnvP& E, const PnpScratch MS, unsigned lds0, const WLD& W, int nvalid) {
  Robot nx;
  PubHook hook{MS};
  robot_substep<PnpScratch, PubHook, WLD, SplitPnp, false>(P, E.R, E.qlag6, MS, &hook, &W, &nx);     // S1, S1b, S1c, S2 inside
  MCG_TICK(ST_POST);
  if ((threadIdx.x & 63) == 0) *(__attribute__((address_space(3))) unsigned*)(uintptr_t)(lds0 + COOP_CTR_SLOT * PNP_LANES * 8) = 0u;      // the cooperative phase's hand-out counter
  __syncthreads();                                                  // S4
  MCG_TICK(ST_W1_WAIT);
  const real fl = MS.ld(XCH_FLAG);
  const bool flag = flag_coupled(fl);
  const unsigned mask = (unsigned)__ballot(flag) & (nvalid >= 32 ? 0xFFFFFFFFu : ((1u << nvalid) - 1u));
  if (mask != 0u) {                                                 // wave-uniform, the same in all four waves
    // one environment at a time here, two in the other three waves; with fewer flagged environments than a round of the workgroup takes
    // this wave has no share (its rank-fixed share is every COOP_ROBOT_EVERY-th) and does not pay the call
    if (C.coop_pair == 0 || __popc(mask) >= COOP_ROBOT_EVERY) coop_phase((unsigned long long)P, lds0, mask, 0, C.coop_pair != 0 ? 2 : 0);
    MCG_TICK(ST_COUPLED);
    __syncthreads();                                                // S5
    MCG_TICK(ST_CO_IDLE);
    // Euler step of the flagged lanes with the coupled acceleration (M a carries the contact forces); the others keep theirs
    real a[NB], rhs[NB];
    static_for<NB>([&](auto I) { constexpr int k = I; a[k] = sel(flag, MS.ld(PUB_WARM + k), nx.warm[k]); });
    const real h = launder(P)->timestep;
    euler_accel(P, h, MS, a, rhs);
    static_for<NB>([&](auto I) { constexpr int k = I;
      const real qd_new = fma(h, rhs[k], E.R.qd[k]), q_new = fma(h, qd_new, E.R.q[k]);
      nx.qd[k] = sel(flag, qd_new, nx.qd[k]); nx.q[k] = sel(flag, q_new, nx.q[k]); nx.warm[k] = a[k]; });
    MCG_TICK(ST_EULER);
  }
  // mj_checkPos / mj_checkVel / mj_checkAcc on the robot's new state, every sub-step (the state the next mj_step would check first), and
  // the cube wave's verdict on the cube this sub-step started with (bit 3 of the flag: mj_resetData resets the robot with it)
  bool bad = false;
  static_for<NB>([&](auto I) { constexpr int k = I; bad = bad || bad_value(nx.q[k]) || bad_value(nx.qd[k]) || bad_value(nx.warm[k]); });
  const bool reset = bad || flag_cube_bad(fl);
  if (__any(reset)) {                                               // wave-uniform; rare: mj_resetData
    static_for<NB>([&](auto I) { constexpr int k = I; nx.q[k] = sel(reset, 0.0, nx.q[k]); nx.qd[k] = sel(reset, 0.0, nx.qd[k]); nx.warm[k] = sel(reset, 0.0, nx.warm[k]); });
    static_for<7>([&](auto I) { constexpr int k = I; E.R.ctrl[k] = sel(reset, 0.0, E.R.ctrl[k]); });
  }
  MS.st(XCH_T1, bad ? 1.0 : 0.0);                                   // for the cube wave, read after the next S1
  static_for<NB>([&](auto I) { constexpr int k = I; E.R.q[k] = nx.q[k]; E.R.qd[k] = nx.qd[k]; E.R.warm[k] = nx.warm[k];
                               MS.st(XCH_Q + k, nx.q[k]); MS.st(XCH_QD + k, nx.qd[k]); });
  (void)C;
  return reset;
}

// the helper / RNE waves of the four-wave PickAndPlace kernel: same barriers as the cube wave; lanes 32-63 are alive for the mesh phase
// and the cooperative phase only
// (inlined into the kernel: these waves hold nothing across a sub-step, so the inlined phases spill nothing)
// what a side wave does between S1 and S1b besides its own share: the broad phase of four arm meshes against the table, the ground and
// the cube, parking the frames of the bodies that carry a candidate
template <int P0, int P1> struct PnpSideWork {
  ModelPtr P; const PnpScratch MS; int mask_slot;
  MCG_DEV void operator()(const real* sn, const real* cs) const { arm_broad_stage<P0, P1>(P, MS, sn, cs, mask_slot); }
};
MCG_DEV void pnp_side_wave(ModelPtr P, const real* __restrict__ poly, const PnpScratch MS, unsigned lds0, int total, bool rne, bool lower, bool pair, real dr1, int nvalid) {
  MCG_TICK_INIT();
  for (int s = 0; s < total; s++) {
    if (lower) {                                                    // S1 inside; its own share, then the arm meshes' broad phase
      if (rne) rne_pre<SplitPnp>(P, MS, PnpSideWork<4, 8>{P, MS, MP_MASK + 1});
      else helper_pre<SplitPnp>(P, MS, PnpSideWork<0, 4>{P, MS, MP_MASK});
    }
    __syncthreads();                                                // S1b
    MCG_TICK(ST_S_S1B);
    mesh_phase(P, poly, (LdsPtr)(uintptr_t)lds0, rne ? 2 : 1, 3);
    MCG_TICK(ST_S_MESH);
    __syncthreads();                                                // S1c
    MCG_TICK(ST_S_S1C);
    if (lower) solver_numbers_share(P, MS, dr1, rne ? 2 : 1);
    MCG_TICK(ST_S_NUMBERS);
    __syncthreads();                                                // S2
    MCG_TICK(ST_C_SOLVE);
    __syncthreads();                                                // S4
    MCG_TICK(ST_C_LS);
    const unsigned mask = flagged_lanes(MS, nvalid);
    if (mask != 0u) {
      coop_phase_body(P, (LdsPtr)(uintptr_t)lds0, __builtin_amdgcn_readfirstlane(mask), rne ? 3 : 2, pair ? 1 : 0);
      __syncthreads();                                              // S5
    }
    MCG_TICK_INIT();                                                // (stage clocks: the cooperative phase keeps its own)
  }
  __syncthreads();                                                  // end of the env-step
}

template <int CONTROLLER>
__global__ __launch_bounds__(256) void step_pnp_kernel(Cfg C, View V, const mcg_model* __restrict__ Pg, const real* __restrict__ poly,
                                                       const float* __restrict__ actions, mcg_step_out O) {
  constexpr bool DUAL = true;                    // (the one-wave variant of rounds 1-2 went with the lane-parallel coupled solve)
  __shared__ __attribute__((aligned(16))) real lds[PNP_SLOTS_DUAL][PNP_LANES];
  // four waves over 32 environments: robot, cube, helper (M), RNE.  The robot wave runs on 32 lanes; the other three keep lanes 32-63
  // for the cooperative phase (two environments per wave there) and mask them off everywhere else (`lower`)
  if (threadIdx.x >= PNP_LANES && threadIdx.x < 64) return;
  const int lane = threadIdx.x & (PNP_LANES - 1);
  const bool lower = (threadIdx.x & PNP_LANES) == 0;
  const PnpScratch MS(&lds[0][lane]);
  const ModelPtr P = as_model_ptr(Pg);
  const int i_raw = blockIdx.x * PNP_LANES + lane;
  // every wave keeps its 32 lanes (the cooperative phases work with all of them); in a ragged last workgroup the surplus lanes shadow
  // the last environment -- same inputs, same instruction stream -- but are never handed out, counted or stored (`valid`)
  const int i = (i_raw >= C.n) ? C.n - 1 : i_raw;
  const bool valid = i_raw < C.n;
  const int nvalid = min(PNP_LANES, C.n - (int)blockIdx.x * PNP_LANES);
  const unsigned lds0 = (unsigned)(uintptr_t)(LdsPtr)&lds[0][0];
  if constexpr (DUAL) {
    if (threadIdx.x >= 64) {
      const int total = (CONTROLLER == MCG_CTRL_IK ? C.control_steps : 1) * C.frame_skip;
      if (threadIdx.x < 128) {
        CubeWaveArgs A; for (int k = 0; k < 7; k++) A.qpos0_cube[k] = C.qpos0_cube[k];
        A.cnt = C.cnt; A.coop_pair = C.coop_pair; A.nvalid = nvalid;
        cube_wave(A, V, P, poly, MS, lds0, i, total, lower);
      } else pnp_side_wave(P, poly, MS, lds0, total, threadIdx.x >= 192, lower, C.coop_pair != 0, lower ? V.dr(1, i) : 1.0, nvalid);
      return;
    }
  }
  MCG_TICK_INIT();
#ifdef MCG_STAGE_CLOCKS
  const unsigned long long wg_t0 = __builtin_readcyclecounter();
#endif
  EnvP E;
  load_envp(V, i, E);
  const bool bad0 = guard_robot(E.R, E.qlag6);   // the state as loaded (a caller may have set it): mj_step's first check
  bool hadbad = bad0;
  if constexpr (DUAL) { static_for<NB>([&](auto I) { constexpr int k = I; MS.st(XCH_Q + k, E.R.q[k]); MS.st(XCH_QD + k, E.R.qd[k]); });   // q(0), qd(0) for the other waves
                        MS.st(XCH_T1, hadbad ? 1.0 : 0.0);
                        MS.st(XCH_ACT0, 0.0); MS.st(XCH_ACT1, 0.0);         // no active set carried into an env-step (mcg_coop.hpp: coop_guess)
                      }
  MCG_TICK(ST_LOAD);
  E.touch = false;
  int nsub = 0;
  auto substep = [&](const auto& W) { hadbad |= pnp_substep_robot(C, P, E, MS, lds0, W, nvalid); nsub++; };
  float act[8];
  _Pragma("unroll") for (int k = 0; k < 8; k++) {   // act_dim is 7, 4 (fetch) or 8 (mocap): static indices keep the array in registers
    const float x = (k < C.act_dim) ? actions[(size_t)i * C.act_dim + (k < C.act_dim ? k : 0)] : 0.f;
    act[k] = fminf(fmaxf(x, -1.f), 1.f);
  }
  const float act_last = C.act_dim == 8 ? act[7] : (C.act_dim == 7 ? act[6] : act[3]);     // the gripper command
  if constexpr (CONTROLLER == MCG_CTRL_IK) {
    EefPose X;
    eef_forward(P, E.qlag6, X, true);
    real tpos[3], tquat[4];
    for (int k = 0; k < 3; k++) tpos[k] = X.pos[k] + (real)(act[k] * 0.2f);
    if (C.fetch) { tquat[0] = 0; tquat[1] = -0.707; tquat[2] = 0; tquat[3] = 0.707; }
    else {
      real e[3], qr[4], cur[4];
      for (int k = 0; k < 3; k++) e[k] = (real)(act[3 + k] * 0.5f);
      euler2quat(e, qr); mat2quat(X.mat, cur); mulquat(qr, cur, tquat);
    }
    const real grip = C.grip_center + (real)act_last * C.grip_range;
    for (int c = 0; c < C.control_steps; c++) {
      if (c > 0) eef_forward(P, E.qlag6, X, true);
      real dq[6];
      ik_delta(X, tpos, tquat, dq);
      for (int k = 0; k < 6; k++) E.R.ctrl[k] += dq[k];
      E.R.ctrl[6] = grip;
      if (c == 0) for (int k = 0; k < 7; k++) E.R.ctrl[k] = sel(bad0, 0.0, E.R.ctrl[k]);
      MCG_TICK(ST_CTRL);
      for (int s = 0; s < C.frame_skip; s++) substep(NoWeld{});
    }
  } else if constexpr (CONTROLLER == MCG_CTRL_MOCAP) {
    Weld W; mocap_target(C, P, E.qlag6, act, W);
    E.R.ctrl[6] = sel(bad0, 0.0, C.grip_center + (real)act_last * C.grip_range);
    MCG_TICK(ST_CTRL);
    for (int s = 0; s < C.frame_skip; s++) substep(W);
  } else {
    for (int k = 0; k < 7; k++) E.R.ctrl[k] = sel(bad0, 0.0, (real)act[k]);
    MCG_TICK(ST_CTRL);
    for (int s = 0; s < C.frame_skip; s++) substep(NoWeld{});
  }
  if constexpr (DUAL) {                          // take the cube back from the cube wave
    __syncthreads();
    cube_from_lds(MS, E.Cb);
    for (int k = 0; k < 7; k++) E.qlag7[k] = MS.ld(XCH_QL7 + k);
    E.touch = MS.ld(XCH_T0) != 0.0;
  }
  hadbad |= guard_robot(E.R, E.qlag6);
  count_event(C, 1, hadbad && valid);
  int i_tail = i; asm volatile("" : "+v"(i_tail));      // recompute the state rows' addresses for the tail (see step_reach_kernel)
  load_episodep(V, i_tail, E);
  {   // same guard for the cube: back to its model pose at rest
    bool bad = false;
    for (int k = 0; k < 3; k++) bad = bad || bad_value(E.Cb.pos[k]);
    for (int k = 0; k < 4; k++) bad = bad || bad_value(E.Cb.quat[k]);
    for (int k = 0; k < 6; k++) bad = bad || bad_value(E.Cb.vel[k]) || bad_value(E.Cb.warm[k]);
    for (int k = 0; k < 3; k++) { E.Cb.pos[k] = sel(bad, C.qpos0_cube[k], E.Cb.pos[k]); E.qlag7[k] = sel(bad, C.qpos0_cube[k], E.qlag7[k]); }
    for (int k = 0; k < 4; k++) { E.Cb.quat[k] = sel(bad, C.qpos0_cube[3 + k], E.Cb.quat[k]); E.qlag7[3 + k] = sel(bad, C.qpos0_cube[3 + k], E.qlag7[3 + k]); }
    for (int k = 0; k < 6; k++) { E.Cb.vel[k] = sel(bad, 0.0, E.Cb.vel[k]); E.Cb.warm[k] = sel(bad, 0.0, E.Cb.warm[k]); }
  }
  if (C.block_gripper) {       // _step_callback: finger joints := 0, mj_forward (poses, contacts of the new state)
    E.R.q[7] = 0; E.R.q[9] = 0;
    for (int k = 0; k < 6; k++) E.qlag6[k] = E.R.q[k];
    CubeSys<PnpScratch> CS(MS, E.Cb, E.dr);
    CS.template collide_primitives<false, false>(P, E.R.q);          // (stage_rewards reads the pads' contacts with the cube: primitive pairs)
    CS.scan_list();
    E.Cb = CS.Cb;
    for (int k = 0; k < 3; k++) E.qlag7[k] = E.Cb.pos[k];
    for (int k = 0; k < 4; k++) E.qlag7[3 + k] = E.Cb.quat[k];
    E.touch = CS.touch[0] && CS.touch[1];
  }
  real obs[25], ag[3];
  observe_pnp(C, P, E, obs, ag);
  real rew_shaped = 0;
  if (C.reward_type == MCG_REWARD_SHAPING) {
    // stage_rewards (mycobot.py:402-448): reach 0.2 (1 - tanh d), grasp 0.5 iff both pads touch the cube, lift
    // 0.5 + 0.4 (1 - tanh d_obj,target); the target0 site stays at its MJCF position unless rendering (Appendix D-8)
    const real gx = obs[0] - obs[3], gy = obs[1] - obs[4], gz = obs[2] - obs[5];
    const real r_reach = (1 - tanh(sqrt(gx * gx + gy * gy + gz * gz))) * 0.2;
    const real r_grasp = E.touch ? 0.5 : 0.0;
    const real tx = obs[3] - P->target0[0], ty = obs[4] - P->target0[1], tz = obs[5] - P->target0[2];
    const real r_lift = E.touch ? 0.5 + (1 - tanh(sqrt(tx * tx + ty * ty + tz * tz))) * (0.9 - 0.5) : 0.0;
    rew_shaped = fmax(fmax(r_reach, r_grasp), r_lift) * 100;
  }
  if (C.hidden) hide_object(obs, ag);
  const int D = C.obs_dim;
  real dx = ag[0] - E.goal[0], dy = ag[1] - E.goal[1], dz = ag[2] - E.goal[2];
  const real dist = sqrt(dx * dx + dy * dy + dz * dz);
  const bool succ = dist < C.distance_threshold;
  real rew = sel(C.reward_type == MCG_REWARD_SPARSE, -(real)(float)(dist > C.distance_threshold), -dist);
  rew = sel(C.reward_type == MCG_REWARD_SHAPING, rew_shaped, rew);
  E.elapsed++; E.eplen++; E.epret += rew;
  const bool term = succ, trunc = succ || (E.elapsed >= C.max_episode_steps);
  if (valid) {                                                      // (a ragged last workgroup's shadow lanes store nothing)
    if (O.reward) O.reward[i] = rew;
    if (O.terminated) O.terminated[i] = term;
    if (O.truncated) O.truncated[i] = trunc;
    if (O.is_success) O.is_success[i] = succ;
    if (O.ep_return) O.ep_return[i] = E.epret;
    if (O.ep_length) O.ep_length[i] = E.eplen;
  }
  const bool done = (term || trunc) && C.auto_reset;
  if (__any(done)) {
    if (done && valid) {
      if (O.final_obs) for (int k = 0; k < 25; k++) if (k < D) O.final_obs[(size_t)i * D + k] = obs[k];
      if (O.final_achieved) for (int k = 0; k < 3; k++) O.final_achieved[(size_t)i * 3 + k] = ag[k];
      if (O.final_desired) for (int k = 0; k < 3; k++) O.final_desired[(size_t)i * 3 + k] = E.goal[k];
    }
    reset_envp(C, i, E, done);
    real obs2[25], ag2[3];
    observe_pnp(C, P, E, obs2, ag2);
    if (C.hidden) hide_object(obs2, ag2);
    for (int k = 0; k < 25; k++) obs[k] = sel(done, obs2[k], obs[k]);
    for (int k = 0; k < 3; k++) ag[k] = sel(done, ag2[k], ag[k]);
  }
  if (valid) { write_obs(O, i, D, obs, ag, E.goal); store_envp(V, i_tail, E); }
  MCG_TICK(ST_POST);
#ifdef MCG_STAGE_CLOCKS
  if (threadIdx.x == 0) g_wg_stat[(blockIdx.x & 4095) * 4] += __builtin_readcyclecounter() - wg_t0;
#endif
  MCG_TICK_FLUSH();
}

__global__ __launch_bounds__(PNP_LANES) void reset_pnp_kernel(Cfg C, View V, const mcg_model* __restrict__ Pg,
                                                              const uint8_t* __restrict__ mask, int reseed, mcg_step_out O) {
  const int i = blockIdx.x * PNP_LANES + threadIdx.x;
  if (i >= C.n) return;
  const ModelPtr P = as_model_ptr(Pg);
  EnvP E;
  load_envp(V, i, E); load_episodep(V, i, E);
  const bool doit = !mask || mask[i];
  E.episode = sel((doit && reseed), 0, E.episode);
  reset_envp(C, i, E, doit);
  store_envp(V, i, E);
  real obs[25], ag[3];
  observe_pnp(C, P, E, obs, ag);
  if (C.hidden) hide_object(obs, ag);
  write_obs(O, i, C.obs_dim, obs, ag, E.goal);
}

// TEST / DEBUG (mcg_debug_contacts): the collision pass of the current state, exported as the step kernels see it.  One wave of 64 lanes
// per 32 environments: lanes 0-31 the lane-parallel part (the cube wave's and the M / RNE waves' shares), all 64 the mesh phase.
__global__ __launch_bounds__(64) void contacts_pnp_kernel(Cfg C, View V, const mcg_model* __restrict__ Pg, const real* __restrict__ poly, int32_t* __restrict__ count,
                                                          int32_t* __restrict__ dropped, double* __restrict__ data) {
  __shared__ real lds[PNP_SLOTS_DUAL][PNP_LANES];
  const int lane = threadIdx.x & (PNP_LANES - 1);
  const bool lower = threadIdx.x < PNP_LANES;
  const int i_raw = blockIdx.x * PNP_LANES + lane;
  const int i = i_raw >= C.n ? C.n - 1 : i_raw;
  const PnpScratch MS(&lds[0][lane]);
  const ModelPtr P = as_model_ptr(Pg);
  const unsigned lds0 = (unsigned)(uintptr_t)(LdsPtr)&lds[0][0];
  EnvP E;
  load_envp(V, i, E);
  CubeSys<PnpScratch> CS(MS, E.Cb, E.dr);
  CS.cnt = nullptr;
  if (lower) CS.template collide_primitives<true, true>(P, E.R.q);
  __syncthreads();
  mesh_phase(P, poly, (LdsPtr)(uintptr_t)lds0, 0, 1);
  __syncthreads();
  if (!lower || i_raw >= C.n) return;
  CS.collect_list();
  count[i] = CS.ncon;
  if (dropped) dropped[i] = CS.ndropped;
  real mult[MAXCON];
  for (int c = 0; c < MAXCON; c++) mult[c] = c < CS.ncon ? MS.ld(LDS_CON + c * CON_STRIDE + CON_D) : 0.0;      // the multiplicities, before the numbers replace them
  for (int r = 0; r < 3; r++) solver_numbers_share(P, MS, E.dr[1], r);
  for (int c = 0; c < MAXCON; c++) {
    const int b = LDS_CON + c * CON_STRIDE;
    double* o = data + ((size_t)i * MAXCON + c) * 10;
    const bool on = c < CS.ncon;
    o[0] = on ? MS.ld(b + CON_DIST) : 0.0;
    for (int k = 0; k < 3; k++) { o[1 + k] = on ? MS.ld(b + k) : 0.0; o[4 + k] = on ? MS.ld(b + 3 + k) : 0.0; }
    o[7] = on ? MS.ld(b + CON_TYPE) : -1.0;
    o[8] = mult[c];
    o[9] = on ? MS.ld(b + CON_D) : 0.0;
  }
}

// compute_reward on batched goals (mycobot.py:289-298) -- the HER entry point
__global__ void reward_kernel(const double* __restrict__ ag, const double* __restrict__ dg, int n, int reward_type,
                              double thr, double* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double dx = ag[3 * i] - dg[3 * i], dy = ag[3 * i + 1] - dg[3 * i + 1], dz = ag[3 * i + 2] - dg[3 * i + 2];
  double d = sqrt(dx * dx + dy * dy + dz * dz);
  out[i] = sel(reward_type == MCG_REWARD_SPARSE, -(double)(float)(d > thr), -d);
}

// state <-> caller arrays (both SoA [dim, N])
__global__ void copy_state_kernel(View V, mcg_state S, int to_engine) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V.n) return;
  const int n = V.n;
#define CP(ptr, acc, cnt) if (S.ptr) for (int k = 0; k < cnt; k++) { if (to_engine) V.acc(k, i) = S.ptr[(size_t)k * n + i]; else S.ptr[(size_t)k * n + i] = V.acc(k, i); }
  CP(qpos, qpos, V.nq) CP(qvel, qvel, V.nv) CP(ctrl, ctrl, 7) CP(warm, warm, V.nv) CP(qpos_lag, qlag, V.nq) CP(goal, goal, 3)
  CP(dr_scale, dr, 2)
#undef CP
  if (S.ep_return) { if (to_engine) V.epret(i) = S.ep_return[i]; else S.ep_return[i] = V.epret(i); }
  if (S.ep_length) { if (to_engine) V.eplen(i) = S.ep_length[i]; else S.ep_length[i] = V.eplen(i); }
  if (S.elapsed) { if (to_engine) V.elapsed(i) = S.elapsed[i]; else S.elapsed[i] = V.elapsed(i); }
  if (S.episode) { if (to_engine) V.episode(i) = S.episode[i]; else S.episode[i] = V.episode(i); }
}

}  // namespace

// ================================================================================================== host ABI
struct mcg_env {
  Cfg cfg;
  View view;
  mcg_model* d_model;
  double* d_poly;                 // the mesh geoms' collision tables (mcg_create: polytopes)
  unsigned long long* d_cnt;      // mcg_counters
  int device;
  int num_cu;
  bool no_split;       // MCG_NO_SPLIT=1 in the environment at mcg_create: always the one-wave REACH kernels (tests, A/B timing)
};

extern "C" {

int mcg_abi_version(void) { return MCG_ABI_VERSION; }
const char* mcg_last_error(void) { return g_err; }

int mcg_default_model(int variant, mcg_model* out) {
  if (!out || variant < 0 || variant >= MCG_NUM_MODEL_VARIANTS) return fail(MCG_ERR_ARG, "mcg_default_model: bad variant%s");
  memcpy(out, &kDefaultModels[variant], sizeof(mcg_model));
  return MCG_OK;
}

int mcg_create(const mcg_config* c, const mcg_model* model, const double* polytopes, int64_t n_polytopes, int device, mcg_env** out) {
  if (!c || !out) return fail(MCG_ERR_ARG, "mcg_create: null argument%s");
  if (polytopes && n_polytopes < 8 * MCG_NMESH) return fail(MCG_ERR_ARG, "mcg_create: polytope block too short%s");
  if (c->n_envs <= 0) return fail(MCG_ERR_ARG, "mcg_create: n_envs must be positive%s");
  if (c->controller != MCG_CTRL_JOINT && c->controller != MCG_CTRL_IK && c->controller != MCG_CTRL_MOCAP) return fail(MCG_ERR_ARG, "mcg_create: controller must be joint, IK or mocap%s");
  {   // the mocap controller needs the model variant with the weld (and without arm actuators), the others the one without
    const mcg_model* mm = model ? model : &kDefaultModels[0];
    if ((c->controller == MCG_CTRL_MOCAP) != (mm->weld_on != 0.0))
      return fail(MCG_ERR_ARG, "mcg_create: the mocap controller goes with the mocap model variants (mcg_default_model 2 / 3), joint and IK with 0 / 1%s");
  }
  if (c->controller == MCG_CTRL_JOINT && c->fetch_env) return fail(MCG_ERR_ARG, "Joint controller not supported for Fetch env%s");  // mycobot.py:96
  if (c->reward_type < MCG_REWARD_SPARSE || c->reward_type > MCG_REWARD_SHAPING) return fail(MCG_ERR_ARG, "mcg_create: bad reward_type%s");
  {
    const mcg_model* mm = model ? model : &kDefaultModels[0];
    bool ok = true;
    for (int k = 0; k < 12; k++) ok = ok && (mm->limit_par[k][6] == 1.0 || mm->limit_par[k][6] == 2.0);
    for (int k = 0; k < 3; k++) ok = ok && (mm->eq_par[k][6] == 1.0 || mm->eq_par[k][6] == 2.0);
    if (mm->weld_on != 0.0) ok = ok && (mm->weld_par[6] == 1.0 || mm->weld_par[6] == 2.0);
    if (!ok) return fail(MCG_ERR_UNSUPPORTED, "mcg_create: solimp power must be 1 or 2 (the MJCF default is 2)%s");
  }
  if (c->frame_skip <= 0 || c->control_steps <= 0 || c->max_episode_steps <= 0) return fail(MCG_ERR_ARG, "mcg_create: frame_skip, control_steps, max_episode_steps must be positive%s");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MCG_ERR_HIP, "mcg_create: no HIP device (this engine has no CPU path)%s");
  if (device < 0 || device >= ndev) return fail(MCG_ERR_ARG, "mcg_create: bad device index%s");
  HIP_OK(hipSetDevice(device));
  mcg_env* e = new (std::nothrow) mcg_env();
  if (!e) return fail(MCG_ERR_ARG, "mcg_create: out of host memory%s");
  const mcg_model* m = model ? model : &kDefaultModels[0];
  Cfg& C = e->cfg;
  // Reach + reward_shaping: stage_rewards reads the cube's site and its pad contacts (mycobot.py:402-448), and the reference only
  // HIDES the cube in Reach (geom and site size := 0, mycobot.py:475-481) -- body, mass and free joint stay.  Those ids run the
  // PickAndPlace kernels on a model whose cube has zero half-size, with Reach's observation, goal and reset.
  C.hidden = (c->reward_type == MCG_REWARD_SHAPING && !c->has_object) ? 1 : 0;
  C.n = c->n_envs; C.has_object = (c->has_object || C.hidden) ? 1 : 0; C.controller = c->controller; C.fetch = c->fetch_env;
  C.reward_type = c->reward_type; C.frame_skip = c->frame_skip; C.control_steps = c->control_steps;
  C.max_episode_steps = c->max_episode_steps; C.target_in_the_air = c->target_in_the_air; C.auto_reset = c->auto_reset;
  C.nq = C.has_object ? 19 : 12; C.nv = C.has_object ? 18 : 12;
  C.obs_dim = c->has_object ? 25 : 10;
  C.act_dim = c->controller == MCG_CTRL_MOCAP ? (c->fetch_env ? 4 : 8)
            : (c->controller == MCG_CTRL_IK && c->fetch_env) ? 4 : 7;                 // mycobot.py:84-103
  C.distance_threshold = c->distance_threshold; C.height_offset = c->height_offset;
  for (int k = 0; k < 3; k++) C.igx[k] = c->initial_gripper_xpos[k];
  C.dt = c->frame_skip * m->timestep;                                                // mycobot.py:346
  C.grip_range = (m->act_ctrlrange[6][1] - m->act_ctrlrange[6][0]) / 2.0;            // mycobot.py:113-115
  C.grip_center = (m->act_ctrlrange[6][1] + m->act_ctrlrange[6][0]) / 2.0;
  memcpy(C.init_qpos, c->init_qpos, sizeof(C.init_qpos));
  memcpy(C.init_qvel, c->init_qvel, sizeof(C.init_qvel));
  memcpy(C.init_ctrl, c->init_ctrl, sizeof(C.init_ctrl));
  C.seed = c->seed; C.env_id_offset = c->env_id_offset;
  C.dr_enable = c->dr_enable && c->has_object;
  C.block_gripper = c->block_gripper;
  for (int k = 0; k < 3; k++) C.qpos0_cube[k] = m->body[12].r[k];      // qpos0 of the free joint = the body's MJCF pose
  C.qpos0_cube[3] = 1; C.qpos0_cube[4] = C.qpos0_cube[5] = C.qpos0_cube[6] = 0;
  C.dr_mass[0] = c->dr_mass_range[0]; C.dr_mass[1] = c->dr_mass_range[1];
  C.dr_fric[0] = c->dr_friction_range[0]; C.dr_fric[1] = c->dr_friction_range[1];
  e->device = device;
  {
    hipDeviceProp_t prop;
    e->num_cu = (hipGetDeviceProperties(&prop, device) == hipSuccess) ? prop.multiProcessorCount : 256;
    const char* ns = getenv("MCG_NO_SPLIT");
    e->no_split = ns && ns[0] == '1';
    // MCG_COOP_PAIR=0: the cooperative phase's first implementation (one environment per wave), kept as a cross-check of the pair solve
    const char* cp = getenv("MCG_COOP_PAIR");
    C.coop_pair = !(cp && cp[0] == '0');
  }
  e->view.n = C.n; e->view.nq = C.nq; e->view.nv = C.nv;
  size_t nd = (size_t)state_doubles(C.nq, C.nv) * C.n;
  hipError_t err = hipMalloc(&e->view.d, nd * sizeof(double));
  if (err == hipSuccess) err = hipMalloc(&e->view.i32, (size_t)3 * C.n * sizeof(int32_t));
  if (err == hipSuccess) err = hipMalloc(&e->d_model, sizeof(mcg_model));
  if (err == hipSuccess) err = hipMalloc(&e->d_cnt, sizeof(mcg_counters));
  if (err == hipSuccess && C.has_object) {
    const double* pb = polytopes ? polytopes : kDefaultPolytopes;
    const size_t np = polytopes ? (size_t)n_polytopes : (size_t)MCG_DEFAULT_POLYTOPES_LEN;
    // the block's own index must stay inside it (a kernel walks these offsets)
    bool ok = true;
    for (int m = 0; m < MCG_NMESH && ok; m++) {
      const double* meta = pb + 8 * m;
      const double end = meta[3] + 3 * meta[4] + 4 * meta[5] + 13 * meta[6];
      ok = meta[0] >= 1 && meta[0] <= meta[4] && meta[1] <= meta[5] && meta[2] <= meta[6] && meta[3] >= 8 * MCG_NMESH && end <= (double)np
           && ((long long)meta[4] % 64) == 0 && ((long long)meta[5] % 64) == 0 && ((long long)meta[6] % 64) == 0
           && meta[4] <= 64 * MESH_VCH && meta[5] <= 64 * MESH_FCH;      // (the narrow phase holds a family's table in registers: csrc/mcg_mesh.hpp)
    }
    if (!ok) { mcg_destroy(e); return fail(MCG_ERR_ARG, "mcg_create: inconsistent polytope block%s"); }
    err = hipMalloc(&e->d_poly, np * sizeof(double));
    if (err == hipSuccess) err = hipMemcpy(e->d_poly, pb, np * sizeof(double), hipMemcpyHostToDevice);
  }
  if (err == hipSuccess) err = hipMemset(e->d_cnt, 0, sizeof(mcg_counters));
  if (err == hipSuccess) C.cnt = e->d_cnt;
  if (err == hipSuccess) err = hipMemset(e->view.d, 0, nd * sizeof(double));
  if (err == hipSuccess) err = hipMemset(e->view.i32, 0, (size_t)3 * C.n * sizeof(int32_t));
  if (err == hipSuccess) {          // domain-randomisation scales start at 1
    std::vector<double> ones((size_t)2 * C.n, 1.0);
    err = hipMemcpy(e->view.d + (size_t)(2 * C.nq + 2 * C.nv + 11) * C.n, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice);
  }
  if (err == hipSuccess) {
    mcg_model mm = *m;
    if (!(mm.contact_rpy > 0.0)) mm.contact_rpy = 2.0;      // a block built before ABI 7 (field zero): the recalled rule
    if (C.hidden) mm.cube_half[0] = mm.cube_half[1] = mm.cube_half[2] = 0.0;       // model.geom_size[object0] = 0
    err = hipMemcpy(e->d_model, &mm, sizeof(mcg_model), hipMemcpyHostToDevice);
  }
  if (err != hipSuccess) { mcg_destroy(e); return fail(MCG_ERR_HIP, "mcg_create: %s", hipGetErrorString(err)); }
  *out = e;
  return MCG_OK;
}

void mcg_destroy(mcg_env* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->view.d) (void)hipFree(e->view.d);
  if (e->view.i32) (void)hipFree(e->view.i32);
  if (e->d_model) (void)hipFree(e->d_model);
  if (e->d_poly) (void)hipFree(e->d_poly);
  if (e->d_cnt) (void)hipFree(e->d_cnt);
  delete e;
}

#ifdef MCG_STAGE_CLOCKS
// development builds only (tools/stage_clocks.py): read and optionally clear the per-stage shader-clock totals
extern "C" int mcg_debug_wg_stat(unsigned long long* out, int n, int clear) {
  if (hipDeviceSynchronize() != hipSuccess) return MCG_ERR_HIP;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(mcg::g_wg_stat), sizeof(unsigned long long) * n) != hipSuccess) return MCG_ERR_HIP;
  if (clear) { static unsigned long long z[4096 * 4]; if (hipMemcpyToSymbol(HIP_SYMBOL(mcg::g_wg_stat), z, sizeof(z)) != hipSuccess) return MCG_ERR_HIP; }
  return MCG_OK;
}
extern "C" int mcg_debug_stage_clocks(unsigned long long* out, int clear) {
  if (hipDeviceSynchronize() != hipSuccess) return MCG_ERR_HIP;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(mcg::g_stage_clocks), sizeof(unsigned long long) * (mcg::ST_COUNT + mcg::CN_COUNT)) != hipSuccess) return MCG_ERR_HIP;
  if (clear) { unsigned long long z[mcg::ST_COUNT + mcg::CN_COUNT] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(mcg::g_stage_clocks), z, sizeof(z)) != hipSuccess) return MCG_ERR_HIP; }
  return MCG_OK;
}
#endif
int mcg_obs_dim(const mcg_env* e) { return e ? e->cfg.obs_dim : -1; }
int mcg_action_dim(const mcg_env* e) { return e ? e->cfg.act_dim : -1; }
int mcg_nq(const mcg_env* e) { return e ? e->cfg.nq : -1; }
int mcg_nv(const mcg_env* e) { return e ? e->cfg.nv : -1; }

static mcg_step_out out_or_empty(const mcg_step_out* o) { mcg_step_out z; memset(&z, 0, sizeof(z)); return o ? *o : z; }

int mcg_reset(mcg_env* e, const uint8_t* mask, int reseed, uint64_t seed, const mcg_step_out* out, void* stream) {
  if (!e) return fail(MCG_ERR_ARG, "mcg_reset: null handle%s");
  if (reseed) e->cfg.seed = seed;
  if (e->cfg.has_object) {
    dim3 grid((e->cfg.n + PNP_LANES - 1) / PNP_LANES), block(PNP_LANES);
    hipLaunchKernelGGL(reset_pnp_kernel, grid, block, 0, (hipStream_t)stream, e->cfg, e->view, e->d_model, mask, reseed, out_or_empty(out));
  } else {
    dim3 grid((e->cfg.n + 63) / 64), block(64);
    hipLaunchKernelGGL(reset_reach_kernel, grid, block, 0, (hipStream_t)stream, e->cfg, e->view, e->d_model, mask, reseed, out_or_empty(out));
  }
  HIP_OK(hipGetLastError());
  return MCG_OK;
}

static int launch_step(mcg_env* e, const float* actions, const mcg_step_out& o, hipStream_t s) {
  if (e->cfg.has_object) {
    // four waves (robot, cube, M, RNE) over 32 environments at every grid size: the 160 KB of LDS allow one workgroup per CU either way
    dim3 grid((e->cfg.n + PNP_LANES - 1) / PNP_LANES);
    const dim3 block(256);
    if (e->cfg.controller == MCG_CTRL_IK) hipLaunchKernelGGL((step_pnp_kernel<MCG_CTRL_IK>), grid, block, 0, s, e->cfg, e->view, e->d_model, e->d_poly, actions, o);
    else if (e->cfg.controller == MCG_CTRL_MOCAP) hipLaunchKernelGGL((step_pnp_kernel<MCG_CTRL_MOCAP>), grid, block, 0, s, e->cfg, e->view, e->d_model, e->d_poly, actions, o);
    else hipLaunchKernelGGL((step_pnp_kernel<MCG_CTRL_JOINT>), grid, block, 0, s, e->cfg, e->view, e->d_model, e->d_poly, actions, o);
    return hipGetLastError() == hipSuccess ? MCG_OK : MCG_ERR_HIP;
  }
  dim3 grid((e->cfg.n + 63) / 64);
  // up to one workgroup per CU three SIMDs of every CU would idle: the two-wave variant puts a helper wave on one of them
  const bool split = (int)grid.x <= e->num_cu && !e->no_split;
  const dim3 block(split ? 192 : 64);
#define MCG_LAUNCH_REACH(CTRL)                                                                                                 \
  do {                                                                                                                         \
    if (split) hipLaunchKernelGGL((step_reach_kernel<CTRL, true>), grid, block, 0, s, e->cfg, e->view, e->d_model, actions, o);     \
    else hipLaunchKernelGGL((step_reach_kernel<CTRL, false>), grid, block, 0, s, e->cfg, e->view, e->d_model, actions, o);          \
  } while (0)
  if (e->cfg.controller == MCG_CTRL_IK) MCG_LAUNCH_REACH(MCG_CTRL_IK);
  else if (e->cfg.controller == MCG_CTRL_MOCAP) MCG_LAUNCH_REACH(MCG_CTRL_MOCAP);
  else MCG_LAUNCH_REACH(MCG_CTRL_JOINT);
#undef MCG_LAUNCH_REACH
  return hipGetLastError() == hipSuccess ? MCG_OK : MCG_ERR_HIP;
}

int mcg_step(mcg_env* e, const float* actions, const mcg_step_out* out, void* stream) {
  if (!e || !actions) return fail(MCG_ERR_ARG, "mcg_step: null argument%s");
  if (launch_step(e, actions, out_or_empty(out), (hipStream_t)stream) != MCG_OK) return fail(MCG_ERR_HIP, "mcg_step: launch failed: %s", hipGetErrorString(hipGetLastError()));
  return MCG_OK;
}

int mcg_time_steps(mcg_env* e, const float* actions, const mcg_step_out* out, int steps, void* stream, float* ms_total) {
  if (!e || !actions || !ms_total || steps <= 0) return fail(MCG_ERR_ARG, "mcg_time_steps: bad argument%s");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t t0, t1;
  HIP_OK(hipEventCreate(&t0)); HIP_OK(hipEventCreate(&t1));
  mcg_step_out o = out_or_empty(out);
  HIP_OK(hipEventRecord(t0, s));
  for (int k = 0; k < steps; k++) if (launch_step(e, actions, o, s) != MCG_OK) return fail(MCG_ERR_HIP, "mcg_time_steps: launch failed%s");
  HIP_OK(hipEventRecord(t1, s));
  HIP_OK(hipEventSynchronize(t1));
  HIP_OK(hipEventElapsedTime(ms_total, t0, t1));
  (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
  return MCG_OK;
}

static int copy_state(mcg_env* e, const mcg_state* s, int to_engine, void* stream) {
  if (!e || !s) return fail(MCG_ERR_ARG, "mcg_get/set_state: null argument%s");
  dim3 grid((e->cfg.n + 255) / 256), block(256);
  hipLaunchKernelGGL(copy_state_kernel, grid, block, 0, (hipStream_t)stream, e->view, *s, to_engine);
  HIP_OK(hipGetLastError());
  return MCG_OK;
}
int mcg_get_state(mcg_env* e, const mcg_state* dst, void* stream) { return copy_state(e, dst, 0, stream); }
uint64_t mcg_get_seed(const mcg_env* e) { return e ? (uint64_t)e->cfg.seed : 0; }
int mcg_set_seed(mcg_env* e, uint64_t seed) { if (!e) return fail(MCG_ERR_ARG, "mcg_set_seed: null handle%s"); e->cfg.seed = seed; return MCG_OK; }
int mcg_set_state(mcg_env* e, const mcg_state* src, void* stream) { return copy_state(e, src, 1, stream); }

#ifdef MCG_COOP_DEBUG
extern "C" int mcg_debug_coop_dump(double* out, int clear) {
  if (out) HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coop_dbg), sizeof(double) * 32 * 512));
  if (clear) { static int z[32]; static double zz[32 * 512]; HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_coop_dbg_done), z, sizeof(z))); HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_coop_dbg), zz, sizeof(zz))); }
  return 0;
}
#endif

int mcg_get_counters(mcg_env* e, mcg_counters* out, int clear) {
  if (!e || !out) return fail(MCG_ERR_ARG, "mcg_get_counters: null argument%s");
  HIP_OK(hipSetDevice(e->device));
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy(out, e->d_cnt, sizeof(mcg_counters), hipMemcpyDeviceToHost));
  if (clear) HIP_OK(hipMemset(e->d_cnt, 0, sizeof(mcg_counters)));
  return MCG_OK;
}

int mcg_debug_contacts(mcg_env* e, int32_t* count, int32_t* dropped, double* data, void* stream) {
  if (!e || !count || !data) return fail(MCG_ERR_ARG, "mcg_debug_contacts: null argument%s");
  if (!e->cfg.has_object) return fail(MCG_ERR_UNSUPPORTED, "mcg_debug_contacts: Reach has no collision pass%s");
  dim3 grid((e->cfg.n + PNP_LANES - 1) / PNP_LANES), block(64);
  hipLaunchKernelGGL(contacts_pnp_kernel, grid, block, 0, (hipStream_t)stream, e->cfg, e->view, e->d_model, e->d_poly, count, dropped, data);
  HIP_OK(hipGetLastError());
  return MCG_OK;
}

int mcg_compute_reward(const double* achieved, const double* desired, int n, int reward_type, double threshold,
                       double* out, void* stream) {
  if (!achieved || !desired || !out || n < 0) return fail(MCG_ERR_ARG, "mcg_compute_reward: bad argument%s");
  if (reward_type != MCG_REWARD_SPARSE && reward_type != MCG_REWARD_DENSE) return fail(MCG_ERR_UNSUPPORTED, "mcg_compute_reward: reward_shaping needs simulator state%s");
  if (n == 0) return MCG_OK;
  hipLaunchKernelGGL(reward_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, achieved, desired, n, reward_type, threshold, out);
  HIP_OK(hipGetLastError());
  return MCG_OK;
}

}  // extern "C"
