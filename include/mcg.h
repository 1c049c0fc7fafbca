/*
 * mcg.h -- C ABI of the MI355X rollout engine for the MyCobotGym step()/reset() hot path.
 *
 * The reference has no FFI of its own: its hot path is MuJoCo driven from Python through the
 * Gymnasium Env protocol.  Each entry point below replaces one reference interface, batched over
 * N environments (citations relative to /root/reference):
 *
 *   mcg_create          MyCobotEnv.__init__ + _env_setup         mycobotgym/envs/mycobot.py:30-115, 450-481
 *                       (MuJoCo model compile is replaced by the precompiled mcg_model block)
 *   mcg_reset           MyCobotEnv.reset / reset_model / _sample_goal   mycobot.py:506-514, 207-243
 *   mcg_step            MyCobotEnv.step (controller branch joint | IK | mocap -> mujoco.mj_step x frame_skip -> _get_obs ->
 *                       _is_success / compute_reward / compute_terminated / compute_truncated)
 *                       mycobot.py:132-205, 245-298, 342-400; IKController mycobotgym/utils.py:499-556;
 *                       TimeLimit(50) from the registration mycobotgym/__init__.py:34; auto-reset as in
 *                       gymnasium.vector (info["final_observation"]) / SB3 VecEnv used by scripts/train.py:80-85
 *   mcg_compute_reward  MyCobotEnv.compute_reward on batched goals (HER)   mycobot.py:289-298, utils.py:24-26
 *   mcg_get_state / mcg_set_state   direct access to data.qpos/qvel/ctrl/qacc_warmstart (set_joint_qpos etc.)
 *
 * Conventions: every pointer in the step/reset/state calls is DEVICE memory owned by the caller;
 * the engine owns its struct-of-arrays state.  Calls enqueue work on `stream` (a hipStream_t passed
 * as void*, NULL = default stream) and return without synchronising.  Return 0 = OK, otherwise an
 * MCG_ERR_* code with text in mcg_last_error().  A handle is bound to one device and is not
 * thread-safe; distinct handles are independent.  There is no CPU fallback: without a HIP device
 * mcg_create fails.
 */
#ifndef MCG_H
#define MCG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCG_ABI_VERSION 8
#define MCG_MAXCON 16          /* entries of an environment's contact list (MuJoCo has no cap; see mcg_counters.contacts_dropped) */
#define MCG_NMESH 14           /* collision polytopes of the mesh geoms */

enum { MCG_OK = 0, MCG_ERR_ARG = 1, MCG_ERR_HIP = 2, MCG_ERR_UNSUPPORTED = 3 };
enum { MCG_CTRL_JOINT = 0, MCG_CTRL_IK = 1, MCG_CTRL_MOCAP = 2 };   /* controller_type "joint" | "IK" | "mocap" */
enum { MCG_REWARD_SPARSE = 0, MCG_REWARD_DENSE = 1, MCG_REWARD_SHAPING = 2 };

/* Numeric model block (produced by mycobotgym_amd/model/specialize.py from the compiled MJCF).
   13 bodies: link1..6, right gear/finger, left gear/finger, right/left hinge, cube. */
typedef struct mcg_body {          /* 16 doubles = 128 B: one body's constants, contiguous for wide scalar loads */
  double r[3];                      /* body origin in its parent's frame */
  double mass, mc[3];               /* mass, mass * centre of mass (body frame, about the origin) */
  double inertia[6];                /* xx yy zz xy xz yz about the body origin */
  double armature, damping;         /* of the body's hinge (cube: unused, see cube_damping) */
  double hull_rad;                  /* robot bodies: largest distance from the body origin to a vertex of the mesh polytopes riding on it */
} mcg_body;

typedef struct mcg_model {
  double timestep;
  double base_pos[3], base_mat[9], gravity_base[3];
  mcg_body body[13];
  double cube_damping[6];
  double jnt_range[12][2];
  double limit_par[12][10];         /* K B d0 dmax width midpoint power 1/width 1/mid^(p-1) 1/(1-mid)^(p-1); refsafe applied */
  double limit_diag[12];            /* dof_invweight0 */
  double eq_anchor1[2][3], eq_anchor2[2][3];
  double eq_par[3][10], eq_diag[3];  /* connect right, connect left, joint coupling */
  double act_gain[7], act_bias[7][3], act_ctrlrange[7][2], act_forcerange[7][2], tendon_coef[2];
  double site_eef[3];               /* EEF site in the link6 frame */
  /* PickAndPlace only */
  double cube_half[3], table_pos[3], table_half[3], pad_box[2][6];
  double contact_par[7][15];        /* table-cube, right pad-cube, left pad-cube, table-right pad, table-left pad, table-mesh (condim 3),
                                       mesh-cube: the 10 solver numbers | friction[5].  The ground plane carries the table's
                                       (default) parameters; all mesh geoms are default geoms (asserted when the block is made). */
  double contact_diag[5][2];        /* summed body_invweight0 (translational, rotational) of the first five pairs */
  /* Convex-mesh collision (mycobot280_main.xml:105-247): the fourteen mesh geoms of the arm and the gripper -- link1..link6,
     flange, gripper_base, right gear / finger link, left gear / finger link, right / left hinge link -- against the ground plane, the
     table and the cube, on each mesh's collision polytope (within 1 mm of its convex hull; mycobotgym_amd/model/polytope.py), by an
     exact separating-axis test, one contact per pair (csrc/mcg_mesh.hpp).  The polytopes' vertex / face / edge tables are a separate
     block (mcg_create: `polytopes`). */
  double mesh_box[MCG_NMESH][6];    /* centre and half extents of the polytope's bounding box in the frame of the robot body it rides on
                                       (polytope m on body m for m < 6; flange, gripper_base on 5; then bodies 6..11): broad phase */
  double mesh_mult;                 /* identical colliding geoms per mesh (the reference attaches every mesh twice: 2) */
  double mesh_fric;                 /* the mesh geoms' own sliding friction (the cube's is re-scaled under domain randomisation, a pair
                                       takes the larger) */
  double pair_tran[5 + 2 * MCG_NMESH];   /* per pair type (csrc/mcg_cube.hpp: 0 static-cube, 1 / 2 pad-cube, 3 / 4 static-pad, 5 + m static-mesh m,
                                       19 + m mesh m-cube) the summed translational body_invweight0 of the two geoms' bodies */
  double geom_friction0[3];         /* sliding friction of the table, pad and cube geoms (re-mixed under domain randomisation) */
  /* mocap variant only (mycobot280_mocap.xml): weld between the mocap body and gripper_tcp */
  double base_quat[4];              /* orientation of the arm's base body: start of the xquat chain */
  double weld_on;                   /* 1 = the model carries the weld (and no arm actuators) */
  double weld_par[10];              /* solver numbers as in limit_par */
  double weld_diag[2];              /* diagApprox of the translational rows 0-2 and of the rotational rows 3-5.  Built-in models: the
                                       same (translational) weight on all six, which is what the reference's mocap keyframe supports
                                       (oracle/RULE_STUDY.md, K2); mj_diagApprox as recalled puts the rotational inverse weight on
                                       rows 3-5: specialize(..., weld_rule="mujoco") / MyCobotVecEnv(weld_rule="mujoco") */
  double weld_anchor[3];            /* weld point on the robot, link6 frame */
  double weld_relpos[3], weld_relquat[4], weld_torquescale;
  double target0[3];                /* MJCF position of site `target0`: what stage_rewards reads unless rendering (mycobot.py:422, 309-311) */
  double contact_rpy;               /* regulariser of a contact's pyramid rows, Rpy = contact_rpy * mu^2 * R(normal row).  2 = the rule as
                                       recalled from MuJoCo (built-in models); 4 = the one single change that reproduces the cube's rest
                                       height in the reference's keyframes (penetration 1.9e-5, mycobot280.xml:6; oracle/RULE_STUDY.md K1):
                                       specialize(..., contact_rule="keyframe") / MyCobotVecEnv(contact_rule="keyframe") */
} mcg_model;

typedef struct mcg_config {
  int32_t n_envs;
  int32_t has_object;          /* 0 = Reach, 1 = PickAndPlace                       (mycobot.py:33).  Reach with reward_type =
                                  MCG_REWARD_SHAPING keeps the cube as a hidden free body (geom and site size zero,
                                  mycobot.py:475-481): stage_rewards reads its site and contacts (mycobot.py:402-448) */
  int32_t controller;          /* MCG_CTRL_*                                       (mycobot.py:36) */
  int32_t fetch_env;           /*                                                  (mycobot.py:41) */
  int32_t reward_type;         /* MCG_REWARD_*                                     (mycobot.py:42) */
  int32_t frame_skip;          /* 20                                               (mycobot.py:43) */
  int32_t control_steps;       /* 5, IK only                                       (mycobot.py:35) */
  int32_t max_episode_steps;   /* 50, TimeLimit                                    (__init__.py:34) */
  int32_t target_in_the_air;   /*                                                  (mycobot.py:38) */
  int32_t auto_reset;          /* reset finished envs inside mcg_step */
  int32_t dr_enable;           /* per-reset domain randomisation (build-defined) */
  int32_t block_gripper;       /* zero the two finger joints after every step      (mycobot.py:34,300-306) */
  double distance_threshold;   /* 0.01                                             (mycobot.py:39) */
  double height_offset;        /* z of site object0 at the initial state           (mycobot.py:470-472) */
  double initial_gripper_xpos[3];   /* EEF site at the initial state               (mycobot.py:464-466) */
  double init_qpos[19], init_qvel[18], init_ctrl[7];   /* snapshot restored by reset (mycobot.py:80-82) */
  double dr_mass_range[2], dr_friction_range[2];
  uint64_t seed;
  int64_t env_id_offset;       /* global id of env 0 of this handle: RNG streams are keyed by global id */
} mcg_config;

/* Output block of mcg_step / mcg_reset; all device pointers, any may be NULL to skip.  D = mcg_obs_dim. */
typedef struct mcg_step_out {
  double* obs;            /* [N, D]  "observation" (after auto-reset where done) */
  double* achieved_goal;  /* [N, 3] */
  double* desired_goal;   /* [N, 3] */
  double* reward;         /* [N]     dense: -d (f64); sparse: -(d > thr) as in the reference's float32 */
  uint8_t* terminated;    /* [N] */
  uint8_t* truncated;     /* [N]     is_success | elapsed >= max_episode_steps */
  uint8_t* is_success;    /* [N] */
  double* final_obs;      /* [N, D]  pre-reset observation, valid where terminated|truncated */
  double* final_achieved; /* [N, 3] */
  double* final_desired;  /* [N, 3] */
  double* ep_return;      /* [N]     running episode return (Monitor's "r" where done) */
  int32_t* ep_length;     /* [N]     running episode length (Monitor's "l" where done) */
} mcg_step_out;

/* State arrays are struct-of-arrays [dim, N] (N fastest), the engine's native layout. */
typedef struct mcg_state {
  double* qpos;      /* [nq, N]   nq = 12 (Reach) | 19 (PickAndPlace) */
  double* qvel;      /* [nv, N]   nv = 12 | 18 */
  double* ctrl;      /* [7, N] */
  double* warm;      /* [nv, N]   qacc_warmstart */
  double* qpos_lag;  /* [nq, N]   qpos of the last forward pass: observations lag one sub-step (SURVEY D-1) */
  double* goal;      /* [3, N] */
  int32_t* elapsed;  /* [N] */
  int32_t* episode;  /* [N]       per-env episode counter (RNG stream position) */
  double* dr_scale;  /* [2, N]    domain-randomisation scales of the current episode: cube mass, sliding friction */
  double* ep_return; /* [N]       running return of the episode in flight (Monitor's "r" when it ends) */
  int32_t* ep_length;/* [N]       running length of the episode in flight (Monitor's "l") */
} mcg_state;

/* Event counters since mcg_create (or the last clearing read).  The first two are this engine's own bounds on things the reference
   leaves unbounded; none of them is expected to move in ordinary use. */
typedef struct mcg_counters {
  uint64_t reset_cap_hits;          /* a rejection loop of reset_model (mycobot.py:218-219, 232-233: unbounded `while`) gave up after 1000 draws */
  uint64_t bad_state_resets;        /* mj_checkPos / mj_checkVel / mj_checkAcc fired: a body's state was reset (per body, per event) */
  uint64_t contacts_dropped;        /* contacts cut off by the build's cap of MCG_MAXCON list entries per environment (MuJoCo has no such cap) */
  uint64_t coupled_env_substeps;    /* environment-sub-steps routed through the cooperative robot + cube solve (a contact reached the robot) */
} mcg_counters;

typedef struct mcg_env mcg_env;

int mcg_abi_version(void);
const char* mcg_last_error(void);
/* built-in model blocks: 0 = legacy mesh inertia (default), 1 = exact mesh inertia; 2, 3 = the same for the mocap variant */
int mcg_default_model(int variant, mcg_model* out);

/* polytopes: the mesh geoms' collision tables in the robot bodies' frames (layout: mycobotgym_amd/model/polytope.py: pack), n_polytopes
   doubles, host memory; NULL = the built-in tables of the reference's meshes */
int mcg_create(const mcg_config* cfg, const mcg_model* model /* NULL = variant 0 */, const double* polytopes, int64_t n_polytopes,
               int device, mcg_env** out);
void mcg_destroy(mcg_env* env);
int mcg_obs_dim(const mcg_env* env);
int mcg_action_dim(const mcg_env* env);
int mcg_nq(const mcg_env* env);
int mcg_nv(const mcg_env* env);

int mcg_reset(mcg_env* env, const uint8_t* mask /* [N] device or NULL = all */, int reseed, uint64_t seed,
              const mcg_step_out* out, void* stream);
int mcg_step(mcg_env* env, const float* actions /* [N, A] row-major, device */, const mcg_step_out* out, void* stream);
int mcg_get_state(mcg_env* env, const mcg_state* dst, void* stream);
int mcg_set_state(mcg_env* env, const mcg_state* src, void* stream);
/* base seed of the reset / domain-randomisation streams (changed by mcg_reset with reseed != 0): part of a checkpoint */
uint64_t mcg_get_seed(const mcg_env* env);
int mcg_set_seed(mcg_env* env, uint64_t seed);
int mcg_compute_reward(const double* achieved /* [n,3] device */, const double* desired, int n, int reward_type,
                       double threshold, double* out, void* stream);

/* Synchronises the device; copies the counters to host memory `out`; clears them when `clear` != 0. */
int mcg_get_counters(mcg_env* env, mcg_counters* out, int clear);

/* TEST / DEBUG: the collision pass (mj_collision restated, P4) of PickAndPlace on the CURRENT state, exported as the kernels see it.
   All device pointers.  count [N]: list entries; dropped [N]: contacts cut by the cap (or NULL); data [N, MCG_MAXCON, 10]: per entry dist,
   pos[3] (midpoint between the surfaces), normal[3] (geom1 -> geom2), pair type (csrc/mcg_cube.hpp PAIR_*), multiplicity (identical
   geoms the entry stands for), D (weight of its pyramid rows). */
int mcg_debug_contacts(mcg_env* env, int32_t* count, int32_t* dropped, double* data, void* stream);

/* Live timing of the step kernel on its own stream with HIP events (used by bench.py's roofline leg). */
int mcg_time_steps(mcg_env* env, const float* actions, const mcg_step_out* out, int steps, void* stream, float* ms_total);

#ifdef __cplusplus
}
#endif
#endif
