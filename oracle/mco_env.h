/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of the reference's env layer on top of mco_physics:
 *   MyCobotEnv.step            /root/reference/mycobotgym/envs/mycobot.py:132-205   (mocap branch :172-189, with
 *                              gymnasium_robotics' mocap_set_action / reset_mocap2body_xpos restated [RECALL])
 *   reset / reset_model        :506-514, :207-236      _sample_goal :238-243
 *   _get_obs / generate_mujoco_observations  :245-283, :342-388
 *   _is_success / compute_reward / compute_terminated / compute_truncated  :285-298, :390-400
 *   IKController.compute_qpos_delta / solve_DLS   /root/reference/mycobotgym/utils.py:499-556
 *   goal_distance, generate_random_point_inside_rectangle   utils.py:14-26
 *   TimeLimit(max_episode_steps=50)   /root/reference/mycobotgym/__init__.py:34
 * plus the vectorised auto-reset contract of gymnasium 0.28 VectorEnv [RECALL].
 *
 * PARITY UNPINNED against MuJoCo/gymnasium (absent here); see mco_physics.h.
 * The reference's reset randomness comes from Python's global `random` and numpy's Generator
 *  and is not reproducible from a seed; this restatement draws the same
 * distributions from Philox4x32-10 keyed by (seed, global env id, episode, draw).
 */
#ifndef MCO_ENV_H
#define MCO_ENV_H

#include <stdint.h>
#include "mco_physics.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { MCO_CTRL_JOINT = 0, MCO_CTRL_IK = 1, MCO_CTRL_MOCAP = 2 };
enum { MCO_REWARD_SPARSE = 0, MCO_REWARD_DENSE = 1, MCO_REWARD_SHAPING = 2 };

typedef struct mco_env_config {
  int32_t n_envs, has_object, controller, fetch_env, reward_type;
  int32_t frame_skip, control_steps, max_episode_steps, target_in_the_air, auto_reset;
  int32_t eef_site, obj_site, obj_jnt, grip_jnt[2], n_threads;
  int32_t dr_enable, pad_geom[2], obj_geom;
  int32_t block_gripper, finger_jnt[2];      /* _step_callback: zero the two finger joints after every step (mycobot.py:300-306) */
  int32_t tcp_body, pad_;                    /* mocap controller: the body welded to the mocap body (gripper_tcp) */
  double distance_threshold, height_offset;
  double init_qpos[MCO_MAXNQ], init_qvel[MCO_MAXNV], init_ctrl[MCO_MAXU];
  double dr_mass_range[2], dr_friction_range[2];
  double init_mocap[7];                      /* mocap pose at construction: body pose, or the keyframe's mpos / mquat (fetch) */
  uint64_t seed;
  int64_t env_id_offset;
} mco_env_config;

typedef struct mco_envs mco_envs;

int mco_env_config_sizeof(void);
mco_envs* mco_envs_create(const mco_model* model, const mco_env_config* cfg);
void mco_envs_destroy(mco_envs* e);
int mco_envs_obs_dim(const mco_envs* e);
int mco_envs_action_dim(const mco_envs* e);
void mco_envs_initial_gripper_xpos(const mco_envs* e, double out[3]);

/* mask NULL = all.  reseed != 0: use `seed` as the new base seed and restart episode counters. */
void mco_envs_reset(mco_envs* e, const uint8_t* mask, int reseed, uint64_t seed,
                    double* obs, double* achieved, double* desired);
void mco_envs_step(mco_envs* e, const float* actions, double* obs, double* achieved, double* desired,
                   double* reward, uint8_t* terminated, uint8_t* truncated, uint8_t* is_success,
                   double* final_obs, double* final_achieved, double* final_desired,
                   double* ep_return, int32_t* ep_length);
/* row-major [n_envs, dim] arrays; qpos_lag = qpos of the last forward pass */
void mco_envs_get_state(const mco_envs* e, double* qpos, double* qvel, double* ctrl, double* warm,
                        double* qpos_lag, double* goal, int32_t* elapsed, int32_t* episode);
void mco_envs_set_state(mco_envs* e, const double* qpos, const double* qvel, const double* ctrl,
                        const double* warm, const double* qpos_lag, const double* goal,
                        const int32_t* elapsed, const int32_t* episode);
mco_data* mco_envs_data(mco_envs* e, int i);
void mco_compute_reward(const double* achieved, const double* desired, int n, int reward_type,
                        double threshold, double* out);
void mco_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
