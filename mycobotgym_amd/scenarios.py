"""Scripted initial states for benchmarks and demos (host-side numpy only)."""
from __future__ import annotations

import numpy as np


def grasp_state(n: int, mesh_inertia: str = "legacy", jitter: float = 0.002, seed: int = 0) -> dict:
    """SoA state ([dim, N]) with the arm at the reference's fetch keyframe (mycobot280.xml:6), the cube between the
    open finger pads and the gripper commanded shut -- the 'close gripper over cube' variant of SURVEY 8(d) config 3.
    Feed it to ``MyCobotVecEnv.set_state`` and step with the keyframe's ctrl as joint targets and gripper action 1."""
    from .model.refdyn import kinematics
    from .vec_env import load_table
    tab = load_table(True, mesh_inertia)
    key = tab["keys"][0]
    q = np.asarray(key["qpos"], dtype=np.float64)
    kin = kinematics(tab, q)
    gn = tab["geom_name"]
    mid = 0.5 * (kin["geom_xpos"][gn.index("right_finger_layer")] + kin["geom_xpos"][gn.index("left_finger_layer")])
    rng = np.random.default_rng(seed)
    qpos = np.tile(q[:, None], (1, n))
    qpos[12:15] = mid[:, None] + rng.normal(size=(3, n)) * jitter
    quat = np.array([1.0, 0, 0, 0])[:, None] + rng.normal(size=(4, n)) * 0.05
    qpos[15:19] = quat / np.linalg.norm(quat, axis=0, keepdims=True)
    ctrl = np.tile(np.asarray(key["ctrl"], dtype=np.float64)[:, None], (1, n)); ctrl[6] = 1.0
    action = np.clip(np.concatenate([np.asarray(key["ctrl"][:6]), [1.0]]), -1, 1).astype(np.float32)
    return {"qpos": qpos, "qvel": np.zeros((18, n)), "ctrl": ctrl, "warm": np.zeros((18, n)), "qpos_lag": qpos.copy(),
            "action": np.tile(action[None, :], (n, 1))}
