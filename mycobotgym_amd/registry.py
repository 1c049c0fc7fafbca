"""Env-ID -> constructor kwargs, reproducing the registration loop of the reference
(/root/reference/mycobotgym/__init__.py:6-45): 30 ``-v0`` ids (state observations) and 20 ``-v1`` ids
(image observations, ``MyCobotImgEnv``), every id wrapped in TimeLimit(max_episode_steps=50).
"""
from __future__ import annotations

import itertools

REWARD = {"dense": "Dense", "sparse": "Sparse", "reward_shaping": "RewardShaping"}
MAX_EPISODE_STEPS = 50


def _build():
    table = {}
    for reward_type, has_object, controller, fetch in itertools.product(
            ["dense", "sparse", "reward_shaping"], [True, False], ["mocap", "IK", "joint"], [True, False]):
        if fetch and controller == "joint":
            continue  # Fetch envs are not supported for the joint controller (__init__.py:21-24)
        kwargs = {
            "model_path": f"./assets/mycobot280{'_mocap' if controller == 'mocap' else ''}.xml",
            "reward_type": reward_type, "has_object": has_object, "controller_type": controller, "fetch_env": fetch,
        }
        name = f"MyCobot{'Fetch' if fetch else ''}{'PickAndPlace' if has_object else 'Reach'}"
        table[f"{name}-{REWARD[reward_type]}-{controller}-v0"] = dict(kwargs, image_obs=False)
        if reward_type != "reward_shaping":   # no image variant for reward shaping (__init__.py:37-39)
            table[f"{name}-{REWARD[reward_type]}-{controller}-v1"] = dict(kwargs, image_obs=True)
    return table


REGISTRY = _build()


def spec(env_id: str) -> dict:
    if env_id not in REGISTRY:
        raise KeyError(f"unknown env id {env_id!r}; registered ids: MyCobot[Fetch]{{Reach,PickAndPlace}}-"
                       "{Dense,Sparse,RewardShaping}-{mocap,IK,joint}-v{0,1}")
    return dict(REGISTRY[env_id])
