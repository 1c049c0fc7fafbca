"""mycobotgym_amd -- MI355X-native vectorised rollout engine for the MyCobotGym step()/reset() hot path.

    from mycobotgym_amd import make
    envs = make("MyCobotReach-Dense-joint-v0", num_envs=8192, device="cuda:0")
    obs, info = envs.reset(seed=0)
    obs, reward, terminated, truncated, info = envs.step(actions)     # torch tensors on the GPU

The numeric path is the HIP library ``libmycobot_hip.so`` (C ABI in ``include/mcg.h``); importing the
package does not need a GPU, constructing an environment does.
"""
from .registry import REGISTRY, spec  # noqa: F401
from .vec_env import MyCobotVecEnv, make  # noqa: F401

__version__ = "0.1.0"
