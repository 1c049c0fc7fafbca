"""Stable-Baselines3 ``VecEnv`` protocol over ``MyCobotVecEnv``.

The reference trains through SB3's ``DummyVecEnv`` / ``SubprocVecEnv`` around ``Monitor(gymnasium.make(id))``
(/root/reference/mycobotgym/scripts/train.py:25-33,80-85) and evaluates with ``evaluate_policy`` reading
``info["is_success"]`` (scripts/eval_model.py:131).  This adapter exposes the same surface [RECALL SB3 2.0 VecEnv]:
numpy observations ``dict[str, ndarray[N, ...]]``, ``rewards float32[N]``, ``dones bool[N]``, ``infos: list[dict]`` with
``terminal_observation``, ``TimeLimit.truncated``, ``episode = {"r", "l", "t"}`` (what ``Monitor`` adds) and
``is_success``; ``env_method("compute_reward", ...)`` for HER.  One device->host copy per step.

stable_baselines3 is not installed in the build image, so the class is duck-typed; when SB3 is importable it also
registers as a virtual subclass of ``stable_baselines3.common.vec_env.VecEnv``.
"""
from __future__ import annotations

import time
from typing import Any, List, Optional, Sequence

import numpy as np
import torch


class MyCobotSB3VecEnv:
    def __init__(self, envs):
        self.envs = envs
        self.num_envs = envs.num_envs
        self.observation_space = _maybe_gym(envs.single_observation_space)
        self.action_space = _maybe_gym(envs.single_action_space)
        self.render_mode = None
        self._actions = None
        self._t0 = time.time()
        self._seed: Optional[int] = None

    # ------------------------------------------------------------------------------------------------ VecEnv API
    def seed(self, seed: Optional[int] = None) -> List[Optional[int]]:
        self._seed = seed
        return [None if seed is None else seed + i for i in range(self.num_envs)]      # train.py:32 uses seed + rank

    def reset(self):
        obs, _ = self.envs.reset(seed=self._seed)
        self._seed = None
        return _to_numpy(obs)

    def step_async(self, actions) -> None:
        self._actions = np.asarray(actions, dtype=np.float32)

    def step_wait(self):
        obs, rew, term, trunc, info = self.envs.step(self._actions)
        obs_np = _to_numpy(obs)
        rew_np = rew.detach().cpu().numpy().astype(np.float32)
        term_np = term.cpu().numpy(); trunc_np = trunc.cpu().numpy()
        dones = term_np | trunc_np
        succ = info["is_success"].cpu().numpy()
        infos: List[dict] = [{"is_success": bool(succ[i])} for i in range(self.num_envs)]
        if dones.any():
            final = _to_numpy(info["final_observation"])
            ep_r = info["episode"]["r"].cpu().numpy(); ep_l = info["episode"]["l"].cpu().numpy()
            for i in np.nonzero(dones)[0]:
                infos[i]["terminal_observation"] = {k: v[i] for k, v in final.items()}
                infos[i]["TimeLimit.truncated"] = bool(trunc_np[i] and not term_np[i])
                infos[i]["episode"] = {"r": float(ep_r[i]), "l": int(ep_l[i]), "t": round(time.time() - self._t0, 6)}
        return obs_np, rew_np, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self) -> None:
        self.envs.close()

    def get_attr(self, attr_name: str, indices=None) -> List[Any]:
        return [getattr(self.envs, attr_name) for _ in self._indices(indices)]

    def set_attr(self, attr_name: str, value: Any, indices=None) -> None:
        setattr(self.envs, attr_name, value)

    def env_method(self, method_name: str, *method_args, indices=None, **method_kwargs) -> List[Any]:
        """HER calls ``env_method("compute_reward", achieved, desired, infos, indices=[0])`` with batched goals."""
        out = getattr(self.envs, method_name)(*method_args, **method_kwargs)
        if isinstance(out, torch.Tensor):
            out = out.detach().cpu().numpy()
        return [out for _ in self._indices(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None) -> List[bool]:
        name = getattr(wrapper_class, "__name__", "")
        return [name in ("Monitor", "TimeLimit") for _ in self._indices(indices)]       # both are built into the engine

    def get_images(self) -> Sequence[Optional[np.ndarray]]:
        return [None] * self.num_envs

    def render(self, mode: Optional[str] = None):
        return None

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        return [indices] if isinstance(indices, int) else indices

    @property
    def unwrapped(self):
        return self


def _to_numpy(obs: dict) -> dict:
    return {k: v.detach().cpu().numpy() for k, v in obs.items()}


def _maybe_gym(space):
    try:
        return space.to_gymnasium()
    except Exception:       # gymnasium not installed
        return space


try:        # pragma: no cover - SB3 is absent in the build image
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv
    _SB3VecEnv.register(MyCobotSB3VecEnv) if hasattr(_SB3VecEnv, "register") else None
except Exception:
    pass
