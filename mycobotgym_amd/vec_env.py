"""``MyCobotVecEnv`` -- N MyCobot environments stepped in lockstep on one MI355X.

Host-side mirror of the reference's ``MyCobotEnv`` (/root/reference/mycobotgym/envs/mycobot.py:27-514):
same constructor keywords, observation Dict {observation, achieved_goal, desired_goal} (float64), Box(-1,1)
float32 actions, ``step -> (obs, reward, terminated, truncated, info)``, ``compute_reward`` on batched goals,
wrapped in the registration's TimeLimit(50) (mycobotgym/__init__.py:34) and batched with Gymnasium
``VectorEnv`` semantics (auto-reset; ``info["final_observation"]`` / ``info["_final_observation"]``).

Everything numeric happens in the HIP library behind ``include/mcg.h``; this file only owns device buffers
(PyTorch tensors) and hands their addresses across the C ABI.  No CPU fallback: without the library or without
a GPU the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Optional

import numpy as np
import torch

from . import _abi
from .registry import MAX_EPISODE_STEPS, spec as _spec
from .spaces import Box, Dict, batch_box

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
_CONTROLLERS = {"joint": _abi.CTRL_JOINT, "IK": _abi.CTRL_IK, "mocap": _abi.CTRL_MOCAP}
_REWARDS = {"sparse": _abi.REWARD_SPARSE, "dense": _abi.REWARD_DENSE, "reward_shaping": _abi.REWARD_SHAPING}


def load_table(has_object: bool, mesh_inertia: str = "legacy", mocap: bool = False) -> dict:
    """Compiled model table: mycobot280.xml (joint / IK) or mycobot280_mocap.xml (mocap controller), mycobot.py:47-58."""
    name = ("mycobot280" + ("_mocap" if mocap else "") + ("" if has_object else "_reach")
            + ("_exactmesh" if mesh_inertia == "exact" else ""))
    from .model.mjcf import load_model
    return load_model(os.path.join(_ASSETS, name + ".json"))


def initial_state(has_object: bool, fetch_env: bool, mesh_inertia: str = "legacy", mocap: bool = False):
    """(init_qpos, init_qvel, init_ctrl, initial_gripper_xpos, height_offset): what ``_env_setup`` and the
    constructor snapshot (mycobot.py:78-82, 450-472).  Non-fetch: qpos0 / zero ctrl; fetch: keyframe 0.
    ``init_ctrl`` always has the engine's 7 slots; the mocap model's single (finger) actuator is slot 6."""
    from .model.specialize import initial_gripper_xpos
    full = load_table(True, mesh_inertia, mocap)
    tab = full if has_object else load_table(False, mesh_inertia, mocap)
    nq, nv = tab["nq"], tab["nv"]
    if fetch_env:
        key = full["keys"][0]
        qpos = np.asarray(key["qpos"], dtype=np.float64)[:nq].copy()
        qvel = np.asarray(key["qvel"], dtype=np.float64)[:nv].copy()
        ctrl = np.zeros(7); kc = np.asarray(key["ctrl"], dtype=np.float64)
        ctrl[7 - len(kc):] = kc
        height = float(key["qpos"][14])            # z of site object0 after mj_resetDataKeyframe + mj_forward
    else:
        qpos = np.asarray(tab["qpos0"], dtype=np.float64).copy()
        qvel = np.zeros(nv); ctrl = np.zeros(7)
        height = float(full["qpos0"][14])
    igx = initial_gripper_xpos(tab, qpos)
    return qpos, qvel, ctrl, igx, height


class MyCobotVecEnv:
    metadata = {"render_modes": [], "render_fps": 25}      # mycobot.py:28

    def __init__(self, num_envs: int, has_object: bool = True, block_gripper: bool = False, control_steps: int = 5,
                 controller_type: str = "IK", obj_range: float = 0.1, target_in_the_air: bool = True,
                 distance_threshold: float = 0.01, fetch_env: bool = False, reward_type: str = "sparse",
                 frame_skip: int = 20, max_episode_steps: int = MAX_EPISODE_STEPS, device="cuda:0", seed: int = 0,
                 env_id_offset: int = 0, auto_reset: bool = True, mesh_inertia: str = "legacy",
                 domain_randomization: Optional[dict] = None, model_path: Optional[str] = None,
                 image_obs: bool = False, model: Optional["_abi.McgModel"] = None, weld_rule: str = "common", contact_rule: str = "mujoco",
                 **unused):
        if image_obs:
            raise NotImplementedError("image observations (-v1 ids, MyCobotImgEnv) need a rasteriser: out of scope")
        if controller_type == "delta_joint":
            raise NotImplementedError("delta_joint has no branch in the reference's step() (SURVEY D-10)")
        if controller_type not in _CONTROLLERS:
            raise ValueError(f"controller_type must be one of mocap, IK, joint, delta_joint; got {controller_type!r}")
        if reward_type not in _REWARDS:
            raise ValueError(f"unknown reward_type {reward_type!r}")
        if controller_type == "joint" and fetch_env:
            raise AssertionError("Joint controller not supported for Fetch env")        # mycobot.py:96
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _abi.McgError("MyCobotVecEnv runs on an AMD GPU only (device='cuda:N'); there is no CPU path")
        self._lib = _abi.load()
        if not torch.cuda.is_available():
            raise _abi.McgError("no GPU visible to PyTorch-ROCm; MyCobotVecEnv has no CPU path")
        if model_path is not None:          # the registry's kwarg (mycobotgym/__init__.py:12): only the built-in, precompiled assets exist here
            want = f"mycobot280{'_mocap' if controller_type == 'mocap' else ''}.xml"
            if os.path.basename(str(model_path)) != want:
                raise ValueError(f"model_path {model_path!r}: this engine runs the precompiled {want} only (compile another MJCF with "
                                 "tools/compile_model.py and pass it as model=McgModel.from_spec(...))")
        self.num_envs = int(num_envs)
        self.has_object, self.fetch_env = bool(has_object), bool(fetch_env)
        # Reach + reward_shaping keeps the (hidden) cube in the physics: stage_rewards reads it (mycobot.py:402-448, 475-481)
        self.hidden_object = (not self.has_object) and reward_type == "reward_shaping"
        self.controller_type, self.reward_type = controller_type, reward_type
        self.distance_threshold = float(distance_threshold)
        self.frame_skip, self.control_steps = int(frame_skip), int(control_steps)
        self.max_episode_steps = int(max_episode_steps)
        self.obj_range = obj_range

        mocap = controller_type == "mocap"
        qpos, qvel, ctrl, igx, height = initial_state(self.has_object or self.hidden_object, self.fetch_env, mesh_inertia, mocap)
        self.initial_gripper_xpos, self.height_offset = igx, height
        cfg = _abi.McgConfig()
        cfg.n_envs = self.num_envs; cfg.has_object = int(self.has_object)
        cfg.controller = _CONTROLLERS[controller_type]; cfg.fetch_env = int(self.fetch_env)
        cfg.reward_type = _REWARDS[reward_type]; cfg.frame_skip = self.frame_skip
        cfg.control_steps = self.control_steps; cfg.max_episode_steps = self.max_episode_steps
        cfg.target_in_the_air = int(target_in_the_air); cfg.auto_reset = int(auto_reset)
        cfg.block_gripper = int(block_gripper)
        cfg.distance_threshold = self.distance_threshold; cfg.height_offset = height
        for k in range(3): cfg.initial_gripper_xpos[k] = igx[k]
        for k, v in enumerate(qpos): cfg.init_qpos[k] = v
        for k, v in enumerate(qvel): cfg.init_qvel[k] = v
        for k, v in enumerate(ctrl): cfg.init_ctrl[k] = v
        if domain_randomization:
            cfg.dr_enable = 1
            cfg.dr_mass_range[0], cfg.dr_mass_range[1] = domain_randomization.get("mass", (1.0, 1.0))
            cfg.dr_friction_range[0], cfg.dr_friction_range[1] = domain_randomization.get("friction", (1.0, 1.0))
        cfg.seed = int(seed) & (2 ** 64 - 1); cfg.env_id_offset = int(env_id_offset)
        self._cfg = cfg
        if model is None and (weld_rule != "common" or contact_rule != "mujoco"):
            # weld_rule "mujoco": the mocap weld with MuJoCo's recalled row weights (rotational rows softer); contact_rule "keyframe": the
            # pyramid regulariser that reproduces the cube's rest height of the reference's keyframes (include/mcg.h: contact_rpy)
            from .model.specialize import specialize
            model = _abi.McgModel.from_spec(specialize(load_table(True, mesh_inertia, mocap), weld_rule=weld_rule, contact_rule=contact_rule))
        if model is None:     # built-in block; a caller-supplied mcg_model (tests, custom robots) overrides it
            model = _abi.McgModel()
            variant = (1 if mesh_inertia == "exact" else 0) + (2 if mocap else 0)     # 2, 3: mocap body + weld (mycobot280_mocap.xml)
            _abi.check(self._lib.mcg_default_model(variant, C.byref(model)), "mcg_default_model")
        self._model = model
        self._h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _abi.check(self._lib.mcg_create(C.byref(cfg), C.byref(model), None, 0, dev_index, C.byref(self._h)), "mcg_create")      # built-in polytope tables
        self.obs_dim = self._lib.mcg_obs_dim(self._h)
        self.action_dim = self._lib.mcg_action_dim(self._h)
        self.nq, self.nv = self._lib.mcg_nq(self._h), self._lib.mcg_nv(self._h)

        # spaces (mycobot.py:108-110, 117-130)
        self.single_action_space = Box(-1.0, 1.0, (self.action_dim,), np.float32)
        self.single_observation_space = Dict({
            "desired_goal": Box(-np.inf, np.inf, (3,), np.float64),
            "achieved_goal": Box(-np.inf, np.inf, (3,), np.float64),
            "observation": Box(-np.inf, np.inf, (self.obs_dim,), np.float64)})
        self.action_space = batch_box(self.single_action_space, self.num_envs)
        self.observation_space = Dict({k: batch_box(v, self.num_envs) for k, v in self.single_observation_space.spaces.items()})

        n, D, dev = self.num_envs, self.obs_dim, self.device
        # flags are torch.bool buffers the kernel fills with 0/1 bytes: no conversion kernels on the step path
        f64 = dict(dtype=torch.float64, device=dev); u8 = dict(dtype=torch.bool, device=dev)
        self._buf = {
            "obs": torch.zeros(n, D, **f64), "achieved_goal": torch.zeros(n, 3, **f64),
            "desired_goal": torch.zeros(n, 3, **f64), "reward": torch.zeros(n, **f64),
            "terminated": torch.zeros(n, **u8), "truncated": torch.zeros(n, **u8), "is_success": torch.zeros(n, **u8),
            "final_obs": torch.zeros(n, D, **f64), "final_achieved": torch.zeros(n, 3, **f64),
            "final_desired": torch.zeros(n, 3, **f64), "ep_return": torch.zeros(n, **f64),
            "ep_length": torch.zeros(n, dtype=torch.int32, device=dev)}
        self._out = _abi.McgStepOut(**{k: v.data_ptr() for k, v in self._buf.items()})
        self._closed = False
        self._needs_reset = True

    # ------------------------------------------------------------------------------------------- Gymnasium API
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _obs(self):
        b = self._buf
        return {"observation": b["obs"], "achieved_goal": b["achieved_goal"], "desired_goal": b["desired_goal"]}

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None, mask: Optional[torch.Tensor] = None):
        """-> (obs dict of device tensors [N, ...], {}).  ``seed`` re-keys the reset RNG (env i uses stream i)."""
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if m.shape != (self.num_envs,):
                raise ValueError("mask must have shape (num_envs,)")
        with torch.cuda.device(self.device):
            _abi.check(self._lib.mcg_reset(self._h, None if m is None else C.c_void_p(m.data_ptr()),
                                           int(seed is not None), C.c_uint64((seed or 0) & (2 ** 64 - 1)),
                                           C.byref(self._out), self._stream()), "mcg_reset")
        self._needs_reset = False
        return self._obs(), {}

    def step(self, actions, copy: bool = True):
        """actions: float32 [N, A] (device tensor; numpy / CPU tensors are copied over).
        -> (obs, reward[N], terminated[N] bool, truncated[N] bool, info).  Like the reference (mycobot.py:280-282) the returned
        arrays are fresh copies; ``copy=False`` hands out the engine's output buffers instead, which the next call overwrites
        (``step_async`` is the raw, packaging-free variant).  The sparse reward is float32, dense / shaped float64 (mycobot.py:293-298)."""
        if self._needs_reset:
            raise RuntimeError("Cannot call env.step() before calling env.reset()")     # OrderEnforcing [RECALL]
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.device).contiguous()
        if a.shape != (self.num_envs, self.action_dim):
            raise ValueError(f"actions must have shape {(self.num_envs, self.action_dim)}, got {tuple(a.shape)}")
        with torch.cuda.device(self.device):
            _abi.check(self._lib.mcg_step(self._h, C.c_void_p(a.data_ptr()), C.byref(self._out), self._stream()), "mcg_step")
        b = {k: v.clone() for k, v in self._buf.items()} if copy else self._buf
        if self.reward_type == "sparse":
            b = dict(b, reward=b["reward"].float())
        terminated, truncated = b["terminated"], b["truncated"]
        done = truncated          # truncated = is_success | time-limit, so it already covers terminated (D-4)
        info = {"is_success": b["is_success"],
                "final_observation": {"observation": b["final_obs"], "achieved_goal": b["final_achieved"],
                                      "desired_goal": b["final_desired"]},
                "_final_observation": done,
                "episode": {"r": b["ep_return"], "l": b["ep_length"]}, "_episode": done}
        obs = {"observation": b["obs"], "achieved_goal": b["achieved_goal"], "desired_goal": b["desired_goal"]}
        return obs, b["reward"], terminated, truncated, info

    def step_async(self, actions: torch.Tensor) -> dict:
        """Launch one step on the current stream and return the raw output buffers (no Python-side packaging).
        `actions` must already be a contiguous float32 [N, A] tensor on this device."""
        with torch.cuda.device(self.device):
            _abi.check(self._lib.mcg_step(self._h, C.c_void_p(actions.data_ptr()), C.byref(self._out), self._stream()), "mcg_step")
        return self._buf

    def compute_reward(self, achieved_goal, desired_goal, info=None):
        """Batched GoalEnv reward (mycobot.py:289-298); sparse is returned as float32 like the reference."""
        if self.reward_type == "reward_shaping":
            raise NotImplementedError("reward_shaping depends on simulator state, not on the goals alone")
        ag = torch.as_tensor(achieved_goal, dtype=torch.float64, device=self.device).contiguous()
        dg = torch.as_tensor(desired_goal, dtype=torch.float64, device=self.device).contiguous()
        assert ag.shape == dg.shape                                                 # utils.py:25
        out = torch.empty(ag.shape[:-1], dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _abi.check(self._lib.mcg_compute_reward(C.c_void_p(ag.data_ptr()), C.c_void_p(dg.data_ptr()), out.numel(),
                                                    _REWARDS[self.reward_type], self.distance_threshold,
                                                    C.c_void_p(out.data_ptr()), self._stream()), "mcg_compute_reward")
        return out.float() if self.reward_type == "sparse" else out

    # ------------------------------------------------------------------------------------- state (tests, checkpoints)
    def _state_bufs(self):
        n, dev = self.num_envs, self.device
        f64 = dict(dtype=torch.float64, device=dev)
        return {"qpos": torch.zeros(self.nq, n, **f64), "qvel": torch.zeros(self.nv, n, **f64),
                "ctrl": torch.zeros(7, n, **f64), "warm": torch.zeros(self.nv, n, **f64),
                "qpos_lag": torch.zeros(self.nq, n, **f64), "goal": torch.zeros(3, n, **f64),
                "elapsed": torch.zeros(n, dtype=torch.int32, device=dev),
                "episode": torch.zeros(n, dtype=torch.int32, device=dev),
                "dr_scale": torch.ones(2, n, **f64), "ep_return": torch.zeros(n, **f64),
                "ep_length": torch.zeros(n, dtype=torch.int32, device=dev)}

    def get_state(self) -> dict:
        """SoA tensors [dim, N] (the engine's layout): qpos qvel ctrl warm qpos_lag goal elapsed episode dr_scale, the running
        episode statistics ep_return / ep_length, and `seed` (the base seed of the reset streams, a one-element int64 tensor): everything a
        freshly constructed engine needs to continue this one's trajectories, auto-resets included."""
        s = self._state_bufs()
        st = _abi.McgState(**{k: v.data_ptr() for k, v in s.items()})
        with torch.cuda.device(self.device):
            _abi.check(self._lib.mcg_get_state(self._h, C.byref(st), self._stream()), "mcg_get_state")
        seed = int(self._lib.mcg_get_seed(self._h))
        s["seed"] = torch.tensor([seed - 2 ** 64 if seed >= 2 ** 63 else seed], dtype=torch.int64)     # uint64 carried in an int64 tensor
        return s

    def set_state(self, **state):
        if state.get("seed") is not None:
            _abi.check(self._lib.mcg_set_seed(self._h, C.c_uint64(int(torch.as_tensor(state["seed"]).reshape(-1)[0]) & (2 ** 64 - 1))), "mcg_set_seed")
        keep = {}
        for k, ref in self._state_bufs().items():
            if k in state and state[k] is not None:
                t = torch.as_tensor(state[k], dtype=ref.dtype, device=self.device).contiguous()
                if t.shape != ref.shape:
                    raise ValueError(f"{k}: expected shape {tuple(ref.shape)}, got {tuple(t.shape)}")
                keep[k] = t
        st = _abi.McgState(**{k: v.data_ptr() for k, v in keep.items()})
        with torch.cuda.device(self.device):
            _abi.check(self._lib.mcg_set_state(self._h, C.byref(st), self._stream()), "mcg_set_state")
            torch.cuda.current_stream(self.device).synchronize()    # `keep` must outlive the copy kernel
        self._needs_reset = False

    state_dict = get_state

    def load_state_dict(self, sd):
        self.set_state(**sd)

    def time_steps(self, actions, steps: int) -> float:
        """Milliseconds for `steps` back-to-back step kernels, measured with HIP events on the launch stream."""
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.device).contiguous()
        ms = C.c_float()
        with torch.cuda.device(self.device):
            _abi.check(self._lib.mcg_time_steps(self._h, C.c_void_p(a.data_ptr()), C.byref(self._out), int(steps),
                                                self._stream(), C.byref(ms)), "mcg_time_steps")
        return float(ms.value)

    def counters(self, clear: bool = False) -> dict:
        """Event counters of the engine (include/mcg.h: mcg_counters); synchronises the device."""
        c = _abi.McgCounters()
        _abi.check(self._lib.mcg_get_counters(self._h, C.byref(c), int(clear)), "mcg_get_counters")
        return {n: int(getattr(c, n)) for n, _ in c._fields_}

    def debug_contacts(self) -> dict:
        """TEST / DEBUG: the collision pass of the current state as the kernels see it (mcg_debug_contacts): per env the number of
        list entries, the contacts the cap cut off, and per entry dist, pos[3], normal[3], pair type, multiplicity, D."""
        n = self.num_envs
        count = torch.zeros(n, dtype=torch.int32, device=self.device); dropped = torch.zeros_like(count)
        data = torch.zeros(n, _abi.MAXCON, 10, dtype=torch.float64, device=self.device)
        _abi.check(self._lib.mcg_debug_contacts(self._h, count.data_ptr(), dropped.data_ptr(), data.data_ptr(), self._stream()), "mcg_debug_contacts")
        torch.cuda.synchronize(self.device)
        return {"count": count, "dropped": dropped, "dist": data[:, :, 0], "pos": data[:, :, 1:4], "normal": data[:, :, 4:7],
                "type": data[:, :, 7].to(torch.int32), "mult": data[:, :, 8], "D": data[:, :, 9]}

    def close(self):
        if not self._closed and getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            self._lib.mcg_destroy(self._h)
            self._closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make(env_id: str, num_envs: int = 1, **kwargs) -> MyCobotVecEnv:
    """``gymnasium.make``-style factory over the reference's id table, vectorised."""
    kw = _spec(env_id)
    kw.update(kwargs)
    return MyCobotVecEnv(num_envs, **kw)
