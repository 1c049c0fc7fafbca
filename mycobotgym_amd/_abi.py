"""ctypes mirror of ``include/mcg.h`` and loader of the in-tree HIP library.

The library is the product: if it is missing or cannot be loaded this module raises --
there is no CPU or PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCG_LIB selects another build of the same library (kernel A/B timing, tools/ab_bench.py); default: the in-tree one
LIB_PATH = os.environ.get("MCG_LIB") or os.path.join(_HERE, "libmycobot_hip.so")

MCG_OK, MCG_ERR_ARG, MCG_ERR_HIP, MCG_ERR_UNSUPPORTED = 0, 1, 2, 3
CTRL_JOINT, CTRL_IK, CTRL_MOCAP = 0, 1, 2
ABI_VERSION = 8
MAXCON, NMESH = 16, 14
REWARD_SPARSE, REWARD_DENSE, REWARD_SHAPING = 0, 1, 2

d = C.c_double


class McgBody(C.Structure):
    _fields_ = [("r", d * 3), ("mass", d), ("mc", d * 3), ("inertia", d * 6), ("armature", d), ("damping", d), ("hull_rad", d)]


class McgModel(C.Structure):
    _fields_ = [
        ("timestep", d),
        ("base_pos", d * 3), ("base_mat", d * 9), ("gravity_base", d * 3),
        ("body", (d * 16) * 13),          # mcg_body[13], see McgBody for the layout of one row
        ("cube_damping", d * 6),
        ("jnt_range", (d * 2) * 12),
        ("limit_par", (d * 10) * 12),
        ("limit_diag", d * 12),
        ("eq_anchor1", (d * 3) * 2), ("eq_anchor2", (d * 3) * 2),
        ("eq_par", (d * 10) * 3), ("eq_diag", d * 3),
        ("act_gain", d * 7), ("act_bias", (d * 3) * 7), ("act_ctrlrange", (d * 2) * 7),
        ("act_forcerange", (d * 2) * 7), ("tendon_coef", d * 2),
        ("site_eef", d * 3),
        ("cube_half", d * 3), ("table_pos", d * 3), ("table_half", d * 3), ("pad_box", (d * 6) * 2),
        ("contact_par", (d * 15) * 7),
        ("contact_diag", (d * 2) * 5),
        ("mesh_box", (d * 6) * NMESH), ("mesh_mult", d), ("mesh_fric", d), ("pair_tran", d * (5 + 2 * NMESH)),
        ("geom_friction0", d * 3),
        ("base_quat", d * 4), ("weld_on", d), ("weld_par", d * 10), ("weld_diag", d * 2), ("weld_anchor", d * 3),
        ("weld_relpos", d * 3), ("weld_relquat", d * 4), ("weld_torquescale", d),
        ("target0", d * 3),
        ("contact_rpy", d),
    ]

    @classmethod
    def from_spec(cls, spec: dict) -> "McgModel":
        """Fill from ``mycobotgym_amd.model.specialize.specialize`` output."""
        m = cls()
        for name, ctype in cls._fields_:
            if name not in spec:
                continue                       # cube fields of a Reach-only table stay zero
            v = np.ascontiguousarray(np.asarray(spec[name], dtype=np.float64))
            if ctype is d:
                setattr(m, name, float(v.reshape(-1)[0]))
            else:
                dst = np.ctypeslib.as_array(getattr(m, name))
                dst[...] = v.reshape(dst.shape)
        return m


class McgConfig(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32), ("has_object", C.c_int32), ("controller", C.c_int32), ("fetch_env", C.c_int32),
        ("reward_type", C.c_int32), ("frame_skip", C.c_int32), ("control_steps", C.c_int32),
        ("max_episode_steps", C.c_int32), ("target_in_the_air", C.c_int32), ("auto_reset", C.c_int32),
        ("dr_enable", C.c_int32), ("block_gripper", C.c_int32),
        ("distance_threshold", d), ("height_offset", d), ("initial_gripper_xpos", d * 3),
        ("init_qpos", d * 19), ("init_qvel", d * 18), ("init_ctrl", d * 7),
        ("dr_mass_range", d * 2), ("dr_friction_range", d * 2),
        ("seed", C.c_uint64), ("env_id_offset", C.c_int64),
    ]


class McgCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("reset_cap_hits", "bad_state_resets", "contacts_dropped", "coupled_env_substeps")]


class McgStepOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "obs", "achieved_goal", "desired_goal", "reward", "terminated", "truncated", "is_success",
        "final_obs", "final_achieved", "final_desired", "ep_return", "ep_length")]


class McgState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qpos", "qvel", "ctrl", "warm", "qpos_lag", "goal", "elapsed", "episode", "dr_scale",
                                          "ep_return", "ep_length")]


EXPORTS = ("mcg_abi_version", "mcg_last_error", "mcg_default_model", "mcg_create", "mcg_destroy", "mcg_obs_dim",
           "mcg_action_dim", "mcg_nq", "mcg_nv", "mcg_reset", "mcg_step", "mcg_get_state", "mcg_set_state",
           "mcg_compute_reward", "mcg_time_steps", "mcg_get_seed", "mcg_set_seed", "mcg_get_counters", "mcg_debug_contacts")

_lib = None


class McgError(RuntimeError):
    pass


def load():
    """dlopen the in-tree library; raise if it has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise McgError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                       "(hipcc --offload-arch=gfx950); this package has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    L.mcg_abi_version.restype = C.c_int
    L.mcg_last_error.restype = C.c_char_p
    L.mcg_default_model.argtypes = [C.c_int, C.POINTER(McgModel)]
    L.mcg_create.argtypes = [C.POINTER(McgConfig), C.POINTER(McgModel), C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
    L.mcg_destroy.argtypes = [C.c_void_p]
    L.mcg_destroy.restype = None
    for f in ("mcg_obs_dim", "mcg_action_dim", "mcg_nq", "mcg_nv"):
        getattr(L, f).argtypes = [C.c_void_p]
    L.mcg_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.POINTER(McgStepOut), C.c_void_p]
    L.mcg_step.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(McgStepOut), C.c_void_p]
    L.mcg_get_state.argtypes = [C.c_void_p, C.POINTER(McgState), C.c_void_p]
    L.mcg_set_state.argtypes = [C.c_void_p, C.POINTER(McgState), C.c_void_p]
    L.mcg_compute_reward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    if hasattr(L, "mcg_get_seed"):        # absent only from pre-v3 builds selected through MCG_LIB for A/B timing
        L.mcg_get_seed.argtypes = [C.c_void_p]; L.mcg_get_seed.restype = C.c_uint64
        L.mcg_set_seed.argtypes = [C.c_void_p, C.c_uint64]
    if hasattr(L, "mcg_get_counters"):    # absent only from older builds selected through MCG_LIB for A/B timing
        L.mcg_get_counters.argtypes = [C.c_void_p, C.POINTER(McgCounters), C.c_int]
        L.mcg_debug_contacts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mcg_time_steps.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(McgStepOut), C.c_int, C.c_void_p, C.POINTER(C.c_float)]
    _lib = L
    return L


def check(code: int, what: str = ""):
    if code != MCG_OK:
        msg = load().mcg_last_error().decode(errors="replace")
        raise McgError(f"{what} failed (code {code}): {msg}")
