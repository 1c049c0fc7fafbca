"""ctypes front-end of the CPU oracle (ORACLE -- TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this.
Builds ``oracle/_build/libmco_oracle.so`` on demand with the Makefile next to this file.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmco_oracle.so")
_lib = None

MAXNQ, MAXNV, MAXU = 24, 24, 8


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("mco_physics.c", "mco_physics.h", "mco_env.c", "mco_env.h",
                                             "mco_collision.c", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.mco_model_sizeof.restype = C.c_int
        L.mco_data_sizeof.restype = C.c_int
        L.mco_env_config_sizeof.restype = C.c_int
        L.mco_envs_create.restype = C.c_void_p
        L.mco_envs_create.argtypes = [C.c_void_p, C.c_void_p]
        L.mco_envs_destroy.argtypes = [C.c_void_p]
        L.mco_envs_data.restype = C.c_void_p
        L.mco_envs_data.argtypes = [C.c_void_p, C.c_int]
        L.mco_envs_obs_dim.argtypes = [C.c_void_p]
        L.mco_envs_action_dim.argtypes = [C.c_void_p]
        L.mco_envs_initial_gripper_xpos.argtypes = [C.c_void_p, C.c_void_p]
        L.mco_envs_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mco_envs_step.argtypes = [C.c_void_p] + [C.c_void_p] * 13
        L.mco_envs_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        L.mco_envs_set_state.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        for f in ("mco_setconst", "mco_forward", "mco_step", "mco_reset_data"):
            getattr(L, f).argtypes = [C.c_void_p] * (1 if f == "mco_setconst" else 2)
        L.mco_jac_site.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.mco_energy.restype = C.c_double
        L.mco_energy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mco_compute_reward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        _lib = L
    return _lib


class EnvConfig(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32), ("has_object", C.c_int32), ("controller", C.c_int32), ("fetch_env", C.c_int32),
        ("reward_type", C.c_int32), ("frame_skip", C.c_int32), ("control_steps", C.c_int32),
        ("max_episode_steps", C.c_int32), ("target_in_the_air", C.c_int32), ("auto_reset", C.c_int32),
        ("eef_site", C.c_int32), ("obj_site", C.c_int32), ("obj_jnt", C.c_int32), ("grip_jnt", C.c_int32 * 2),
        ("n_threads", C.c_int32), ("dr_enable", C.c_int32), ("pad_geom", C.c_int32 * 2), ("obj_geom", C.c_int32),
        ("block_gripper", C.c_int32), ("finger_jnt", C.c_int32 * 2), ("tcp_body", C.c_int32), ("pad_", C.c_int32),
        ("distance_threshold", C.c_double), ("height_offset", C.c_double),
        ("init_qpos", C.c_double * MAXNQ), ("init_qvel", C.c_double * MAXNV), ("init_ctrl", C.c_double * MAXU),
        ("dr_mass_range", C.c_double * 2), ("dr_friction_range", C.c_double * 2), ("init_mocap", C.c_double * 7),
        ("seed", C.c_uint64), ("env_id_offset", C.c_int64),
    ]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleModel:
    """An ``mco_model`` filled from one of ``mycobotgym_amd/assets/*.json``."""

    def __init__(self, table: dict, enable_contact: bool = False, scope_geom: int = -1, mesh_collision: bool = True, maxentry: int = 16):
        L = lib()
        self.table = table
        self.buf = C.create_string_buffer(L.mco_model_sizeof())
        self.nq, self.nv, self.nu = table["nq"], table["nv"], table["nu"]
        si, sd = self._set_i, self._set_d
        for k in ("nbody", "njnt", "nq", "nv", "ngeom", "nsite", "nu", "neq", "ntendon"):
            si(k, [table[k]])
        si("nexclude", [len(table["excludes"])])
        si("enable_contact", [int(enable_contact)])
        si("collide_scope_geom", [int(scope_geom)])
        if scope_geom >= 0:        # the build's scoped set also holds finger pad <-> table / ground plane (box-box, plane-box)
            extra = [0] * table["ngeom"]
            for g in range(table["ngeom"]):
                if table["geom_type"][g] == 7: continue
                if table["body_weldid"][table["geom_body"][g]] == 0: extra[g] = 1
                elif table["geom_name"][g] in ("right_finger_layer", "left_finger_layer"): extra[g] = 2
            si("collide_extra", extra)
        # the mesh geoms' collision polytopes (both twins of every mesh): ground, table, cube (mco_collision.c)
        gp = [-1] * 48
        if scope_geom >= 0 and mesh_collision:
            from mycobotgym_amd.model import polytope as pt
            blob, _ = pt.load_asset()
            self._poly = np.ascontiguousarray(blob, dtype=np.float64)          # kept alive: the model holds the pointer
            for g in range(table["ngeom"]):
                if table["geom_type"][g] == 7 and table["geom_mesh"][g] in pt.MESH_NAMES and table["geom_contype"][g] and table["geom_conaffinity"][g]:
                    assert np.allclose(table["geom_pos"][g], 0) and np.allclose(table["geom_quat"][g], [1, 0, 0, 0])
                    gp[g] = pt.MESH_NAMES.index(table["geom_mesh"][g])
            L.mco_model_set_poly.argtypes = [C.c_void_p, C.c_void_p]
            L.mco_model_set_poly(self.buf, _ptr(self._poly))
        si("geom_poly", gp); si("maxentry", [int(maxentry)])
        sd("timestep", [table["opt"]["timestep"]]); sd("gravity", table["opt"]["gravity"])
        for k in ("body_parent", "body_rootid", "body_weldid", "body_dofadr", "body_dofnum", "jnt_type", "jnt_body",
                  "jnt_qposadr", "jnt_dofadr", "dof_body", "dof_jnt", "dof_parent", "geom_type", "geom_body",
                  "geom_condim", "geom_contype", "geom_conaffinity", "site_body"):
            si(k, table[k])
        si("jnt_limited", [int(x) for x in table["jnt_limited"]])
        mocap = [bool(x) for x in table.get("body_mocap", [])]
        ids, nxt = [-1] * 32, 0                     # MCO_MAXBODY entries; -1 = not a mocap body
        for b, is_m in enumerate(mocap):
            if is_m: ids[b] = nxt; nxt += 1
        assert nxt <= 1, "the oracle holds one mocap body (MCO_MAXMOCAP)"
        si("body_mocapid", ids)
        for k in ("body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia", "jnt_pos",
                  "jnt_axis", "jnt_range", "jnt_solref", "jnt_solimp", "dof_armature", "dof_damping", "qpos0",
                  "geom_pos", "geom_quat", "geom_size", "geom_friction", "geom_solref", "geom_solimp",
                  "site_pos", "site_quat"):
            v = table[k]
            sd(k, v)
        acts = table["actuators"]
        si("act_trntype", [0 if a["trntype"] == "joint" else 1 for a in acts])
        si("act_trnid", [a["trnid"] for a in acts])
        si("act_ctrllimited", [int(a["ctrllimited"]) for a in acts])
        si("act_forcelimited", [int(a["forcelimited"]) for a in acts])
        sd("act_gear", [a["gear"] for a in acts])
        sd("act_gainprm", [a["gainprm"] for a in acts]); sd("act_biasprm", [a["biasprm"] for a in acts])
        sd("act_ctrlrange", [a["ctrlrange"] for a in acts]); sd("act_forcerange", [a["forcerange"] for a in acts])
        tj = np.zeros((2, 4), dtype=np.int32); tc = np.zeros((2, 4))
        for t, ten in enumerate(table["tendons"]):
            tj[t, :len(ten["joints"])] = ten["joints"]; tc[t, :len(ten["coefs"])] = ten["coefs"]
        si("ten_num", [len(t["joints"]) for t in table["tendons"]]); si("ten_jnt", tj); sd("ten_coef", tc)
        eq = table["eq"]
        si("eq_type", [e["type"] for e in eq]); si("eq_obj1", [e["obj1"] for e in eq]); si("eq_obj2", [e["obj2"] for e in eq])
        sd("eq_data", [e["data"] for e in eq]); sd("eq_solref", [e["solref"] for e in eq]); sd("eq_solimp", [e["solimp"] for e in eq])
        if table["excludes"]:
            si("exclude", table["excludes"])
        L.mco_setconst(self.buf)

    def _set_i(self, name, v):
        a = np.ascontiguousarray(np.asarray(v, dtype=np.int32).ravel())
        if a.size and lib().mco_model_set_i(self.buf, name.encode(), _ptr(a), a.size) != 0:
            raise KeyError(name)

    def _set_d(self, name, v):
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel())
        if a.size and lib().mco_model_set_d(self.buf, name.encode(), _ptr(a), a.size) != 0:
            raise KeyError(name)

    def get(self, name, n):
        out = np.zeros(n)
        if lib().mco_model_get_d(self.buf, name.encode(), _ptr(out), n) != 0:
            raise KeyError(name)
        return out

    @classmethod
    def from_json(cls, path: str, **kw):
        with open(path) as f:
            return cls(json.load(f), **kw)


class OracleData:
    """A single ``mco_data`` for low-level checks (forward / step on one state)."""

    def __init__(self, model: OracleModel, ptr=None):
        self.model = model
        if ptr is None:
            self.buf = C.create_string_buffer(lib().mco_data_sizeof())
            self.ptr = C.addressof(self.buf)
            lib().mco_reset_data(model.buf, C.c_void_p(self.ptr))
        else:
            self.ptr = ptr

    def get(self, name, shape, dtype=np.float64):
        n = int(np.prod(shape))
        out = np.zeros(n, dtype=dtype)
        fn = lib().mco_data_get_d if dtype == np.float64 else lib().mco_data_get_i
        fn.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        if fn(C.c_void_p(self.ptr), name.encode(), _ptr(out), n) != 0:
            raise KeyError(name)
        return out.reshape(shape)

    def _raw(self, name, n):
        # qpos, qvel, ctrl, qacc_warmstart are the leading members of mco_data, in that order
        off = {"qpos": 0, "qvel": MAXNQ, "ctrl": MAXNQ + MAXNV, "qacc_warmstart": MAXNQ + MAXNV + MAXU}[name]
        return np.ctypeslib.as_array((C.c_double * n).from_address(self.ptr + 8 * off))

    def set_state(self, qpos=None, qvel=None, ctrl=None, warm=None):
        m = self.model
        if qpos is not None: self._raw("qpos", m.nq)[:] = qpos
        if qvel is not None: self._raw("qvel", m.nv)[:] = qvel
        if ctrl is not None: self._raw("ctrl", m.nu)[:] = ctrl
        if warm is not None: self._raw("qacc_warmstart", m.nv)[:] = warm

    @property
    def qpos(self): return self._raw("qpos", self.model.nq)
    @property
    def qvel(self): return self._raw("qvel", self.model.nv)
    @property
    def ctrl(self): return self._raw("ctrl", self.model.nu)

    def set_mocap(self, pos, quat):
        L = lib(); L.mco_data_set_d.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        p = np.ascontiguousarray(pos, dtype=np.float64); q = np.ascontiguousarray(quat, dtype=np.float64)
        assert L.mco_data_set_d(C.c_void_p(self.ptr), b"mocap_pos", _ptr(p), 3) == 0
        assert L.mco_data_set_d(C.c_void_p(self.ptr), b"mocap_quat", _ptr(q), 4) == 0

    def forward(self): lib().mco_forward(self.model.buf, C.c_void_p(self.ptr))
    def step(self, n=1):
        for _ in range(n): lib().mco_step(self.model.buf, C.c_void_p(self.ptr))

    def dense(self, name):
        nv = self.model.nv
        return self.get(name, (MAXNV, MAXNV))[:nv, :nv]

    def vec(self, name, n=None):
        n = self.model.nv if n is None else n
        return self.get(name, (n,))

    def jac_site(self, site):
        nv = self.model.nv
        jp = np.zeros((3, nv)); jr = np.zeros((3, nv))
        lib().mco_jac_site(self.model.buf, C.c_void_p(self.ptr), _ptr(jp), _ptr(jr), site)
        return jp, jr

    def energy(self):
        pe = C.c_double(); ke = C.c_double()
        lib().mco_energy(self.model.buf, C.c_void_p(self.ptr), C.byref(pe), C.byref(ke))
        return pe.value, ke.value


class OracleEnvs:
    """Batched reference environments (the CPU side of every parity test)."""

    def __init__(self, model: OracleModel, cfg: EnvConfig):
        self.model, self.cfg = model, cfg
        self.h = C.c_void_p(lib().mco_envs_create(model.buf, C.byref(cfg)))
        self.n = cfg.n_envs
        self.obs_dim = lib().mco_envs_obs_dim(self.h)
        self.act_dim = lib().mco_envs_action_dim(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().mco_envs_destroy(self.h); self.h = None

    def _obs_bufs(self):
        return np.zeros((self.n, self.obs_dim)), np.zeros((self.n, 3)), np.zeros((self.n, 3))

    def initial_gripper_xpos(self):
        out = np.zeros(3); lib().mco_envs_initial_gripper_xpos(self.h, _ptr(out)); return out

    def reset(self, mask=None, seed=None):
        obs, ag, dg = self._obs_bufs()
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().mco_envs_reset(self.h, _ptr(m), int(seed is not None), C.c_uint64(seed or 0), _ptr(obs), _ptr(ag), _ptr(dg))
        return obs, ag, dg

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        assert a.shape == (self.n, self.act_dim), a.shape
        obs, ag, dg = self._obs_bufs()
        fobs, fag, fdg = self._obs_bufs()
        rew = np.zeros(self.n); term = np.zeros(self.n, np.uint8); trunc = np.zeros(self.n, np.uint8)
        succ = np.zeros(self.n, np.uint8); epr = np.zeros(self.n); epl = np.zeros(self.n, np.int32)
        lib().mco_envs_step(self.h, _ptr(a), _ptr(obs), _ptr(ag), _ptr(dg), _ptr(rew), _ptr(term), _ptr(trunc),
                            _ptr(succ), _ptr(fobs), _ptr(fag), _ptr(fdg), _ptr(epr), _ptr(epl))
        return dict(obs=obs, achieved=ag, desired=dg, reward=rew, terminated=term, truncated=trunc,
                    is_success=succ, final_obs=fobs, final_achieved=fag, final_desired=fdg,
                    ep_return=epr, ep_length=epl)

    def get_state(self):
        n, m = self.n, self.model
        s = dict(qpos=np.zeros((n, m.nq)), qvel=np.zeros((n, m.nv)), ctrl=np.zeros((n, m.nu)),
                 warm=np.zeros((n, m.nv)), qpos_lag=np.zeros((n, m.nq)), goal=np.zeros((n, 3)),
                 elapsed=np.zeros(n, np.int32), episode=np.zeros(n, np.int32))
        lib().mco_envs_get_state(self.h, *[_ptr(s[k]) for k in ("qpos", "qvel", "ctrl", "warm", "qpos_lag", "goal", "elapsed", "episode")])
        return s

    def set_state(self, **s):
        args = []
        for k in ("qpos", "qvel", "ctrl", "warm", "qpos_lag", "goal"):
            v = s.get(k)
            args.append(None if v is None else np.ascontiguousarray(v, dtype=np.float64))
        for k in ("elapsed", "episode"):
            v = s.get(k)
            args.append(None if v is None else np.ascontiguousarray(v, dtype=np.int32))
        lib().mco_envs_set_state(self.h, *[_ptr(a) for a in args])

    def data(self, i) -> OracleData:
        return OracleData(self.model, ptr=lib().mco_envs_data(self.h, i))


def compute_reward(achieved, desired, reward_type: int, threshold: float):
    a = np.ascontiguousarray(achieved, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(desired, dtype=np.float64).reshape(-1, 3)
    out = np.zeros(a.shape[0])
    lib().mco_compute_reward(_ptr(a), _ptr(d), a.shape[0], reward_type, threshold, _ptr(out))
    return out
