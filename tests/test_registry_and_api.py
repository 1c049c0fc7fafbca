"""Host-side logic that needs no GPU: id table, spaces, loud failure without a device, ABI surface."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_reproduces_reference_ids():
    from mycobotgym_amd.registry import REGISTRY, spec
    assert len(REGISTRY) == 50                                                   # 30 -v0 + 20 -v1 (Appendix E)
    assert sum(k.endswith("-v0") for k in REGISTRY) == 30 and sum(k.endswith("-v1") for k in REGISTRY) == 20
    assert not any("Fetch" in k and "-joint-" in k for k in REGISTRY)            # __init__.py:21-24
    assert not any("RewardShaping" in k and k.endswith("-v1") for k in REGISTRY)  # __init__.py:37-39
    s = spec("MyCobotReach-Dense-IK-v0")
    assert s == {"model_path": "./assets/mycobot280.xml", "reward_type": "dense", "has_object": False,
                 "controller_type": "IK", "fetch_env": False, "image_obs": False}
    assert spec("MyCobotFetchPickAndPlace-Sparse-mocap-v0")["model_path"] == "./assets/mycobot280_mocap.xml"
    with pytest.raises(KeyError):
        spec("MyCobotReach-v0")                                                   # no such literal id exists


def test_spaces_surface():
    from mycobotgym_amd.spaces import Box, Dict, batch_box
    a = Box(-1.0, 1.0, (7,), np.float32)
    assert a.sample().dtype == np.float32 and a.contains(a.sample()) and not a.contains(np.full(7, 2, np.float32))
    o = Dict({"observation": Box(-np.inf, np.inf, (10,), np.float64)})
    assert o["observation"].shape == (10,) and batch_box(a, 4).shape == (4, 7)


def test_no_cpu_fallback(built):
    """Constructing an env without a GPU must raise, not silently compute somewhere else."""
    import torch
    from mycobotgym_amd import MyCobotVecEnv, _abi
    with pytest.raises(_abi.McgError):
        MyCobotVecEnv(4, has_object=False, controller_type="joint", device="cpu")
    if not torch.cuda.is_available():
        with pytest.raises(_abi.McgError):
            MyCobotVecEnv(4, has_object=False, controller_type="joint", device="cuda:0")


def test_unsupported_configurations_raise():
    from mycobotgym_amd import MyCobotVecEnv
    with pytest.raises(NotImplementedError):
        MyCobotVecEnv(1, controller_type="delta_joint")          # no branch in the reference's step() (SURVEY D-10)
    with pytest.raises(NotImplementedError):
        MyCobotVecEnv(1, image_obs=True)
    with pytest.raises(AssertionError, match="Joint controller not supported for Fetch env"):   # mycobot.py:96
        MyCobotVecEnv(1, controller_type="joint", fetch_env=True)
    with pytest.raises(ValueError):
        MyCobotVecEnv(1, controller_type="bogus")


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "mcg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mcg_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(built):
    from mycobotgym_amd import _abi
    lib = _abi.load()
    names = _declared_functions()
    assert set(names) == set(_abi.EXPORTS), (names, _abi.EXPORTS)
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.mcg_abi_version() == _abi.ABI_VERSION == 8


def test_ctypes_mirror_matches_header_layout(built, tmp_path):
    """sizeof/offsetof of every ABI struct as the C compiler sees include/mcg.h == the ctypes mirror."""
    from mycobotgym_amd import _abi
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mcg.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(mcg_model),sizeof(mcg_config),sizeof(mcg_step_out),sizeof(mcg_state),sizeof(mcg_body),'
                   'offsetof(mcg_model,limit_par),offsetof(mcg_config,seed),offsetof(mcg_model,contact_par));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(_abi.McgModel), C.sizeof(_abi.McgConfig), C.sizeof(_abi.McgStepOut), C.sizeof(_abi.McgState),
            C.sizeof(_abi.McgBody), _abi.McgModel.limit_par.offset, _abi.McgConfig.seed.offset, _abi.McgModel.contact_par.offset]
    assert got == want


def test_default_model_blocks_match_specializer(built):
    """The model block compiled into the library is the specialiser's output for the committed tables."""
    from mycobotgym_amd import _abi
    from mycobotgym_amd.model.mjcf import _np_model
    from mycobotgym_amd.model.specialize import specialize
    from tests.common import load_json
    lib = _abi.load()
    for variant, name in ((0, "mycobot280"), (1, "mycobot280_exactmesh"), (2, "mycobot280_mocap"), (3, "mycobot280_mocap_exactmesh")):
        got = _abi.McgModel()
        assert lib.mcg_default_model(variant, C.byref(got)) == 0
        want = _abi.McgModel.from_spec(specialize(_np_model(load_json(name))))
        assert bytes(got) == bytes(want)
    assert lib.mcg_default_model(7, C.byref(got)) != 0 and b"variant" in lib.mcg_last_error()


def test_create_argument_errors_without_touching_the_gpu(built):
    from mycobotgym_amd import _abi
    lib = _abi.load()
    h = C.c_void_p()
    cfg = _abi.McgConfig(n_envs=0)
    assert lib.mcg_create(C.byref(cfg), None, None, 0, 0, C.byref(h)) == _abi.MCG_ERR_ARG
    cfg = _abi.McgConfig(n_envs=4, controller=1, fetch_env=0, reward_type=7, frame_skip=20, control_steps=5, max_episode_steps=50)
    assert lib.mcg_create(C.byref(cfg), None, None, 0, 0, C.byref(h)) == _abi.MCG_ERR_ARG and b"reward_type" in lib.mcg_last_error()
    cfg = _abi.McgConfig(n_envs=4, controller=0, fetch_env=1, frame_skip=20, control_steps=5, max_episode_steps=50)
    assert lib.mcg_create(C.byref(cfg), None, None, 0, 0, C.byref(h)) == _abi.MCG_ERR_ARG
    assert b"Joint controller not supported for Fetch env" in lib.mcg_last_error()


def test_mocap_tables_and_initial_state():
    """Host side of the mocap controller: model variant selection and the constructor snapshot."""
    from mycobotgym_amd.vec_env import initial_state, load_table
    tab = load_table(False, "legacy", mocap=True)
    assert tab["nu"] == 1 and any(tab["body_mocap"]) and [e["type"] for e in tab["eq"]] == [1, 0, 0, 2]
    qpos, qvel, ctrl, igx, height = initial_state(False, True, "legacy", mocap=True)
    key = load_table(True, "legacy", mocap=True)["keys"][0]
    assert np.allclose(qpos, key["qpos"][:12]) and ctrl.shape == (7,) and ctrl[6] == key["ctrl"][0]
    # the keyframe's mocap position is the gripper position up to the weld's sag
    assert np.linalg.norm(np.asarray(key["mpos"]) - igx) < 2e-3
    qpos, qvel, ctrl, igx, height = initial_state(False, False, "legacy", mocap=True)
    mb = tab["body_mocap"].index(True)
    assert np.allclose(igx, tab["body_pos"][mb], atol=1e-8)      # mocap.xml:3 == FK(EEF; qpos0)
