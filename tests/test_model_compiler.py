"""Model compiler (MJCF -> tables) against the reference-internal known answers of SURVEY.md Appendix E.

These are the only machine-checkable numbers the reference itself contains (it ships no tests): cross-file
constants such as "FK(EEF, qpos=0) equals the mocap body's rest pose".  They pin the kinematic tree, the
include/default-class resolution and the keyframes of the compiled tables.
"""
import os

import numpy as np
import pytest

from tests.common import ASSETS, load_json

REF_ASSETS = "/root/reference/mycobotgym/envs/assets"


def _np(tab):
    from mycobotgym_amd.model.mjcf import _np_model
    return _np_model(tab)


def test_dimensions_match_survey_appendix_a2():
    full, reach, mocap = load_json("mycobot280"), load_json("mycobot280_reach"), load_json("mycobot280_mocap")
    assert (full["nbody"], full["njnt"], full["nq"], full["nv"], full["nu"], full["ngeom"], full["nsite"]) == (25, 13, 19, 18, 7, 35, 3)
    assert (full["neq"], len(full["excludes"]), len(full["keys"]), full["ntendon"]) == (3, 8, 1, 1)
    assert (reach["nbody"], reach["nq"], reach["nv"]) == (24, 12, 12)
    assert (mocap["nbody"], mocap["ngeom"], mocap["nu"], mocap["neq"]) == (26, 39, 1, 4)
    assert len(full["keys"][0]["qpos"]) == 19 and len(full["keys"][0]["ctrl"]) == 7


def test_fk_known_answers():
    from mycobotgym_amd.model.refdyn import kinematics
    m = _np(load_json("mycobot280"))
    s = m["site_name"].index("EEF")
    # EEF at qpos0 == rest pose of the mocap body (mocap.xml:3)
    kin = kinematics(m, m["qpos0"])
    assert np.allclose(kin["site_xpos"][s], [0.0138673, 0.01864658, 0.61236], atol=1e-8)
    assert np.allclose(kin["site_xmat"][s], [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-12)
    # EEF at the joint keyframe (mycobot280.xml:6) == mocap keyframe mpos (mycobot280_mocap.xml:8)
    kin = kinematics(m, np.asarray(m["keys"][0]["qpos"]))
    assert np.allclose(kin["site_xpos"][s], [-0.05154491, 0.01053502, 0.3448586], atol=2e-7)
    # EEF at the mocap keyframe (mycobot280_mocap.xml:7): ~1 mm below mpos (weld sag)
    mm = _np(load_json("mycobot280_mocap"))
    kin = kinematics(mm, np.asarray(mm["keys"][0]["qpos"]))
    assert np.allclose(kin["site_xpos"][mm["site_name"].index("EEF")], [-0.051354, 0.009859, 0.343950], atol=2e-6)


def test_mass_budget_and_mesh_rules():
    m = load_json("mycobot280")
    name = {n: i for i, n in enumerate(m["body_name"])}
    # CAD inertials equal exact mesh volume x 1000 within 0.4 % (Appendix A.7)
    for link, mass in (("link1", 0.0427369), ("link2", 0.0668491), ("link3", 0.0533251), ("link4", 0.0240612), ("link5", 0.0327884)):
        assert abs(m["meshes"][link]["volume_exact"] * 1000 / mass - 1) < 4e-3
        assert m["body_mass"][name[link]] == mass
    # flange / gripper_base have no <inertial>: legacy rule over-counts, exact rule gives the CAD volume
    assert abs(m["body_mass"][name["flange"]] - 0.0843) < 1e-4 and abs(m["body_mass"][name["gripper_base"]] - 0.2142) < 1e-4
    e = load_json("mycobot280_exactmesh")
    assert abs(e["body_mass"][name["flange"]] - 0.02500) < 1e-5 and abs(e["body_mass"][name["gripper_base"]] - 0.03228) < 1e-5
    # cube and pads from primitive geoms at density 1000
    assert abs(m["body_mass"][name["object0"]] - 0.008) < 1e-12
    assert abs(m["body_mass"][name["right_finger_layer"]] - 1.04e-3) < 1e-12
    assert m["meshes"]["base_link"]["missing"]          # .MISSING_LARGE_BLOBS: static body, no dynamics


def test_default_classes_and_actuators():
    m = load_json("mycobot280")
    j = {n: i for i, n in enumerate(m["jnt_name"])}
    arm = [j[f"robot0:joint{k}"] for k in range(1, 7)]
    assert all(m["dof_armature"][m["jnt_dofadr"][a]] == 0.1 and m["dof_damping"][m["jnt_dofadr"][a]] == 1.0 for a in arm)
    drv = j["robot0:right_gear_joint"]
    assert m["jnt_range"][drv] == [0.0, 0.7] and m["jnt_limited"][drv] and m["jnt_solref"][drv] == [0.005, 1.0]
    assert m["dof_armature"][m["jnt_dofadr"][drv]] == 0.005 and m["dof_damping"][m["jnt_dofadr"][drv]] == 0.1
    fol, cpl = j["right_finger_joint"], j["right_hinge_joint"]
    assert m["jnt_limited"][fol] and m["dof_armature"][m["jnt_dofadr"][fol]] == 0.0
    assert not m["jnt_limited"][cpl]
    a = m["actuators"]
    assert [x["gainprm"][0] for x in a] == [4500, 4500, 3500, 2000, 2000, 2000, 70]
    assert [x["forcerange"][1] for x in a] == [87, 87, 87, 12, 12, 12, 5]
    assert a[6]["trntype"] == "tendon" and a[6]["biasprm"] == [0, -100, -10] and a[6]["ctrlrange"] == [0, 1]
    assert m["tendons"][0]["coefs"] == [0.5, 0.5]


def test_three_mass_matrix_formulations_agree():
    """numpy Jacobian-sum M, and the invweight0 the specialiser derives from it, are symmetric positive definite."""
    from mycobotgym_amd.model.refdyn import kinematics, mass_matrix, invweight0
    m = _np(load_json("mycobot280"))
    rng = np.random.default_rng(0)
    q = m["qpos0"].copy(); q[:12] = rng.uniform(-1, 1, 12); q[15:19] = rng.normal(size=4)
    M = mass_matrix(m, kinematics(m, q))
    assert np.allclose(M, M.T, atol=1e-18) and np.linalg.eigvalsh(M).min() > 0
    biw, diw = invweight0(m)
    assert np.all(diw > 0) and np.allclose(diw[12:15], 1 / 0.008) and np.allclose(diw[15:18], 1 / 5.333333333333336e-07)


def test_specializer_structure_and_welding():
    from mycobotgym_amd.model.specialize import specialize
    sp = specialize(_np(load_json("mycobot280")))
    # link6 carries flange + gripper_base (+ massless frames)
    assert abs(sp["mass"][5] - (0.0649501 + 0.0843151 + 0.214181)) < 1e-6
    assert abs(sp["mass"][7] - (0.00694636 + 0.00104)) < 1e-12
    assert np.allclose(sp["site_eef"], [0.13, -0.01, -0.001])
    assert np.allclose(sp["gravity_base"], [0, 0, 9.81])
    assert sp["body"].shape == (13, 16) and sp["eq_par"].shape == (3, 10)
    # refsafe: positive time constants are at least 2 timesteps; cube-pad mix is the direct (negative) solref
    assert np.isclose(sp["contact_par"][1][0], 20000 / 0.999 ** 2) and np.isclose(sp["contact_par"][1][1], 500 / 0.999)
    assert list(sp["contact_par"][0][10:]) == [1.0, 1.0, 0.3, 0.1, 0.1]


def test_generated_header_is_current():
    """mycobotgym_amd/csrc/model_gen.h must be what tools/gen_model_header.py emits from the committed tables."""
    import subprocess, sys, tempfile, shutil
    root = os.path.dirname(ASSETS.rstrip("/")).rsplit("/mycobotgym_amd", 1)[0]
    hdr = os.path.join(root, "mycobotgym_amd", "csrc", "model_gen.h")
    before = open(hdr).read()
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_model_header.py")], check=True, capture_output=True)
    assert open(hdr).read() == before


@pytest.mark.skipif(not os.path.isdir(REF_ASSETS), reason="reference tree not present (GPU box)")
def test_tables_reproducible_from_reference():
    from mycobotgym_amd.model.mjcf import MjcfCompiler, _to_jsonable
    fresh = _to_jsonable(MjcfCompiler(os.path.join(REF_ASSETS, "mycobot280.xml")).compile())
    stored = load_json("mycobot280")
    for k in ("body_mass", "body_pos", "body_ipos", "jnt_axis", "qpos0", "body_inertia"):
        assert np.allclose(np.asarray(fresh[k], dtype=float), np.asarray(stored[k], dtype=float), rtol=1e-12, atol=1e-15), k
    assert fresh["body_name"] == stored["body_name"] and fresh["eq"] == stored["eq"]
