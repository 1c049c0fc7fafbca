"""Golden fixtures: the reference's data-file known answers, and regression vectors of this repo's oracle."""
import json
import os

import numpy as np

from tests.common import load_json, make_oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_reference_known_answers_fixture(built):
    from mycobotgym_amd.model.mjcf import _np_model
    from mycobotgym_amd.model.refdyn import kinematics
    from mycobotgym_amd.registry import REGISTRY
    from oracle import pyoracle as po
    ka = json.load(open(os.path.join(GOLD, "reference_known_answers.json")))
    tab = load_json("mycobot280"); m = _np_model(tab)
    s = m["site_name"].index("EEF")
    d = po.OracleData(po.OracleModel(tab))
    for name, q in (("eef_at_qpos0", m["qpos0"]), ("eef_at_joint_keyframe", np.asarray(tab["keys"][0]["qpos"]))):
        want, atol = ka[name]["value"], ka[name]["atol"]
        assert np.allclose(kinematics(m, q)["site_xpos"][s], want, atol=atol)              # product-side numpy FK
        d.set_state(qpos=q); d.forward()
        assert np.allclose(d.get("site_xpos", (8, 3))[s], want, atol=atol)                # oracle FK
    assert np.allclose(tab["keys"][0]["qpos"][:6], ka["joint_keyframe_qpos"]["value"])
    for k, v in ka["dimensions"]["value"].items():
        assert (len(tab["excludes"]) if k == "nexclude" else tab[k]) == v
    for link, mass in ka["inertial_masses"]["value"].items():
        assert tab["body_mass"][tab["body_name"].index(link)] == mass
    assert sum(k.endswith("-v0") for k in REGISTRY) == ka["registered_ids"]["value"]["v0"]
    assert sum(k.endswith("-v1") for k in REGISTRY) == ka["registered_ids"]["value"]["v1"]


def test_oracle_regression_vectors(built):
    """One env-step from reset is far inside the predictability horizon, so these replay to ~1e-9 on any x86 host."""
    gold = json.load(open(os.path.join(GOLD, "oracle_regression.json")))
    for key, (has_object, controller) in {"reach_joint": (False, "joint"), "reach_ik": (False, "IK"), "pnp_joint": (True, "joint"),
                                           "reach_mocap": (False, "mocap")}.items():
        g = gold[key]
        ora = make_oracle(4, has_object=has_object, controller_type=controller, reward_type="dense", seed=2024, n_threads=1)
        obs, ag, dg = ora.reset(seed=2024)
        assert np.array_equal(dg, np.asarray(g["reset_goal"])) and np.allclose(obs, g["reset_obs"], atol=1e-13)
        o = ora.step(np.asarray(g["steps"][0]["action"], dtype=np.float32))
        assert np.abs(o["obs"] - np.asarray(g["steps"][0]["obs"])).max() < 1e-6
        assert np.abs(o["reward"] - np.asarray(g["steps"][0]["reward"])).max() < 1e-6
