"""Env-layer semantics of the oracle (restating mycobot.py) and the chaos measurement that shapes the parity tests."""
import numpy as np
import pytest

from tests.common import make_oracle


def test_config1_plumbing_one_env_1000_steps(built):
    """BASELINE configs[0]: 'MyCobotReach-Dense-IK-v0'-equivalent, 1 env, 1000 steps, seed 0: API shapes, TimeLimit."""
    from mycobotgym_amd.registry import spec
    kw = spec("MyCobotReach-Dense-IK-v0")
    ora = make_oracle(1, has_object=kw["has_object"], controller_type=kw["controller_type"], reward_type=kw["reward_type"], seed=0)
    obs, ag, dg = ora.reset(seed=0)
    assert obs.shape == (1, 10) and ag.shape == (1, 3) and dg.shape == (1, 3) and ora.act_dim == 7
    assert np.allclose(ag[0], [0.0138673, 0.01864658, 0.61236], atol=1e-8)        # gripper at the initial pose
    assert 0.21 <= dg[0, 2] <= 0.31 and abs(dg[0, 0]) <= 0.12 and abs(dg[0, 1]) <= 0.06
    rng = np.random.default_rng(0)
    lengths = []
    for t in range(1000):
        o = ora.step(rng.uniform(-1, 1, (1, 7)).astype(np.float32))
        assert np.isfinite(o["obs"]).all() and o["reward"][0] <= 0
        if o["truncated"][0]:
            lengths.append(int(o["ep_length"][0]))
    assert lengths and all(l <= 50 for l in lengths) and lengths.count(50) >= len(lengths) - 1   # TimeLimit(50)


def test_reset_distribution(built):
    """Reset parity with the reference is distributional (Appendix D-6): rectangle, rejection radius, z lift."""
    ora = make_oracle(4096, controller_type="joint", seed=3)
    _, ag, dg = ora.reset(seed=3)
    igx = ora.initial_gripper_xpos()
    assert np.all(np.abs(dg[:, 0]) <= 0.12) and np.all(np.abs(dg[:, 1]) <= 0.06)
    assert np.all(np.hypot(dg[:, 0] - igx[0], dg[:, 1] - igx[1]) >= 0.1)           # mycobot.py:232
    lifted = dg[:, 2] > 0.21
    assert 0.45 < lifted.mean() < 0.55 and dg[:, 2].max() <= 0.31 and np.all(dg[~lifted, 2] == 0.21)
    # reseeding reproduces, a different seed does not
    _, _, dg2 = ora.reset(seed=3); _, _, dg3 = ora.reset(seed=4)
    assert np.array_equal(dg, dg2) and not np.array_equal(dg, dg3)


def test_rewards_flags_and_autoreset(built):
    from oracle import pyoracle as po
    ora = make_oracle(8, controller_type="joint", reward_type="sparse", seed=0)
    ora.reset(seed=0)
    s = ora.get_state()
    o = ora.step(np.zeros((8, 7), np.float32))
    assert set(np.unique(o["reward"])) <= {-1.0, 0.0}                              # -(d > thr) as float32
    assert not o["terminated"].any() and np.array_equal(o["terminated"], o["is_success"])
    # put the goal on the gripper: success -> terminated == truncated == True (Appendix D-4), auto-reset
    s = ora.get_state(); s["goal"] = o["achieved"].copy(); ora.set_state(**s)
    o2 = ora.step(np.zeros((8, 7), np.float32))
    near = np.linalg.norm(o2["final_achieved"] - s["goal"], axis=1) < 0.01
    assert near.any()
    assert np.array_equal(o2["terminated"].astype(bool), near) and np.array_equal(o2["truncated"].astype(bool), near)
    assert np.all(o2["reward"][near] == 0.0)
    st = ora.get_state()
    assert np.all(st["elapsed"][near] == 0) and np.all(st["episode"][near] == 2)
    assert not np.array_equal(o2["desired"][near], s["goal"][near])                # fresh goal after the reset
    assert np.allclose(po.compute_reward(o2["final_achieved"], o2["final_desired"], 0, 0.01)[near], 0.0)


def test_joint_controller_overwrites_ctrl_and_ik_accumulates(built):
    """Appendix D-2 / D-3."""
    a = np.full((2, 7), 0.3, np.float32)
    j = make_oracle(2, controller_type="joint"); j.reset(seed=0); j.step(a); j.step(a)
    assert np.allclose(j.get_state()["ctrl"], 0.3, atol=1e-7)                      # not 0.6: do_simulation overwrites
    k = make_oracle(2, controller_type="IK"); k.reset(seed=0); k.step(a)
    c1 = k.get_state()["ctrl"].copy(); k.step(a); c2 = k.get_state()["ctrl"]
    assert np.allclose(c1[:, 6], 0.5 + 0.5 * 0.3, atol=1e-7) and not np.allclose(c1[:, :6], c2[:, :6])


def test_oracle_self_sensitivity():
    """The chaos measurement behind the parity-test design: the oracle against ITSELF from a state perturbed by
    1e-14 diverges by many orders of magnitude within a few env-steps (h*kv/M ~ 8 for the arm servos)."""
    n = 128
    A = make_oracle(n, controller_type="joint", seed=1); B = make_oracle(n, controller_type="joint", seed=1)
    A.reset(seed=1); B.reset(seed=1)
    s = B.get_state()
    s["qpos"] = s["qpos"] + 1e-14 * np.sign(np.random.default_rng(0).normal(size=s["qpos"].shape)); s["qpos_lag"] = s["qpos"]
    B.set_state(**s)
    rng = np.random.default_rng(42)
    med = []
    for t in range(8):
        a = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
        med.append(np.median(np.abs(A.step(a)["obs"] - B.step(a)["obs"]).max(axis=1)))
    assert med[0] < 1e-11 and med[-1] > 1e-6 and med[-1] / max(med[0], 1e-300) > 1e6
