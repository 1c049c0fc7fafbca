"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerances: observations / rewards within 1e-4 over 100 steps from identical state (BASELINE.json north_star);
one step from identical state within 1e-9; flags, episode lengths and reset draws bit-exact.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_100_STEPS = 1e-4     # north_star tolerance
TOL_ONE_STEP = 1e-9


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


@pytest.mark.parametrize("controller", ["joint", "IK"])
def test_reset_bit_exact(torch_cuda, controller):
    from tests.common import make_pair
    envs, ora = make_pair(512, controller_type=controller, seed=123)
    obs, info = envs.reset(seed=123)
    o_obs, o_ag, o_dg = ora.reset(seed=123)
    assert info == {}
    assert np.array_equal(obs["desired_goal"].cpu().numpy(), o_dg)          # Philox draws: bit-exact
    assert np.abs(obs["observation"].cpu().numpy() - o_obs).max() < 1e-14
    assert np.abs(obs["achieved_goal"].cpu().numpy() - o_ag).max() < 1e-14
    s, so = envs.get_state(), ora.get_state()
    assert np.array_equal(s["episode"].cpu().numpy(), so["episode"])
    envs.close()


@pytest.mark.parametrize("controller,reward", [("joint", "dense"), ("joint", "sparse"), ("IK", "dense")])
def test_100_steps_from_identical_state(torch_cuda, controller, reward):
    from tests.common import make_pair, compare_step
    n = 256
    envs, ora = make_pair(n, controller_type=controller, reward_type=reward, seed=1)
    envs.reset(seed=1); ora.reset(seed=1)
    rng = np.random.default_rng(42)
    worst = 0.0
    for t in range(100):
        a = rng.uniform(-1, 1, (n, envs.action_dim)).astype(np.float32)
        worst = max(worst, compare_step(envs, ora, a))
    print(f"\n[{controller}/{reward}] max |hip - oracle| over 100 steps x {n} envs = {worst:.3e}")
    assert worst < TOL_100_STEPS
    envs.close()


@pytest.mark.parametrize("controller", ["joint", "IK"])
def test_one_step_from_random_states(torch_cuda, controller):
    """Scatter the oracle over random (but physical) states, copy them across, compare one step."""
    from tests.common import make_pair, compare_step, sync_oracle_to
    n = 512
    envs, ora = make_pair(n, controller_type=controller, reward_type="dense", seed=3)
    envs.reset(seed=3); ora.reset(seed=3)
    rng = np.random.default_rng(5)
    for _ in range(5):      # drive the oracle somewhere interesting
        ora.step(rng.uniform(-1, 1, (n, ora.act_dim)).astype(np.float32))
    sync_oracle_to(envs, ora)
    a = rng.uniform(-1, 1, (n, envs.action_dim)).astype(np.float32)
    err = compare_step(envs, ora, a)
    print(f"\n[{controller}] one-step max err = {err:.3e}")
    assert err < TOL_ONE_STEP
    s, so = envs.get_state(), ora.get_state()
    assert np.abs(s["qpos"].cpu().numpy().T - so["qpos"]).max() < TOL_ONE_STEP
    assert np.abs(s["qvel"].cpu().numpy().T - so["qvel"]).max() < 1e-7
    envs.close()


def test_compute_reward_batched(torch_cuda):
    import torch
    from mycobotgym_amd import MyCobotVecEnv
    from oracle import pyoracle as po
    rng = np.random.default_rng(0)
    ag = rng.normal(size=(1000, 3)) * 0.02; dg = rng.normal(size=(1000, 3)) * 0.02
    for rt, code in (("sparse", 0), ("dense", 1)):
        envs = MyCobotVecEnv(4, has_object=False, controller_type="joint", reward_type=rt)
        r = envs.compute_reward(torch.as_tensor(ag), torch.as_tensor(dg), {})
        ref = po.compute_reward(ag, dg, code, 0.01)
        assert np.array_equal(r.cpu().numpy().astype(np.float64), ref) or np.abs(r.cpu().numpy() - ref).max() < 1e-15
        assert r.dtype == (torch.float32 if rt == "sparse" else torch.float64)     # mycobot.py:293,295
        envs.close()
