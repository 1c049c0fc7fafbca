"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on identical seeded inputs.

The reference's actuator gains make its explicit-Euler step an unstable map (h*kv/M ~ 8: every unsaturated
sub-step multiplies a velocity error by ~ -7; saturation bounds it).  The CPU oracle run against itself from a
state perturbed by 1e-14 diverges to 1e-4 within ONE env-step in the worst of 256 envs and to 1e-2 within ten
(tests/test_oracle_invariants.py::test_oracle_self_sensitivity).  No two implementations that round differently
can therefore satisfy "1e-4 over 100 free-running steps" on this model; parity is established instead by

  * bit-exact reset draws, flags, counters;
  * every one of the 2000 physics sub-steps of a 100-step rollout compared from identical state (teacher-forced)
    at 1e-12 (positions);
  * env-steps from identical state: joint (20 sub-steps) 1e-8 in the worst env; IK (100 sub-steps) by quantiles, tied to the
    oracle's own sensitivity in tests/test_gpu_fullsize_parity.py;
  * free-running divergence no faster than the oracle's own divergence from a 1-ulp-perturbed copy;
  * the literal criterion -- 1e-4 over 100 free-running steps -- on a contractive variant of the model
    (actuator velocity gains x0.1), where it is attainable.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_SUBSTEP = 1e-12     # measured on MI355X: obs 6e-16, qpos 6e-15 (bounds are <= 100x what is measured, per quantity)
TOL_100_STEPS = 1e-4     # north_star tolerance


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


def test_every_substep_of_a_100_step_rollout(torch_cuda):
    """frame_skip=1 engines, state re-synchronised before every sub-step: 100 env-steps x 20 sub-steps, the action
    changing every 20.  Compares observation, qpos, qvel and warm-start after each sub-step."""
    from tests.common import make_pair, sync_oracle_to, step_errors
    n = 256
    kw = dict(controller_type="joint", reward_type="dense", seed=11, frame_skip=1, max_episode_steps=10 ** 9)
    envs, ora = make_pair(n, **kw)
    envs.reset(seed=11); ora.reset(seed=11)
    rng = np.random.default_rng(2)
    worst_obs = worst_q = worst_v = worst_w = 0.0
    for t in range(100):
        a = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
        for s in range(20):
            sync_oracle_to(envs, ora)
            e, flags_equal, o = step_errors(envs, ora, a)
            assert flags_equal
            st, so = envs.get_state(), ora.get_state()
            worst_obs = max(worst_obs, e.max())
            worst_q = max(worst_q, np.abs(st["qpos"].cpu().numpy().T - so["qpos"]).max())
            worst_v = max(worst_v, np.abs(st["qvel"].cpu().numpy().T - so["qvel"]).max())
            worst_w = max(worst_w, np.abs(st["warm"].cpu().numpy().T - so["warm"]).max())
    print(f"\n2000 sub-steps x {n} envs: max err obs {worst_obs:.2e} qpos {worst_q:.2e} qvel {worst_v:.2e} qacc {worst_w:.2e}")
    assert worst_obs < 1e-13 and worst_q < TOL_SUBSTEP             # measured 6.1e-16, 6.2e-15
    assert worst_v < 3e-10 and worst_w < 4e-8      # measured 3.1e-12, 4.1e-10 (qvel = h * qacc; accelerations reach 1e4 rad/s^2)
    envs.close()


@pytest.mark.parametrize("controller", ["joint", "IK"])
def test_env_steps_from_identical_state(torch_cuda, controller):
    """100 env-steps, state re-synchronised before each: the per-step error distribution.  The IK controller's 100 sub-steps of stiff
    servos amplify a last-bit difference chaotically: its quantiles are held to the oracle's OWN sensitivity -- a
    twin oracle started 1e-14 away -- not to an absolute number."""
    from tests.common import make_pair, make_oracle, sync_oracle_to, step_errors, twin_errors, assert_within_oracle_sensitivity
    n = 256
    envs, ora = make_pair(n, controller_type=controller, reward_type="dense", seed=1)
    twin = make_oracle(n, controller_type=controller, reward_type="dense", seed=1) if controller == "IK" else None
    envs.reset(seed=1); ora.reset(seed=1)
    if twin: twin.reset(seed=1)
    rng = np.random.default_rng(42); prng = np.random.default_rng(7)
    errs, terrs = [], []
    mismatched_flags = 0
    for t in range(100):
        sync_oracle_to(envs, ora)
        state = ora.get_state()
        a = rng.uniform(-1, 1, (n, envs.action_dim)).astype(np.float32)
        e, flags_equal, o = step_errors(envs, ora, a)
        mismatched_flags += (not flags_equal)
        errs.append(e)
        if twin: terrs.append(twin_errors(twin, state, a, o, prng))
    if twin: assert_within_oracle_sensitivity(errs, terrs, "[Reach IK env-step]")
    errs = np.concatenate(errs)
    q50, q99, mx = np.median(errs), np.quantile(errs, 0.99), errs.max()
    print(f"\n[{controller}] one env-step from identical state, {errs.size} samples: median {q50:.2e} p99 {q99:.2e} max {mx:.2e}")
    assert mismatched_flags == 0
    if controller == "joint":       # measured: median 3.5e-16, p99 2.0e-13, max 8.6e-11
        assert q50 < 1e-13 and q99 < 2e-11 and mx < 1e-8
    envs.close()


def test_free_running_divergence_is_the_oracles_own(torch_cuda):
    """HIP-vs-oracle free-running error must not grow faster than oracle-vs-(oracle + 1e-14)."""
    from tests.common import make_pair, make_oracle, step_errors
    n = 256
    envs, ora = make_pair(n, controller_type="joint", reward_type="dense", seed=1)
    twin = make_oracle(n, controller_type="joint", reward_type="dense", seed=1)
    envs.reset(seed=1); ora.reset(seed=1); twin.reset(seed=1)
    s = twin.get_state()
    s["qpos"] = s["qpos"] + 1e-14 * np.sign(np.random.default_rng(0).normal(size=s["qpos"].shape))
    s["qpos_lag"] = s["qpos"]
    twin.set_state(**s)
    rng = np.random.default_rng(42)
    med_hip, med_twin = [], []
    for t in range(12):
        a = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
        e, _, o = step_errors(envs, ora, a)
        ot = twin.step(a)
        med_hip.append(np.median(e)); med_twin.append(np.median(np.abs(ot["obs"] - o["obs"]).max(axis=1)))
    print("\nmedian err per step  hip-vs-oracle:", " ".join(f"{x:.1e}" for x in med_hip))
    print("                  oracle-vs-oracle+1e-14:", " ".join(f"{x:.1e}" for x in med_twin))
    for a_, b_ in zip(med_hip, med_twin):
        assert a_ < 100 * b_ + 1e-13
    envs.close()


@pytest.mark.parametrize("controller", ["joint", "IK"])
def test_100_free_running_steps_contractive_model(torch_cuda, controller):
    """The north_star criterion, literally, where the dynamics allow it: actuator velocity gains x0.1."""
    from tests.common import make_pair, compare_step, load_json, soften_gains
    n = 256
    tab = soften_gains(load_json("mycobot280_reach"))
    envs, ora = make_pair(n, table=tab, controller_type=controller, reward_type="dense", seed=1)
    envs.reset(seed=1); ora.reset(seed=1)
    rng = np.random.default_rng(42)
    worst = 0.0
    for t in range(100):
        a = rng.uniform(-1, 1, (n, envs.action_dim)).astype(np.float32)
        worst = max(worst, compare_step(envs, ora, a))
    print(f"\n[{controller}, kv x0.1] max |hip - oracle| over 100 free-running steps x {n} envs = {worst:.3e}")
    assert worst < TOL_100_STEPS
    assert worst < (1e-6 if controller == "joint" else 1e-12)      # measured 6.5e-9 / 5.8e-15
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
        assert np.abs(r.cpu().numpy() - ref).max() < 1e-15
        assert r.dtype == (torch.float32 if rt == "sparse" else torch.float64)     # mycobot.py:293,295
        envs.close()
