"""GPU parity for the mocap controller (mycobot.py:172-189, mocap.xml:15-20): the arm has no servos and
hangs on a six-row weld between the mocap body and gripper_tcp; an env-step moves the mocap pose by
(0.1 a[:3], a[3:7] - xquat_tcp) from the welded body's (lagged) pose and takes 20 sub-steps.

Same method as tests/test_gpu_parity.py: sub-steps and env-steps from identical state (teacher-forced) against the
CPU oracle, whose weld is pinned by the reference's mocap keyframe (tests/test_oracle_known_answers.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def _actions(rng, n, dim, quat=True):
    a = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    if quat and dim == 8:        # orientations a policy would emit around the current one: identity-ish to far rotations
        a[: n // 2, 3:7] = (np.array([0.70710678, 0, 0, 0.70710678]) + 0.3 * rng.normal(size=(n // 2, 4))).astype(np.float32)
    return a


@pytest.mark.parametrize("fetch", [False, True])
def test_mocap_reset_and_dims(torch_cuda, fetch):
    from tests.common import make_pair
    envs, ora = make_pair(256, controller_type="mocap", fetch_env=fetch, seed=5)
    assert envs.action_dim == (4 if fetch else 8) == ora.act_dim         # mycobot.py:98-103
    obs, _ = envs.reset(seed=5)
    o_obs, o_ag, o_dg = ora.reset(seed=5)
    assert np.array_equal(obs["desired_goal"].cpu().numpy(), o_dg)
    assert np.abs(obs["observation"].cpu().numpy() - o_obs).max() < 1e-14
    envs.close()


def test_mocap_substeps_from_identical_state(torch_cuda):
    """frame_skip = 1: every sub-step of 30 env-steps compared from identical state."""
    from tests.common import make_pair, sync_oracle_to, step_errors
    n = 256
    envs, ora = make_pair(n, controller_type="mocap", reward_type="dense", seed=3, frame_skip=1, max_episode_steps=10 ** 9)
    envs.reset(seed=3); ora.reset(seed=3)
    rng = np.random.default_rng(7)
    worst_obs = worst_q = worst_v = 0.0
    for t in range(30):
        a = _actions(rng, n, 8)
        for s in range(20):
            sync_oracle_to(envs, ora)
            e, flags_equal, o = step_errors(envs, ora, a)
            assert flags_equal
            st, so = envs.get_state(), ora.get_state()
            worst_obs = max(worst_obs, e.max())
            worst_q = max(worst_q, np.abs(st["qpos"].cpu().numpy().T - so["qpos"]).max())
            worst_v = max(worst_v, np.abs(st["qvel"].cpu().numpy().T - so["qvel"]).max())
    print(f"\nmocap: 600 sub-steps x {n} envs from identical state: max err obs {worst_obs:.2e} qpos {worst_q:.2e} qvel {worst_v:.2e}")
    assert worst_obs < 3e-13 and worst_q < 1e-10 and worst_v < 4e-8      # measured 2.8e-15, 7.5e-13, 3.7e-10
    envs.close()


@pytest.mark.parametrize("fetch", [False, True])
def test_mocap_env_steps_from_identical_state(torch_cuda, fetch):
    from tests.common import make_pair, sync_oracle_to, step_errors
    n = 256
    envs, ora = make_pair(n, controller_type="mocap", fetch_env=fetch, reward_type="dense", seed=1)
    envs.reset(seed=1); ora.reset(seed=1)
    rng = np.random.default_rng(42)
    errs = []; bad_flags = 0
    for t in range(60):
        sync_oracle_to(envs, ora)
        e, flags_equal, o = step_errors(envs, ora, _actions(rng, n, envs.action_dim))
        bad_flags += (not flags_equal)
        errs.append(e)
    errs = np.concatenate(errs)
    print(f"\nmocap{' fetch' if fetch else ''}: one env-step from identical state, {errs.size} samples: "
          f"median {np.median(errs):.2e} p99 {np.quantile(errs, 0.99):.2e} max {errs.max():.2e}")
    assert bad_flags == 0
    assert np.median(errs) < 1e-13 and errs.max() < 2e-10                  # measured: median 9e-16, max 1.4e-12 (no stiff servos)
    envs.close()


def test_mocap_tracks_its_target(torch_cuda):
    """Physics sanity of the whole path: with a constant displacement command the gripper follows the mocap body."""
    import torch
    from mycobotgym_amd import MyCobotVecEnv
    n = 64
    envs = MyCobotVecEnv(n, has_object=False, controller_type="mocap", fetch_env=True, reward_type="dense")
    obs, _ = envs.reset(seed=0)
    p0 = obs["achieved_goal"].clone()
    a = torch.zeros(n, 4, device="cuda"); a[:, 2] = -0.3          # 3 cm down per step (0.1 * a)
    for _ in range(5): obs, *_ = envs.step(a)
    dz = (obs["achieved_goal"][:, 2] - p0[:, 2]).cpu().numpy()
    assert np.all(dz < -0.03) and np.all(dz > -0.16), dz[:4]
    assert torch.isfinite(obs["observation"]).all()
    envs.close()


def test_mocap_with_object_substeps(torch_cuda):
    """PickAndPlace + mocap: cube contacts and the weld in one sub-step pipeline."""
    from tests.common import make_pair, sync_oracle_to, step_errors
    n = 128
    envs, ora = make_pair(n, has_object=True, controller_type="mocap", reward_type="dense", seed=9, frame_skip=1,
                          max_episode_steps=10 ** 9)
    envs.reset(seed=9); ora.reset(seed=9)
    rng = np.random.default_rng(5)
    worst = 0.0
    for t in range(10):
        a = _actions(rng, n, 8)
        for s in range(20):
            sync_oracle_to(envs, ora)
            e, flags_equal, o = step_errors(envs, ora, a)
            assert flags_equal
            worst = max(worst, e.max())
    print(f"\nmocap + cube: 200 sub-steps x {n} envs from identical state: max obs err {worst:.2e}")
    assert worst < 1e-12                                                   # measured 9.6e-15
    envs.close()


def test_mocap_needs_its_model_variant(torch_cuda):
    import ctypes as C
    from mycobotgym_amd import _abi
    L = _abi.load()
    cfg = _abi.McgConfig(); cfg.n_envs = 4; cfg.controller = _abi.CTRL_MOCAP; cfg.frame_skip = 20; cfg.control_steps = 5
    cfg.max_episode_steps = 50; cfg.reward_type = 1
    model = _abi.McgModel(); assert L.mcg_default_model(0, C.byref(model)) == 0        # no weld in variant 0
    h = C.c_void_p()
    assert L.mcg_create(C.byref(cfg), C.byref(model), None, 0, 0, C.byref(h)) != 0
    assert b"mocap" in L.mcg_last_error()


def test_mocap_weld_with_mujoco_row_weights(torch_cuda):
    """`weld_rule="mujoco"`: the rotational inverse weight on the weld's rows 3-5 (mj_diagApprox as recalled) instead of one common
    weight (the variant the reference's keyframe supports, oracle/RULE_STUDY.md): same parity bar, and a visibly softer orientation."""
    from tests.common import make_pair, sync_oracle_to, step_errors
    n = 128
    envs, ora = make_pair(n, controller_type="mocap", reward_type="dense", seed=3, frame_skip=1, max_episode_steps=10 ** 9,
                          weld_rule="mujoco")
    ref, _ = make_pair(n, controller_type="mocap", reward_type="dense", seed=3, frame_skip=1, max_episode_steps=10 ** 9)
    envs.reset(seed=3); ora.reset(seed=3); ref.reset(seed=3)
    rng = np.random.default_rng(7)
    worst = 0.0; differs = 0.0
    for t in range(10):
        a = _actions(rng, n, 8)
        for s in range(20):
            sync_oracle_to(envs, ora)
            ref.set_state(**{k: v for k, v in envs.get_state().items()})
            e, flags_equal, o = step_errors(envs, ora, a)
            assert flags_equal
            worst = max(worst, e.max())
            import torch
            ob, *_ = ref.step(torch.as_tensor(a))
            differs = max(differs, float(np.abs(ob["observation"].cpu().numpy() - o["obs"]).max()))
    print(f"\nmocap, weld_rule=mujoco: 200 sub-steps x {n} envs from identical state: max obs err {worst:.2e}; "
          f"against the common-weight weld the same sub-steps differ by up to {differs:.2e}")
    assert worst < 3e-13 and differs > 1e-9
    envs.close(); ref.close()
