"""GPU: size-independent properties at BASELINE's full size (8192 envs) and API behaviour of the HIP engine."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N_FULL = 8192


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    assert torch.cuda.is_available()
    return torch


def _roll(envs, torch, steps, seed=0):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    outs = []
    for t in range(steps):
        a = torch.rand(envs.num_envs, envs.action_dim, device="cuda", generator=g) * 2 - 1
        obs, rew, term, trunc, info = envs.step(a)
        outs.append((obs["observation"].clone(), rew.clone(), trunc.clone(), obs["desired_goal"].clone()))
    return outs


@pytest.mark.parametrize("controller", ["joint", "IK"])
def test_full_size_determinism_and_sanity(torch_cuda, controller):
    """Two engines, same seed, 8192 envs, 60 steps (crosses the TimeLimit reset): bit-identical; values sane."""
    torch = torch_cuda
    from mycobotgym_amd import MyCobotVecEnv
    runs = []
    for rep in range(2):
        envs = MyCobotVecEnv(N_FULL, has_object=False, controller_type=controller, reward_type="dense", seed=11)
        envs.reset(seed=11)
        runs.append(_roll(envs, torch, 60 if controller == "joint" else 52))
        envs.close()
    for (o1, r1, t1, g1), (o2, r2, t2, g2) in zip(*runs):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(t1, t2) and torch.equal(g1, g2)
    obs, rew, trunc, goal = runs[0][-1]
    assert torch.isfinite(obs).all() and (rew <= 0).all()
    assert (obs[:, :3].abs() < 1.0).all()                         # the gripper stays within the arm's reach
    assert runs[0][49][2].all()                                   # TimeLimit(50): every env truncates at step 50
    assert (goal[:, 0].abs() <= 0.12).all() and (goal[:, 1].abs() <= 0.06).all()


def test_shard_invariance(torch_cuda):
    """Env i of an 8192-env engine == env 0.. of a small engine created with env_id_offset = i (global-id RNG keys,
    no cross-env coupling): the multi-GPU sharding rule, checked on one GPU."""
    torch = torch_cuda
    from mycobotgym_amd import MyCobotVecEnv
    big = MyCobotVecEnv(N_FULL, has_object=False, controller_type="joint", reward_type="dense", seed=3)
    off, n = 5000, 192
    small = MyCobotVecEnv(n, has_object=False, controller_type="joint", reward_type="dense", seed=3, env_id_offset=off)
    ob, _ = big.reset(seed=3); os_, _ = small.reset(seed=3)
    assert torch.equal(ob["desired_goal"][off:off + n], os_["desired_goal"])
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for t in range(55):
        a = torch.rand(N_FULL, 7, device="cuda", generator=g) * 2 - 1
        ob, rb, _, tb, _ = big.step(a); os_, rs, _, ts, _ = small.step(a[off:off + n].contiguous())
        assert torch.equal(ob["observation"][off:off + n], os_["observation"]) and torch.equal(rb[off:off + n], rs)
        assert torch.equal(ob["desired_goal"][off:off + n], os_["desired_goal"]) and torch.equal(tb[off:off + n], ts)
    big.close(); small.close()


def test_state_roundtrip_and_api(torch_cuda):
    torch = torch_cuda
    from mycobotgym_amd import make
    envs = make("MyCobotReach-Sparse-joint-v0", num_envs=300)
    with pytest.raises(RuntimeError, match="before calling env.reset"):
        envs.step(torch.zeros(300, 7))
    obs, info = envs.reset(seed=0)
    assert set(obs) == {"observation", "achieved_goal", "desired_goal"} and info == {}
    assert obs["observation"].shape == (300, 10) and obs["observation"].dtype == torch.float64
    assert envs.single_action_space.shape == (7,) and envs.action_space.shape == (300, 7)
    with pytest.raises(ValueError):
        envs.step(torch.zeros(300, 6))
    obs, rew, term, trunc, info = envs.step(np.zeros((300, 7), np.float32))       # numpy actions are accepted
    assert rew.shape == (300,) and term.dtype == torch.bool and set(torch.unique(rew).tolist()) <= {-1.0, 0.0}
    assert {"is_success", "final_observation", "_final_observation", "episode"} <= set(info)
    s = envs.get_state()
    assert s["qpos"].shape == (12, 300) and s["elapsed"].eq(1).all()
    s2 = {k: v.clone() for k, v in s.items()}; s2["qpos"] += 0.01
    envs.set_state(**s2)
    assert torch.equal(envs.get_state()["qpos"], s2["qpos"])
    envs.load_state_dict(s)
    assert all(torch.equal(envs.state_dict()[k], s[k]) for k in s)
    # actions outside [-1, 1] are clipped (mycobot.py:133)
    envs.load_state_dict(s); o1 = envs.step(torch.full((300, 7), 5.0))[0]["observation"].clone()
    envs.load_state_dict(s); o2 = envs.step(torch.full((300, 7), 1.0))[0]["observation"].clone()
    assert torch.equal(o1, o2)
    envs.close()


def test_masked_reset(torch_cuda):
    torch = torch_cuda
    from mycobotgym_amd import MyCobotVecEnv
    envs = MyCobotVecEnv(128, has_object=False, controller_type="joint", reward_type="dense", seed=2)
    envs.reset(seed=2)
    for _ in range(3):
        envs.step(torch.rand(128, 7, device="cuda") * 2 - 1)
    before = envs.get_state()
    mask = torch.zeros(128, dtype=torch.bool, device="cuda"); mask[::4] = True
    envs.reset(mask=mask)
    after = envs.get_state()
    assert after["elapsed"][mask].eq(0).all() and after["elapsed"][~mask].eq(3).all()
    assert torch.equal(after["qpos"][:, ~mask], before["qpos"][:, ~mask]) and after["qpos"][:, mask].eq(0).all()
    assert after["episode"][mask].eq(2).all() and after["episode"][~mask].eq(1).all()
    envs.close()


def test_exact_mesh_variant_and_fetch(torch_cuda):
    """Model variants share the binary: exact mesh inertia; fetch keyframe with the fixed target quaternion."""
    from tests.common import make_pair, make_oracle, sync_oracle_to, step_errors, twin_errors, assert_within_oracle_sensitivity
    rng = np.random.default_rng(0); prng = np.random.default_rng(1)
    for kw in (dict(controller_type="joint", mesh_inertia="exact"), dict(controller_type="IK", fetch_env=True)):
        envs, ora = make_pair(128, reward_type="dense", seed=4, **kw)
        ik = kw["controller_type"] == "IK"
        twin = make_oracle(128, reward_type="dense", seed=4, **kw) if ik else None
        o_hip, _ = envs.reset(seed=4); o_ora = ora.reset(seed=4)
        if twin: twin.reset(seed=4)
        assert np.abs(o_hip["observation"].cpu().numpy() - o_ora[0]).max() < 1e-12
        assert envs.action_dim == (4 if kw.get("fetch_env") else 7)
        errs, terrs = [], []
        for t in range(10):
            sync_oracle_to(envs, ora)
            state = ora.get_state()
            a = rng.uniform(-1, 1, (128, envs.action_dim)).astype(np.float32)
            e, flags, o = step_errors(envs, ora, a)
            assert flags; errs.append(e)
            if twin: terrs.append(twin_errors(twin, state, a, o, prng))
        if ik: assert_within_oracle_sensitivity(errs, terrs, "[fetch IK env-step]")      # (the fetch keyframe is a near-singular wrist pose: the oracle's own sensitivity is large there)
        else: assert np.concatenate(errs).max() < 1e-8
        envs.close()


def test_block_gripper(torch_cuda):
    """S7 `_step_callback` (mycobot.py:300-306): finger joints forced to zero after every step, observations un-lagged."""
    from tests.common import make_pair, sync_oracle_to, step_errors
    rng = np.random.default_rng(0)
    for has_object in (False, True):
        envs, ora = make_pair(64, has_object=has_object, controller_type="joint", reward_type="dense", seed=6, block_gripper=True)
        envs.reset(seed=6); ora.reset(seed=6)
        errs = []
        for t in range(8):
            sync_oracle_to(envs, ora)
            e, flags, _ = step_errors(envs, ora, rng.uniform(-1, 1, (64, 7)).astype(np.float32))
            assert flags; errs.append(e)
            q = envs.get_state()["qpos"]
            assert (q[7] == 0).all() and (q[9] == 0).all()
        assert np.concatenate(errs).max() < 1e-8
        envs.close()


def test_bad_state_recovers(torch_cuda):
    """A NaN / huge state is reset like mj_checkPos does (qpos0, zero velocities) instead of poisoning the env forever."""
    torch = torch_cuda
    from mycobotgym_amd import MyCobotVecEnv
    for has_object in (False, True):
        envs = MyCobotVecEnv(64, has_object=has_object, controller_type="joint", reward_type="dense", seed=1)
        envs.reset(seed=1)
        s = envs.get_state()
        s["qvel"][0, 3] = float("nan"); s["qpos"][2, 5] = 1e300
        if has_object:
            s["qvel"][14, 7] = float("inf")
        envs.set_state(**s)
        for _ in range(3):
            obs, rew, term, trunc, info = envs.step(torch.zeros(64, 7, device="cuda"))
        assert torch.isfinite(obs["observation"]).all() and torch.isfinite(rew).all()
        assert all(torch.isfinite(v.double()).all() for v in envs.get_state().values())
        envs.close()


def test_every_v0_id_constructs_and_steps(torch_cuda):
    """The 30 `-v0` registrations of the reference (mycobotgym/__init__.py:26-45): every one builds an engine and steps (the five
    Reach-RewardShaping ids keep the cube as a hidden free body, mycobot.py:475-481)."""
    torch = torch_cuda
    from mycobotgym_amd import make
    from mycobotgym_amd.registry import REGISTRY
    ids = sorted(k for k in REGISTRY if k.endswith("-v0"))
    assert len(ids) == 30
    built_ids = 0
    for env_id in ids:
        built_ids += 1
        envs = make(env_id, num_envs=16)
        obs, _ = envs.reset(seed=1)
        a = torch.rand(16, envs.action_dim, device="cuda") * 2 - 1
        for _ in range(3):
            obs, rew, term, trunc, info = envs.step(a)
        assert torch.isfinite(obs["observation"]).all() and torch.isfinite(rew).all(), env_id
        want = {"mocap": 8, "IK": 7, "joint": 7}[REGISTRY[env_id]["controller_type"]]
        if REGISTRY[env_id]["fetch_env"]:
            want = 4
        assert envs.action_dim == want, env_id
        assert obs["observation"].shape == (16, 25 if REGISTRY[env_id]["has_object"] else 10)
        envs.close()
    assert built_ids == 30


@pytest.mark.parametrize("controller", ["joint", "IK", "mocap"])
def test_two_wave_and_one_wave_kernels_agree(torch_cuda, controller, monkeypatch):
    """Reach grids of at most one workgroup per CU run the two-wave kernels (a helper wave computes M and the Euler
    factor); larger grids, or MCG_NO_SPLIT=1 at construction, the one-wave kernels.  Same
    mathematics (a' = a - h (M+hB)^-1 B a  ==  (M+hB)^-1 M a): one env-step from identical state must agree."""
    torch = torch_cuda
    from mycobotgym_amd import MyCobotVecEnv
    n = 512
    a_env = MyCobotVecEnv(n, has_object=False, controller_type=controller, reward_type="dense", seed=4)
    monkeypatch.setenv("MCG_NO_SPLIT", "1")
    b_env = MyCobotVecEnv(n, has_object=False, controller_type=controller, reward_type="dense", seed=4)
    monkeypatch.delenv("MCG_NO_SPLIT")
    oa, _ = a_env.reset(seed=4); ob, _ = b_env.reset(seed=4)
    assert torch.equal(oa["observation"], ob["observation"])
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    from tests.common import make_oracle, assert_within_oracle_sensitivity
    errs, terrs = [], []
    ora = twin = None
    if controller == "IK":
        ora = make_oracle(n, has_object=False, controller_type=controller, reward_type="dense", seed=4); ora.reset(seed=4)
        twin = make_oracle(n, has_object=False, controller_type=controller, reward_type="dense", seed=4); twin.reset(seed=4)
    prng = np.random.default_rng(3)
    for t in range(20):
        act = torch.rand(n, a_env.action_dim, device="cuda", generator=g) * 2 - 1
        b_env.set_state(**{k: v for k, v in a_env.get_state().items()})
        if ora is not None:      # the oracle's sensitivity on this very state and action
            st = {k: v.cpu().numpy() for k, v in a_env.get_state().items()}
            s0 = dict(qpos=st["qpos"].T.copy(), qvel=st["qvel"].T.copy(), ctrl=st["ctrl"].T.copy(), warm=st["warm"].T.copy(), qpos_lag=st["qpos_lag"].T.copy(),
                      goal=st["goal"].T.copy(), elapsed=st["elapsed"].astype(np.int32), episode=st["episode"].astype(np.int32))
            ora.set_state(**s0); o_ref = ora.step(act.cpu().numpy())
            s1 = dict(s0); s1["qpos"] = s0["qpos"] + 1e-14 * np.sign(prng.normal(size=s0["qpos"].shape))
            twin.set_state(**s1); o_tw = twin.step(act.cpu().numpy())
            terrs.append(np.abs(o_tw["obs"] - o_ref["obs"]).max(axis=1))
        oa, ra, *_ = a_env.step(act); ob, rb, *_ = b_env.step(act)
        errs.append(torch.maximum((oa["observation"] - ob["observation"]).abs().amax(dim=1), (ra - rb).abs()).cpu().numpy())
    errs = np.concatenate(errs)
    print(f"\n[{controller}] two-wave vs one-wave kernels, one env-step from identical state: median {np.median(errs):.2e} "
          f"p99 {np.quantile(errs, 0.99):.2e} max {errs.max():.2e}")
    # the servo-driven controllers are chaotic: the two kernels differ from each other no more than the oracle does from
    # a twin of itself started 1e-14 away on the same states and actions (no absolute number)
    if controller == "IK": assert_within_oracle_sensitivity([errs], [np.concatenate(terrs)], "[Reach IK two-wave vs one-wave]")
    else: assert errs.max() < 1e-8
    # a grid beyond one workgroup per CU takes the one-wave path by itself
    big = MyCobotVecEnv(64 * 300, has_object=False, controller_type=controller, reward_type="dense")
    o, _ = big.reset(seed=1)
    o, r, *_ = big.step(torch.zeros(64 * 300, big.action_dim, device="cuda"))
    assert torch.isfinite(o["observation"]).all()
    a_env.close(); b_env.close(); big.close()


@pytest.mark.parametrize("controller", ["joint", "mocap"])
def test_two_cooperative_routings_agree(torch_cuda, controller, monkeypatch):
    """PickAndPlace, two different solvers for the same sub-step.  By default the cooperative phase solves TWO flagged environments per wave
    (mcg_coop.hpp: coop_solve_pair -- assembly on the matrix cores, L D L^T in one 16-lane DPP row with rows 16 / 17 carried transposed);
    with MCG_COOP_PAIR=0 it solves one per wave with the first implementation (coop_solve: LDS-window assembly on the VALU, 18-lane
    factorisation by v_readlane).  The minimiser is unique: one env-step from identical state must agree -- scripted grasp for the joint
    controller (cube between the pads), random mocap motion (pads and links on the table) for mocap."""
    torch = torch_cuda
    from mycobotgym_amd import MyCobotVecEnv
    n = 256
    kw = dict(has_object=True, controller_type=controller, reward_type="reward_shaping", seed=6, max_episode_steps=10 ** 9)
    a_env = MyCobotVecEnv(n, **kw)
    monkeypatch.setenv("MCG_COOP_PAIR", "0")
    b_env = MyCobotVecEnv(n, **kw)
    monkeypatch.delenv("MCG_COOP_PAIR")
    a_env.reset(seed=6); b_env.reset(seed=6)
    a_env.counters(clear=True)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    act = None
    if controller == "joint":
        from mycobotgym_amd.scenarios import grasp_state
        st = grasp_state(n, seed=3)
        act = torch.as_tensor(st.pop("action"), device="cuda")
        a_env.set_state(**st)
    errs = []; touched = 0
    for t in range(30):
        a = act if act is not None else torch.rand(n, a_env.action_dim, device="cuda", generator=g) * 2 - 1
        b_env.set_state(**{k: v for k, v in a_env.get_state().items()})
        oa, ra, *_ = a_env.step(a); ob, rb, *_ = b_env.step(a)
        errs.append(torch.maximum((oa["observation"] - ob["observation"]).abs().amax(dim=1), (ra - rb).abs() * 1e-2).cpu().numpy())
        touched += int((ra >= 50.0).sum())          # grasp / lift stage of the shaped reward: both pads on the cube
    errs = np.concatenate(errs)
    c = a_env.counters()
    print(f"\n[{controller}] PickAndPlace two environments per wave vs one per wave, one env-step from identical state: median {np.median(errs):.2e} "
          f"p99 {np.quantile(errs, 0.99):.2e} max {errs.max():.2e}; env-steps with both pads on the cube: {touched}; coupled env-sub-steps {c['coupled_env_substeps']}")
    assert errs.max() < 1e-8 and c["coupled_env_substeps"] > 0
    a_env.close(); b_env.close()


@pytest.mark.parametrize("has_object,n", [(False, 100), (False, 1), (True, 50), (True, 33)])
def test_ragged_env_counts_with_multi_wave_kernels(torch_cuda, has_object, n):
    """Env counts that leave the last workgroup partly (or almost entirely) empty: the lanes beyond N leave in every wave
    of the multi-wave kernels alike, so the workgroup barriers stay matched; results still match the oracle."""
    from tests.common import make_pair, sync_oracle_to, step_errors
    envs, ora = make_pair(n, has_object=has_object, controller_type="joint", reward_type="dense", seed=2)
    envs.reset(seed=2); ora.reset(seed=2)
    rng = np.random.default_rng(1)
    for t in range(4):
        sync_oracle_to(envs, ora)
        e, flags_equal, _ = step_errors(envs, ora, rng.uniform(-1, 1, (n, 7)).astype(np.float32))
        assert flags_equal and e.max() < 1e-8
    envs.close()


def test_ragged_workgroup_with_flagged_lanes_equals_the_full_one(torch_cuda):
    """in a ragged last workgroup (n % 32 != 0) the surplus lanes shadow the last environment; they must not be handed
    out to the cooperative solves, counted, or stored.  Scripted-grasp states (EVERY lane flagged, more than seven per workgroup): an
    engine of 40 environments must reproduce, bit for bit, the first 40 of an engine of 64 over whole env-steps, and its coupled
    env-sub-step counter must count 40 lanes per sub-step, not 64."""
    torch = torch_cuda
    from mycobotgym_amd import MyCobotVecEnv
    from mycobotgym_amd.scenarios import grasp_state
    st = grasp_state(64, seed=0)
    act = torch.as_tensor(st.pop("action"), device="cuda")
    outs = []
    for n in (64, 40):
        envs = MyCobotVecEnv(n, has_object=True, controller_type="joint", reward_type="dense", seed=3, max_episode_steps=10 ** 9)
        envs.reset(seed=3)
        envs.set_state(**{k: (torch.as_tensor(v)[..., :n] if torch.as_tensor(v).ndim and torch.as_tensor(v).shape[-1] == 64 else v) for k, v in st.items()})
        envs.counters(clear=True)
        for t in range(6): obs, *_ = envs.step(act[:n])
        s = envs.get_state()
        outs.append((obs["observation"].clone(), s["qpos"].clone(), s["qvel"].clone(), envs.counters()))
        envs.close()
    (o64, q64, v64, c64), (o40, q40, v40, c40) = outs
    assert torch.equal(o64[:40], o40) and torch.equal(q64[:, :40], q40) and torch.equal(v64[:, :40], v40)
    print(f"\nragged workgroup: counters of 64 envs {c64}, of 40 envs {c40}")
    # 90 % of the env-sub-steps are coupled here: 24 shadow lanes counted along would push the 40-env count (4318 measured) past its ceiling
    assert 0.8 * 40 * 6 * 20 < c40["coupled_env_substeps"] <= 40 * 6 * 20 and c64["coupled_env_substeps"] <= 64 * 6 * 20


def test_long_random_rollouts_stay_finite(torch_cuda):
    """Every task / controller / fetch combination, 2048 envs, 150 random env-steps (three episodes, auto-resets, pad contacts
    in the fetch PickAndPlace starts): nothing may go non-finite, hang or trip the bad-state guard into a reset storm."""
    torch = torch_cuda
    from mycobotgym_amd import MyCobotVecEnv
    n = 2048
    for obj in (False, True):
        for ctrl in ("joint", "IK", "mocap"):
            for fetch in ((False,) if ctrl == "joint" else (False, True)):
                envs = MyCobotVecEnv(n, has_object=obj, controller_type=ctrl, fetch_env=fetch,
                                     reward_type="reward_shaping" if obj else "dense",
                                     domain_randomization={"mass": (0.5, 2.0), "friction": (0.5, 1.5)} if obj else None)
                envs.reset(seed=0)
                g = torch.Generator(device="cuda"); g.manual_seed(0)
                steps = 150 if ctrl != "IK" else 60
                len_sum = 0.0; len_cnt = 0
                for t in range(steps):
                    a = torch.rand(n, envs.action_dim, device="cuda", generator=g) * 2 - 1
                    obs, rew, term, trunc, info = envs.step(a)
                    if t % 25 == 24:
                        assert torch.isfinite(obs["observation"]).all() and torch.isfinite(rew).all(), (obj, ctrl, fetch, t)
                        assert float(obs["observation"].abs().max()) < 20.0      # (a cube struck by a finger link can leave at > 5 m/s)
                    done = term | trunc
                    if done.any():
                        len_sum += float(info["episode"]["l"][done].float().sum()); len_cnt += int(done.sum())
                if steps >= 100:
                    assert len_cnt >= 2 * n and len_sum / len_cnt > 40      # random policies rarely succeed: most episodes hit the time limit
                envs.close()
