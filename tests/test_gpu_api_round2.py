"""GPU: the callers either side of the hot path and the remaining API rows, on the real engine.

* the five Reach-RewardShaping ids: Reach observation / goal / reset over physics that still carries the (hidden) cube
  (mycobot.py:296-298, 402-448, 475-481), against the oracle configured the same way;
* the SB3 VecEnv adapter over the real engine (scripts/train.py:80-107, scripts/eval_model.py:96-147): terminal_observation,
  TimeLimit.truncated, Monitor's episode statistics, is_success -- checked against the oracle's outputs;
* a checkpoint (state_dict) loaded into a freshly constructed engine with another seed continues bit for bit, auto-resets and
  running episode statistics included;
* two fresh processes sharing cuda:0 (gloo rendezvous, the BENCH_SHARE_GPU rehearsal of the multi-GPU run) produce the two halves of
  the single-process 2N-env run bit for bit, and reduce their episode statistics through sharding.reduce_episode_stats;
* the bad-state guard (mj_checkPos / mj_checkVel): a poisoned env is back at qpos0 after the step, its neighbours are untouched.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


# ------------------------------------------------------------------------------------- Reach + reward_shaping (hidden cube)
@pytest.mark.parametrize("controller", ["joint", "IK", "mocap"])
def test_reach_reward_shaping_hidden_cube(torch_cuda, controller):
    from tests.common import make_pair, make_oracle, sync_oracle_to, step_errors, twin_errors, assert_within_oracle_sensitivity
    n = 128
    envs, ora = make_pair(n, has_object=False, controller_type=controller, reward_type="reward_shaping", seed=4)
    twin = make_oracle(n, has_object=False, controller_type=controller, reward_type="reward_shaping", seed=4); twin.reset(seed=4)
    prng = np.random.default_rng(2); terrs = []
    assert envs.obs_dim == 10 == ora.obs_dim and envs.nq == 19            # Reach observation over physics with the cube
    obs, _ = envs.reset(seed=4)
    o_obs, o_ag, o_dg = ora.reset(seed=4)
    assert np.array_equal(obs["desired_goal"].cpu().numpy(), o_dg)
    assert np.abs(obs["observation"].cpu().numpy() - o_obs).max() < 1e-14
    assert np.abs(obs["achieved_goal"].cpu().numpy() - o_obs[:, :3]).max() < 1e-14          # achieved goal = gripper position
    rng = np.random.default_rng(3)
    errs, rewards, cube_z = [], [], []
    for t in range(55):                                          # crosses the TimeLimit reset: the cube returns to its MJCF pose
        sync_oracle_to(envs, ora)
        state = ora.get_state()
        a = rng.uniform(-1, 1, (n, envs.action_dim)).astype(np.float32)
        e, flags_equal, o = step_errors(envs, ora, a)
        assert flags_equal
        errs.append(e); rewards.append(o["reward"])
        if controller == "IK": terrs.append(twin_errors(twin, state, a, o, prng))
        cube_z.append(envs.get_state()["qpos"][14].cpu().numpy().copy())
    if controller == "IK": assert_within_oracle_sensitivity(errs, terrs, "[reach reward_shaping IK env-step]")
    errs = np.concatenate(errs); rewards = np.concatenate(rewards)
    print(f"\n[reach reward_shaping, {controller}] env-step from identical state: median {np.median(errs):.2e} max {errs.max():.2e}; "
          f"reward range {rewards.min():.2f} .. {rewards.max():.2f}; cube z after 1 / 49 / 50 steps {cube_z[0][0]:.4f} {cube_z[48][0]:.4f} {cube_z[49][0]:.4f}")
    if controller == "IK": assert np.median(errs) < 3e-10
    else: assert errs.max() < 1e-8
    assert 0 < rewards.min() and rewards.max() <= 20.0 + 1e-9    # the reach stage only: 100 * 0.2 * (1 - tanh d)
    on_table = (cube_z[48] < 0.2005) & (cube_z[48] > 0.19)                   # the size-zero cube has dropped onto the table top (a pad may press it in a
    assert on_table.mean() > 0.9 and np.isfinite(cube_z[48]).all()           # little; a finger link that sweeps over the point carries it off: mocap, rare) ...
    assert np.all(cube_z[49] == 0.21)                                        # ... and the reset at step 50 puts it back at z = 0.21 (init_qpos)
    envs.close()


# ------------------------------------------------------------------------------------------------ SB3 adapter on the engine
@pytest.mark.parametrize("env_id", ["MyCobotReach-Dense-joint-v0", "MyCobotPickAndPlace-Sparse-IK-v0"])
def test_sb3_adapter_over_the_real_engine(torch_cuda, env_id):
    from mycobotgym_amd import make
    from mycobotgym_amd.registry import REGISTRY
    from mycobotgym_amd.sb3_adapter import MyCobotSB3VecEnv
    from tests.common import make_oracle
    n = 64
    kw = REGISTRY[env_id]
    thr = 0.05                                   # a generous threshold so that successes (terminated, not truncated) occur too
    venv = MyCobotSB3VecEnv(make(env_id, num_envs=n, seed=9, distance_threshold=thr))
    ora = make_oracle(n, has_object=kw["has_object"], controller_type=kw["controller_type"], reward_type=kw["reward_type"], seed=9,
                      distance_threshold=thr)
    # the oracle's own sensitivity (a second oracle from the same states perturbed by 1e-14): the yardstick of the chaotic IK env-steps
    twin = make_oracle(n, has_object=kw["has_object"], controller_type=kw["controller_type"], reward_type=kw["reward_type"], seed=9,
                       distance_threshold=thr) if kw["controller_type"] == "IK" else None
    if twin is not None: twin.reset(seed=9)
    prng = np.random.default_rng(1); twin_errs = []
    venv.seed(9)
    obs = venv.reset()
    o_obs, _, o_dg = ora.reset(seed=9)
    assert isinstance(obs["observation"], np.ndarray) and np.array_equal(obs["desired_goal"], o_dg)
    rng = np.random.default_rng(0)
    n_done = n_succ = 0
    term_errs = []
    ret = np.zeros(n); length = np.zeros(n, dtype=int)
    for t in range(60):
        # teacher-forced: the engine continues from the oracle's state (chaotic dynamics)
        s = ora.get_state()
        venv.envs.set_state(qpos=s["qpos"].T.copy(), qvel=s["qvel"].T.copy(), ctrl=s["ctrl"].T.copy(), warm=s["warm"].T.copy(),
                            qpos_lag=s["qpos_lag"].T.copy(), goal=s["goal"].T.copy(), elapsed=s["elapsed"], episode=s["episode"])
        a = rng.uniform(-1, 1, (n, venv.envs.action_dim)).astype(np.float32)
        if twin is not None:
            s2 = dict(s); s2["qpos"] = s["qpos"] + 1e-14 * np.sign(prng.normal(size=s["qpos"].shape)); twin.set_state(**s2)
        obs, rew, dones, infos = venv.step(a)
        o = ora.step(a)
        if twin is not None:
            ot = twin.step(a)
            od = o["terminated"].astype(bool) | o["truncated"].astype(bool)
            twin_errs.append(np.abs(ot["final_obs"][od] - o["final_obs"][od]).max(axis=1) if od.any() else np.zeros(0))
        assert rew.dtype == np.float32 and dones.dtype == bool and len(infos) == n
        o_done = o["terminated"].astype(bool) | o["truncated"].astype(bool)
        assert np.array_equal(dones, o_done)
        assert np.median(np.abs(obs["observation"] - o["obs"]).max(axis=1)) < (1e-7 if kw["controller_type"] == "IK" else 1e-12)
        assert np.abs(rew - o["reward"].astype(np.float32)).max() <= (1e-6 if kw["reward_type"] == "dense" else 0)
        ret += o["reward"]; length += 1
        for i in range(n):
            assert infos[i]["is_success"] == bool(o["is_success"][i])
            if dones[i]:
                n_done += 1; n_succ += int(o["is_success"][i])
                term = infos[i]["terminal_observation"]
                assert set(term) == {"observation", "achieved_goal", "desired_goal"}
                term_errs.append(np.abs(term["observation"] - o["final_obs"][i]).max())
                assert np.array_equal(term["desired_goal"], o["final_desired"][i])
                assert infos[i]["TimeLimit.truncated"] == (bool(o["truncated"][i]) and not bool(o["terminated"][i]))
                assert infos[i]["episode"]["l"] == int(o["ep_length"][i]) == length[i]
                ret[i] = 0; length[i] = 0
            else:
                assert "terminal_observation" not in infos[i] and "episode" not in infos[i]
    assert n_done >= n                                            # every env hit the TimeLimit at least once
    term_errs = np.array(term_errs)                               # terminal_observation == the oracle's final observation
    if kw["controller_type"] == "IK":      # 100 chaotic sub-steps, pads meeting the table: bounded by the oracle's own sensitivity, quantile by quantile
        from tests.common import assert_within_oracle_sensitivity
        assert_within_oracle_sensitivity([term_errs], [np.concatenate(twin_errs)], "SB3 terminal observations (IK)")
    else: assert term_errs.max() < 1e-9
    print(f"\n[{env_id}] SB3 adapter over the engine: {n_done} episodes ended in 60 steps, {n_succ} by success")
    r = venv.env_method("compute_reward", obs["achieved_goal"], obs["desired_goal"], None, indices=[0])
    assert len(r) == 1 and r[0].shape == (n,)
    venv.close()


def test_sb3_episode_returns_match_monitor_semantics(torch_cuda):
    """Free-running: infos[i]["episode"]["r"] is the sum of the rewards this adapter handed out during that episode."""
    from mycobotgym_amd import make
    from mycobotgym_amd.sb3_adapter import MyCobotSB3VecEnv
    n = 256
    venv = MyCobotSB3VecEnv(make("MyCobotReach-Dense-joint-v0", num_envs=n, seed=2))
    venv.seed(2); venv.reset()
    rng = np.random.default_rng(1)
    acc = np.zeros(n); cnt = np.zeros(n, dtype=int); checked = 0
    for t in range(120):
        obs, rew, dones, infos = venv.step(rng.uniform(-1, 1, (n, 7)).astype(np.float32))
        acc += rew.astype(np.float64); cnt += 1
        for i in np.nonzero(dones)[0]:
            assert infos[i]["episode"]["l"] == cnt[i]
            assert abs(infos[i]["episode"]["r"] - acc[i]) < 1e-4 * max(1.0, abs(acc[i]))      # rewards were handed out as float32
            acc[i] = 0; cnt[i] = 0; checked += 1
    assert checked >= 2 * n
    venv.close()


# ------------------------------------------------------------------------------------------------------------ checkpoint
@pytest.mark.parametrize("has_object", [False, True])
def test_checkpoint_into_a_fresh_engine(torch_cuda, has_object):
    torch = torch_cuda
    from mycobotgym_amd import MyCobotVecEnv
    n = 192
    kw = dict(has_object=has_object, controller_type="joint", reward_type="dense",
              domain_randomization={"mass": (0.5, 2.0), "friction": (0.5, 1.5)} if has_object else None)
    a_env = MyCobotVecEnv(n, seed=5, **kw)
    a_env.reset(seed=77)                                          # reset(seed=...) re-keys the streams: part of the checkpoint
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    acts = [torch.rand(n, 7, device="cuda", generator=g) * 2 - 1 for _ in range(70)]
    for t in range(30):
        a_env.step(acts[t])
    sd = {k: v.clone() for k, v in a_env.state_dict().items()}
    assert int(sd["seed"][0]) == 77 and sd["ep_length"].eq(30).all() and (sd["ep_return"] < 0).all()
    b_env = MyCobotVecEnv(n, seed=123456, **kw)                   # a different seed, never reset
    b_env.load_state_dict(sd)
    for t in range(30, 70):                                       # crosses the auto-reset at step 50: goals, cube, DR draws, Monitor stats
        oa, ra, ta, tra, ia = a_env.step(acts[t]); ob, rb, tb, trb, ib = b_env.step(acts[t])
        for k in oa: assert torch.equal(oa[k], ob[k]), (t, k)
        assert torch.equal(ra, rb) and torch.equal(tra, trb)
        assert torch.equal(ia["episode"]["r"], ib["episode"]["r"]) and torch.equal(ia["episode"]["l"], ib["episode"]["l"])
        assert torch.equal(ia["final_observation"]["observation"], ib["final_observation"]["observation"])
    sa, sb = a_env.state_dict(), b_env.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    a_env.close(); b_env.close()


def test_step_returns_fresh_copies(torch_cuda):
    torch = torch_cuda
    from mycobotgym_amd import MyCobotVecEnv
    envs = MyCobotVecEnv(32, has_object=False, controller_type="joint", reward_type="sparse")
    envs.reset(seed=0)
    o1, r1, *_ = envs.step(torch.zeros(32, 7, device="cuda"))
    keep = o1["observation"].clone()
    o2, r2, *_ = envs.step(torch.ones(32, 7, device="cuda"))
    assert torch.equal(o1["observation"], keep) and o1["observation"].data_ptr() != o2["observation"].data_ptr()   # mycobot.py:280-282
    assert r1.dtype == torch.float32                                                                              # mycobot.py:293
    o3, r3, *_ = envs.step(torch.ones(32, 7, device="cuda"), copy=False)
    o4, r4, *_ = envs.step(torch.ones(32, 7, device="cuda"), copy=False)
    assert o3["observation"].data_ptr() == o4["observation"].data_ptr()
    envs.close()


def test_model_path_is_checked(torch_cuda):
    from mycobotgym_amd import MyCobotVecEnv
    with pytest.raises(ValueError, match="precompiled"):
        MyCobotVecEnv(4, has_object=False, controller_type="joint", model_path="./assets/some_other_robot.xml")
    with pytest.raises(ValueError, match="precompiled"):
        MyCobotVecEnv(4, has_object=False, controller_type="mocap", model_path="./assets/mycobot280.xml")
    MyCobotVecEnv(4, has_object=False, controller_type="joint", model_path="./assets/mycobot280.xml").close()


# ------------------------------------------------------------------------------------------- two processes, one GPU (gloo)
_CHILD = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from mycobotgym_amd import MyCobotVecEnv
from mycobotgym_amd.sharding import shard, reduce_episode_stats
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
n, task = %(n)d, %(task)r
off, total = shard(rank, world, n)
envs = MyCobotVecEnv(n, has_object=(task == "pnp"), controller_type="joint", reward_type="dense", seed=31, env_id_offset=off)
envs.reset(seed=31)
g = torch.Generator(device="cuda"); g.manual_seed(5)
tot = {"episodes": 0.0}
outs = []
for t in range(%(steps)d):
    a = (torch.rand(total, 7, device="cuda", generator=g) * 2 - 1)[off:off + n].contiguous()     # the global action batch, this rank's rows
    obs, rew, term, trunc, info = envs.step(a)
    outs.append(torch.cat([obs["observation"], obs["desired_goal"], rew[:, None], trunc[:, None].double()], dim=1).cpu())
    st = reduce_episode_stats(info["episode"]["r"], info["episode"]["l"], info["is_success"], trunc, device="cpu")
    tot["episodes"] += st["episodes"]
torch.save(torch.stack(outs), %(out)r + f".{rank}.pt")
if rank == 0:
    json.dump(tot, open(%(out)r + ".json", "w"))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("task", ["reach", "pnp"])
def test_two_processes_share_the_gpu_and_match_one(torch_cuda, task, tmp_path):
    torch = torch_cuda
    from mycobotgym_amd import MyCobotVecEnv
    n, steps = 96, 55
    out = str(tmp_path / f"shard_{task}")
    code = _CHILD % dict(root=ROOT, n=n, task=task, steps=steps, out=out)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533" if task == "reach" else "29534", WORLD_SIZE="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0, 0]
    got = torch.cat([torch.load(out + f".{r}.pt") for r in range(2)], dim=1)            # [steps, 2n, ...]
    one = MyCobotVecEnv(2 * n, has_object=(task == "pnp"), controller_type="joint", reward_type="dense", seed=31)
    one.reset(seed=31)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    episodes = 0
    for t in range(steps):
        a = torch.rand(2 * n, 7, device="cuda", generator=g) * 2 - 1
        obs, rew, term, trunc, info = one.step(a)
        want = torch.cat([obs["observation"], obs["desired_goal"], rew[:, None], trunc[:, None].double()], dim=1).cpu()
        assert torch.equal(got[t], want), t                       # bit for bit, auto-reset draws (global env ids) included
        episodes += int(trunc.sum())
    assert json.load(open(out + ".json"))["episodes"] == episodes >= 2 * n
    one.close()


# ----------------------------------------------------------------------------------------------------- bad-state guard
@pytest.mark.parametrize("has_object", [False, True])
def test_bad_state_guard(torch_cuda, has_object):
    """mj_checkPos / mj_checkVel [RECALL]: MuJoCo resets mjData when a coordinate is NaN or beyond 1e10, at the start of the mj_step that
    meets it.  The engine checks the state it loads and every sub-step's new state (round 2: once per env-step): a poisoned env is reset
    to qpos0 with zero velocity, ctrl and warm start BEFORE its first sub-step and then simulated like the oracle's -- the same
    observations and the same state after the step; the others are untouched.  (With an object: the whole mjData is reset, the cube too.)"""
    torch = torch_cuda
    from tests.common import make_pair, sync_oracle_to, step_errors
    n = 64
    envs, ora = make_pair(n, has_object=has_object, controller_type="joint", reward_type="dense", seed=8)
    envs.reset(seed=8); ora.reset(seed=8)
    envs.counters(clear=True)
    a = np.random.default_rng(0).uniform(-1, 1, (n, 7)).astype(np.float32)
    step_errors(envs, ora, a)
    sync_oracle_to(envs, ora)
    s = envs.get_state()
    bad = [3, 17, 40]
    s["qvel"][2, bad[0]] = float("nan"); s["qpos"][5, bad[1]] = 3e10; s["qvel"][7, bad[2]] = float("inf")
    envs.set_state(**s)
    so = ora.get_state()
    so["qvel"][bad[0], 2] = np.nan; so["qpos"][bad[1], 5] = 3e10; so["qvel"][bad[2], 7] = np.inf
    ora.set_state(**so)
    obs, rew, term, trunc, info = envs.step(torch.as_tensor(a))
    o = ora.step(a)
    err = np.abs(obs["observation"].cpu().numpy() - o["obs"]).max(axis=1)
    st, so = envs.get_state(), ora.get_state()
    qerr = np.abs(st["qpos"].cpu().numpy().T - so["qpos"]).max(axis=1)
    print(f"\nbad-state reset, has_object={has_object}: obs error of the three reset envs {err[bad]}, of the others {np.delete(err, bad).max():.2e}; "
          f"qpos error of the reset envs {qerr[bad]}; counters {envs.counters()}")
    assert err.max() < 1e-8 and qerr.max() < 1e-8            # reset envs included: 20 sub-steps from qpos0 with zero controls, as the oracle
    assert all(torch.isfinite(v.double()).all() for v in st.values())
    assert all(int(ora.data(i).get("warning_badstate", (1,), np.int32)[0]) >= 1 for i in bad)
    assert envs.counters()["bad_state_resets"] >= 3
    # one step later both sides are ordinary states again
    obs, *_ = envs.step(torch.as_tensor(a))
    assert torch.isfinite(obs["observation"]).all()
    envs.close()


def test_bad_state_in_the_middle_of_an_env_step(torch_cuda):
    """mj_checkAcc inside a LATER sub-step of an env-step (the PickAndPlace kernels checked the first sub-step
    only).  A cube in the air beside the table falling at 1e8 m/s passes mj_checkPos / mj_checkVel, touches nothing in sub-step 0 and is
    2e5 m below the ground plane in sub-step 1: its contact acceleration exceeds 1e10, MuJoCo [RECALL] resets mjData there and carries
    on from qpos0.  The engine sees the cube's acceleration when the NEXT sub-step starts (the verdict rides on the flags the waves
    exchange anyway) and resets the cube there and the robot one exchange later: one / two sub-steps of the remaining eighteen behind
    the oracle, both from rest -- so the env-step's end state agrees to the little that one sub-step at rest moves (asserted), the
    counters report the resets, and the neighbours are untouched."""
    torch = torch_cuda
    from tests.common import make_pair, sync_oracle_to
    n = 64
    envs, ora = make_pair(n, has_object=True, controller_type="joint", reward_type="dense", seed=8)
    envs.reset(seed=8); ora.reset(seed=8)
    a = np.zeros((n, 7), np.float32)
    sync_oracle_to(envs, ora)
    envs.counters(clear=True)
    bad = [5, 33]
    so = ora.get_state()
    for i in bad:
        so["qpos"][i, 12:15] = [0.5, 0.0, 1.0]; so["qvel"][i, 14] = -1e8
    so["qpos_lag"] = so["qpos"].copy()
    ora.set_state(**so)
    sync_oracle_to(envs, ora)
    obs, rew, term, trunc, info = envs.step(torch.as_tensor(a))
    o = ora.step(a)
    err = np.abs(obs["observation"].cpu().numpy() - o["obs"]).max(axis=1)
    st, s2 = envs.get_state(), ora.get_state()
    qerr = np.abs(st["qpos"].cpu().numpy().T - s2["qpos"]).max(axis=1)
    c = envs.counters()
    print(f"\nbad acceleration in sub-step 1 of an env-step: obs error of the two reset envs {err[bad]}, of the others {np.delete(err, bad).max():.2e}; "
          f"qpos error of the reset envs {qerr[bad]}; counters {c}")
    assert all(int(ora.data(i).get("warning_badstate", (1,), np.int32)[0]) >= 1 for i in bad)
    assert c["bad_state_resets"] >= 2
    assert all(torch.isfinite(v.double()).all() for v in st.values())
    assert np.delete(err, bad).max() < 1e-8                       # the neighbours never notice
    assert err[bad].max() < 5e-3 and qerr[bad].max() < 5e-3       # the reset envs: the oracle's state up to the one / two sub-steps of lag (at rest)
    assert np.abs(st["qpos"].cpu().numpy()[14, bad] - 0.21).max() < 0.02                     # (the cube is back at its model pose on the table, not 2e6 m down)
    envs.close()


# ------------------------------------------------------------------------------------------------- RCCL, one rank
def test_rccl_single_rank_reduction(torch_cuda, tmp_path):
    """The logging collective through RCCL itself (backend "nccl"), world size 1 on this box's one GPU: the call the 8-GPU run makes
    (sharding.reduce_episode_stats on device tensors).  reduce_episode_stats issues dist.all_reduce whenever a process group exists
    (no world-size guard); that the library really ran a collective is read from RCCL's own log (NCCL_DEBUG_SUBSYS=COLL prints one
    "AllReduce" line per call)."""
    code = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from mycobotgym_amd import MyCobotVecEnv
from mycobotgym_amd.sharding import reduce_episode_stats
envs = MyCobotVecEnv(256, has_object=False, controller_type="joint", reward_type="dense", seed=3)
envs.reset(seed=3)
g = torch.Generator(device="cuda"); g.manual_seed(0)
for t in range(50):
    obs, rew, term, trunc, info = envs.step(torch.rand(256, 7, device="cuda", generator=g) * 2 - 1)
st = reduce_episode_stats(info["episode"]["r"], info["episode"]["l"], info["is_success"], trunc)      # all_reduce over RCCL
assert dist.get_backend() == "nccl"
json.dump(st, open(%r, "w"))
dist.destroy_process_group()
''' % (ROOT, str(tmp_path / "rccl.json"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,COLL")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    log = p.stdout + p.stderr
    coll = [ln for ln in log.splitlines() if "AllReduce" in ln]
    print("\nRCCL log lines naming the collective:", len(coll), coll[:2])
    assert coll, "RCCL logged no AllReduce: the collective did not run\n" + log[-1500:]
    st = json.load(open(tmp_path / "rccl.json"))
    assert st["episodes"] == 256 and st["mean_length"] == 50.0 and st["mean_return"] < 0
