"""The KERNELS' collision pass (exported by the test-only mcg_debug_contacts) against the independent exact computation of
tests/indep_collision.py -- not against the oracle, with which the kernels share an author."""
import numpy as np
import pytest

from tests import indep_collision as ic
from tests.common import load_json

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def _poses():
    """Arm / gripper / cube configurations with contacts of every kind the kernels generate."""
    from tests.test_gpu_pickandplace import _contact_poses, _finger_mesh_poses, _link_cube_poses
    from mycobotgym_amd.scenarios import grasp_state
    tab = load_json("mycobot280")
    q0 = np.array(tab["qpos0"], float)
    out = []
    for p in _contact_poses("pad", 96):
        q = q0.copy(); q[:12] = p[:12]; out.append(q)
    for p in _contact_poses("mesh", 96, seed=1):
        q = q0.copy(); q[:12] = p[:12]; out.append(q)
    out += list(_finger_mesh_poses(96))
    out += list(_finger_mesh_poses(48, seed=8, meshes=("right_gear_link", "left_gear_link", "right_hinge_link", "left_hinge_link", "gripper_base")))
    out += list(_link_cube_poses(64))
    for p in _contact_poses("gripper_mesh", 64, seed=3):
        q = q0.copy(); q[:12] = p[:12]; out.append(q)
    g = np.asarray(grasp_state(64, seed=0)["qpos"]); g = g.T if g.shape[0] == 19 else g
    out += list(g)
    rng = np.random.default_rng(5)
    for k in range(160):                 # the cube alone: tumbling on the table, over its rim, on the ground, faces and edges parallel
        q = q0.copy()
        ang = rng.choice([0.0, 1e-9, 3e-4, 0.3, 1.0]); ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        q[15:19] = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
        if k % 3 == 0: q[12:15] = [rng.uniform(-0.15, 0.15), rng.uniform(-0.1, 0.2), 0.2 + rng.uniform(0.006, 0.017)]
        elif k % 3 == 1: q[12:15] = [0.2 + rng.uniform(-0.012, 0.012), rng.uniform(-0.1, 0.2), 0.2 + rng.uniform(0.004, 0.012)]
        else: q[12:15] = [rng.uniform(0.3, 0.5), rng.uniform(-0.3, 0.3), rng.uniform(0.005, 0.017)]
        out.append(q)
    return tab, np.array(out)


def test_kernel_contact_lists_against_the_exact_rule(torch_cuda):
    torch = torch_cuda
    from oracle import pyoracle as po                      # kinematics only (body and geom poses); its collision code is not consulted
    from mycobotgym_amd import MyCobotVecEnv
    from mycobotgym_amd.model.mjcf import _np_model
    from mycobotgym_amd.model.specialize import specialize
    tab, Q = _poses()
    n = len(Q)
    spec = specialize(_np_model(tab))
    envs = MyCobotVecEnv(n, has_object=True, controller_type="joint", reward_type="dense", seed=0)
    envs.reset(seed=0)
    st = envs.get_state()
    st["qpos"] = torch.as_tensor(Q.T.copy(), device="cuda"); st["qvel"] = torch.zeros_like(st["qvel"])
    envs.set_state(**st)
    kc = {k: v.cpu().numpy() for k, v in envs.debug_contacts().items()}
    d = po.OracleData(po.OracleModel(tab, enable_contact=False))
    stats = {}; checked = 0; with_contacts = 0; types = set()
    nb, ng = tab["nbody"], tab["ngeom"]
    for i in range(n):
        if kc["dropped"][i] > 0:
            continue                                      # the cap cut this list: its tail is missing by construction
        d.set_state(qpos=Q[i], qvel=np.zeros(18)); d.forward()
        sc = ic.Scene(tab, spec, d.get("xpos", (nb, 3)), d.get("xmat", (nb, 9)), d.get("geom_xpos", (ng, 3)), d.get("geom_xmat", (ng, 9)))
        con = ic.kernel_contacts(kc["count"][i], kc["dist"][i], kc["pos"][i], kc["normal"][i], kc["type"][i])
        ic.check_scene(sc, con, stats, f"env {i}")
        checked += 1; with_contacts += kc["count"][i] > 0
        types |= set(int(t) for t in kc["type"][i][:kc["count"][i]])
    print(f"\nkernel contact lists against the exact rule: {checked} environments checked ({n - checked} cut by the cap), "
          f"{with_contacts} with contacts, pair types seen {sorted(types)}; " + ic.summarize(stats))
    assert checked > 0.8 * n and {0, 1, 2}.issubset(types) and (types & {3, 4}) and len(types & set(range(5, 13))) >= 4 and (types & set(range(13, 19)))
    assert (types & {28, 30}) and (types & {27, 29, 31, 32}) and (types & set(range(19, 26))), sorted(types)      # finger links, gear / hinge links, arm links on the cube
    envs.close()


def test_counters_under_a_random_policy(torch_cuda):
    """mcg_counters after a random-policy PickAndPlace-IK rollout (the reference's default controller): the engine's own bounds -- the
    reset rejection cap, the bad-state reset -- never fire; how often the cap of 16 list entries truncates a list is REPORTED (MuJoCo has no
    such cap) and BOUNDED: at most one contact dropped per hundred coupled environment-sub-steps (round 3's cap of 12 contacts dropped 0.9
    per coupled sub-step), with the share of environment-sub-steps that went through the coupled solve."""
    torch = torch_cuda
    from mycobotgym_amd import MyCobotVecEnv
    n, steps = 2048, 60
    envs = MyCobotVecEnv(n, has_object=True, controller_type="IK", reward_type="dense", seed=2)
    envs.reset(seed=2)
    envs.counters(clear=True)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for t in range(steps):
        envs.step(torch.rand(n, envs.action_dim, device="cuda", generator=g) * 2 - 1)
    c = envs.counters()
    sub = n * steps * 100
    print(f"\nrandom-policy PickAndPlace-IK, {n} envs x {steps} steps: {c}; contacts dropped per million env-sub-steps "
          f"{c['contacts_dropped'] / sub * 1e6:.1f}; coupled env-sub-steps {100.0 * c['coupled_env_substeps'] / sub:.2f} %")
    assert c["reset_cap_hits"] == 0 and c["bad_state_resets"] == 0
    assert 0 < c["coupled_env_substeps"] < sub
    print(f"contacts dropped per coupled env-sub-step {c['contacts_dropped'] / c['coupled_env_substeps']:.4f}")
    assert c["contacts_dropped"] <= 0.01 * c["coupled_env_substeps"]
    envs.close()
