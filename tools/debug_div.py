import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from tests.common import make_pair, load_json, soften_gains, sync_oracle_to
n = 256
for soft in (True, False):
  for ctrl in ("IK", "joint"):
    tab = soften_gains(load_json("mycobot280_reach")) if soft else None
    envs, ora = make_pair(n, table=tab, controller_type=ctrl, reward_type="dense", seed=1)
    o0, _ = envs.reset(seed=1); ora.reset(seed=1)
    g0 = o0["desired_goal"].cpu().numpy().copy()
    rng = np.random.default_rng(42)
    for t in range(51):
        a = rng.uniform(-1, 1, (n, envs.action_dim)).astype(np.float32)
        obs, rew, term, trunc, info = envs.step(torch.as_tensor(a))
        o = ora.step(a)
        if t == 49:
            dg = obs["desired_goal"].cpu().numpy()
            mism = np.abs(dg - o["desired"]).max(axis=1) > 0
            stale = np.abs(dg - g0).max(axis=1) == 0
            st = envs.get_state()
            print(f"soft={soft} {ctrl}: mismatches {mism.sum()} of {n}; stale-goal lanes {stale.sum()}; mism&stale {np.sum(mism&stale)}; idx {np.nonzero(mism)[0][:20]}",
                  "state goal == out dg:", np.abs(st["goal"].cpu().numpy().T - dg).max())
    envs.close()
