#!/usr/bin/env python3
"""Development helper: one FULL env-step (all sub-steps inside one launch) of the selected build against the CPU oracle, from states a
random-policy run reached (tools/contacts_ab.py dump) -- the per-sub-step parity tests teacher-force every sub-step and so cannot see a
fault that needs two sub-steps in one launch.

    MCG_LIB=ab/x.so python tools/state_vs_oracle.py gpurun_out/x.pt [case] [n]

Picks the environments whose contact list holds an arm-mesh entry (PAIR_TABLE_LINK0..+7) plus as many without, n in all."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from tests.common import make_pair, step_errors

path = sys.argv[1]
case = sys.argv[2] if len(sys.argv) > 2 else "pnp-IK"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
task, controller, dr, grasp = bench.CASES[case]
d = torch.load(path)
ty, cnt = d["contacts"]["type"], d["contacts"]["count"]
idx = torch.arange(ty.shape[1])[None, :] < cnt[:, None]
arm = ((ty >= 5) & (ty < 13) & idx).any(dim=1)
pick = torch.cat([arm.nonzero().flatten()[: n // 2], (~arm).nonzero().flatten()[: n - min(n // 2, int(arm.sum()))]])[:n]
n = len(pick)
envs, ora = make_pair(n, has_object=True, controller_type=controller, reward_type="dense", seed=0, max_episode_steps=10 ** 9)
envs.reset(seed=0); ora.reset(seed=0)
st = {k: v[..., pick] if v.ndim and v.shape[-1] == ty.shape[0] else v for k, v in d["state"].items()}
ost = {k: st[k].numpy().T.copy() for k in ("qpos", "qvel", "warm", "qpos_lag", "goal")}
ost["ctrl"] = st["ctrl"].numpy().T.copy()
ost["elapsed"] = np.zeros(n, np.int32); ost["episode"] = st["episode"].numpy().astype(np.int32)
ora.set_state(**ost)
envs.set_state(qpos=st["qpos"], qvel=st["qvel"], ctrl=st["ctrl"], warm=st["warm"], qpos_lag=st["qpos_lag"], goal=st["goal"],
               elapsed=torch.zeros(n, dtype=torch.int32), episode=st["episode"])
g = torch.Generator(device="cuda"); g.manual_seed(1234)
a = (torch.rand(n, envs.action_dim, device="cuda", generator=g) * 2 - 1).float().cpu().numpy()
for t in range(2):
    e, flags_equal, o = step_errors(envs, ora, a)
    isarm = arm[pick].numpy()
    for name, m in (("arm-mesh entry at the start", isarm), ("none", ~isarm)):
        if m.any():
            q = np.quantile(e[m], [0.5, 0.9, 0.99, 1.0])
            print(f"{case} env-step {t}: {name}: {m.sum()} envs, error vs oracle 50/90/99/100 %: " + " ".join(f"{v:.2e}" for v in q) + f", above 1e-8: {(e[m] > 1e-8).sum()}")
envs.close()
