#!/usr/bin/env python3
"""Development helper: census of the contact lists under a bench.py workload, by environment and by 32-environment workgroup.

    python tools/contact_census.py pnp-IK [warmup] [samples]

For every sampled step: the share of environments whose list reaches the robot (the flagged ones, solved by the cooperative phase),
the histogram of their list lengths / row counts, and per workgroup the number of flagged environments (= the depth of the phase).
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
from mycobotgym_amd import MyCobotVecEnv, _abi

case = sys.argv[1] if len(sys.argv) > 1 else "pnp-IK"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 100
samples = int(sys.argv[3]) if len(sys.argv) > 3 else 8
task, controller, dr, grasp = bench.CASES[case]
n = 8192
envs = MyCobotVecEnv(n, has_object=True, controller_type=controller, reward_type="dense", seed=0)
envs.reset(seed=0)
g = torch.Generator(device="cuda"); g.manual_seed(1234)
envs.set_state(elapsed=torch.randint(0, 50, (n,), device="cuda", generator=g, dtype=torch.int32))
pool = torch.rand(16, n, envs.action_dim, device="cuda", generator=g) * 2 - 1
for t in range(warm): envs.step_async(pool[t % 16])
hist_n = np.zeros(_abi.MAXCON + 1, int); hist_rows = np.zeros(8, int); hist_wg = np.zeros(33, int); flagged_tot = 0; types = np.zeros(5 + 2 * _abi.NMESH, int)
for s in range(samples):
    for t in range(3): envs.step_async(pool[(s * 3 + t) % 16])
    kc = {k: v.cpu().numpy() for k, v in envs.debug_contacts().items()}
    cnt, typ = kc["count"], kc["type"]
    valid = np.arange(_abi.MAXCON)[None, :] < cnt[:, None]
    robot = valid & (typ != 0)                       # PAIR_TABLE_CUBE = 0 is the only pair that does not reach the robot
    flagged = robot.any(1)
    flagged_tot += flagged.sum()
    for c in cnt[flagged]: hist_n[c] += 1
    # rows if packed tightly: 6 per condim-4 contact, 4 per condim-3 (a mesh on the table / the ground: types 5 .. 5 + NMESH - 1)
    cd3 = valid & (typ >= 5) & (typ < 5 + _abi.NMESH)
    rows = (6 * valid.sum(1) - 2 * cd3.sum(1))[flagged]
    for r in rows: hist_rows[min(r // 16, 7)] += 1
    for w in flagged.reshape(-1, 32).sum(1): hist_wg[w] += 1
    for t_ in typ[robot]: types[t_] += 1
tot = n * samples
print(f"{case}: flagged {flagged_tot / tot * 100:.2f} % of environments")
print("  list length of flagged envs:", {k: int(v) for k, v in enumerate(hist_n) if v})
print("  contact rows (tight packing, without limits) in bins of 16:", {f"{16*k}-{16*k+15}": int(v) for k, v in enumerate(hist_rows) if v})
print("  flagged per workgroup:", {k: int(v) for k, v in enumerate(hist_wg) if v})
print("  robot-reaching entries by pair type:", {k: int(v) for k, v in enumerate(types) if v})
