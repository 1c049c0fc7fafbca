#!/usr/bin/env python3
"""Development aid: the share of environments with a robot contact, step by step, under one random IK policy -- HIP engine against the
CPU oracle, both free-running from the same reset (statistics only: trajectories diverge chaotically)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tests.common import make_pair

n = 1024
envs, ora = make_pair(n, has_object=True, controller_type="IK", reward_type="dense", seed=2)
envs.reset(seed=2); ora.reset(seed=2)
rng = np.random.default_rng(0)
envs.counters(clear=True)
for t in range(50):
    a = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
    envs.step(torch.as_tensor(a)); ora.step(a)
    kc = envs.debug_contacts()
    live = torch.arange(kc["type"].shape[1], device=kc["type"].device)[None, :] < kc["count"][:, None]
    g = float(((kc["type"] != 0) & live).any(dim=1).float().mean())
    ne = np.array([int(ora.data(i).get("nentry", (1,), np.int32)[0]) for i in range(n)])
    nc = np.array([int(ora.data(i).get("ncon", (1,), np.int32)[0]) for i in range(n)])
    print(f"step {t:2d}: robot-contact share  gpu {g:.3f}  oracle(entries>4) {(ne > 4).mean():.3f}; gpu mean entries {float(kc['count'].float().mean()):.2f} oracle {ne.mean():.2f}", flush=True)
c = envs.counters()
print(c, "coupled share of env-sub-steps", c["coupled_env_substeps"] / (n * 50 * 100))
