#!/usr/bin/env python3
"""Soak run: every task / controller at 8192 envs, random actions, many env-steps; everything must stay finite and the
episode statistics must look like a random policy's.  (Development helper; the parity tests are in tests/.)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mycobotgym_amd import MyCobotVecEnv

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = 8192
for obj in (False, True):
    for ctrl in ("joint", "IK", "mocap"):
        for fetch in ((False,) if ctrl == "joint" else (False, True)):
            envs = MyCobotVecEnv(n, has_object=obj, controller_type=ctrl, fetch_env=fetch, reward_type="reward_shaping" if obj else "dense",
                                 domain_randomization={"mass": (0.5, 2.0), "friction": (0.5, 1.5)} if obj else None)
            envs.reset(seed=0)
            g = torch.Generator(device="cuda"); g.manual_seed(0)
            k = steps if ctrl != "IK" else steps // 5
            t0 = time.perf_counter(); done = 0; bad = 0
            for t in range(k):
                a = torch.rand(n, envs.action_dim, device="cuda", generator=g) * 2 - 1
                obs, rew, term, trunc, info = envs.step(a)
                if t % 50 == 49:
                    bad += int((~torch.isfinite(obs["observation"])).any(dim=1).sum()) + int((~torch.isfinite(rew)).sum())
                    done += int((term | trunc).sum())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"{'pnp' if obj else 'reach'}-{ctrl}{'-fetch' if fetch else ''}: {k} steps, {n * k / dt:.3e} env-steps/s (with host checks), "
                  f"non-finite rows {bad}, max |obs| {float(obs['observation'].abs().max()):.3f}, mean reward {float(rew.mean()):.3f}", flush=True)
            assert bad == 0
            envs.close()
print("soak ok")
