#!/usr/bin/env python3
"""Throughput against the number of environments per GPU (development helper; bench.py is the contract).

BASELINE.json's metric is quoted at N = 8192 per GPU, which is 128 (Reach) / 256 (PickAndPlace) waves on 1024 SIMDs; this
sweep shows what the same kernels deliver once the chip is full.

    python tools/sweep_envs.py > profiles/<tag>/sweep_envs.json
"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mycobotgym_amd import MyCobotVecEnv

out = []
for obj, ctrl in ((False, "joint"), (False, "IK"), (False, "mocap"), (True, "joint")):
    for n in (8192, 16384, 32768, 65536, 131072, 262144):
        envs = MyCobotVecEnv(n, has_object=obj, controller_type=ctrl, reward_type="dense")
        envs.reset(seed=0)
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        pool = torch.rand(16, n, envs.action_dim, device="cuda", generator=g) * 2 - 1     # as bench.py: 16 resident action batches
        k = max(8, min(200, int(2.0e6 / n) * (1 if ctrl != "IK" else 1) // (5 if ctrl == "IK" else 1)))
        for t in range(max(60, k, 3000 * 8192 // n)): envs.step_async(pool[t % 16])     # ~1 s of sustained load first: short windows run at idle clocks
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(k): envs.step_async(pool[t % 16])
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / k
        row = {"task": "pnp" if obj else "reach", "controller": ctrl, "envs": n, "ms_per_step": ms, "env_steps_per_sec": n / ms * 1e3}
        out.append(row); print(json.dumps(row), file=sys.stderr, flush=True)
        envs.close()
print(json.dumps(out, indent=1))
