#!/usr/bin/env python3
"""Step time with lock-stepped episodes (every env resets on the same step) against desynchronised ones (per-env random
initial `elapsed`), per build.   python tools/desync_probe.py [lib.so ...]

With desynchronised episodes every wave always holds some environment in its first steps after a reset, which is what a
training run looks like after the first few hundred steps; the lock-stepped figure hides whatever a fresh episode costs."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r'''
import sys, os, torch
sys.path.insert(0, %r)
from mycobotgym_amd import MyCobotVecEnv
n = 8192
for obj, ctrl, k in ((False, "joint", 400), (False, "IK", 100), (False, "mocap", 200), (True, "joint", 200), (True, "IK", 50)):
    row = []
    for desync in (False, True):
        envs = MyCobotVecEnv(n, has_object=obj, controller_type=ctrl, reward_type="dense")
        envs.reset(seed=0)
        if desync:
            g0 = torch.Generator(device="cuda"); g0.manual_seed(99)
            envs.set_state(elapsed=torch.randint(0, 50, (n,), device="cuda", generator=g0, dtype=torch.int32))
        g = torch.Generator(device="cuda"); g.manual_seed(1234)
        pool = torch.rand(16, n, envs.action_dim, device="cuda", generator=g) * 2 - 1
        for t in range(100): envs.step_async(pool[t %% 16])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(k): envs.step_async(pool[t %% 16])
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / k)
        envs.close()
    print(f"   {'pnp' if obj else 'reach'}-{ctrl}: lockstep {row[0]:.4f}  desync {row[1]:.4f} ms/step  (x{row[1] / row[0]:.2f})", flush=True)
''' % ROOT
for lib in (sys.argv[1:] or [os.path.join(ROOT, "mycobotgym_amd", "libmycobot_hip.so")]):
    print(lib, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MCG_LIB=os.path.abspath(lib)), check=True)
