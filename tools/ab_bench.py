#!/usr/bin/env python3
"""A/B timing of several builds of libmycobot_hip.so on the same GPU box (development helper).

    python tools/ab_bench.py ab/old.so ab/new.so      # each build in its own process, via MCG_LIB
"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r'''
import sys, os, torch
sys.path.insert(0, %r)
from mycobotgym_amd import MyCobotVecEnv
n = 8192
for obj, ctrl, k in ((False, "joint", 400), (False, "IK", 100), (True, "joint", 200), (True, "IK", 40)):
    envs = MyCobotVecEnv(n, has_object=obj, controller_type=ctrl, reward_type="dense")
    envs.reset(seed=0)
    a = torch.rand(n, envs.action_dim, device="cuda") * 2 - 1
    for _ in range(60): envs.step(a)          # past the first auto-reset, cubes settled
    envs.time_steps(a, k)                     # sustained load first: short windows run at idle clocks
    torch.cuda.synchronize()
    ms = min(envs.time_steps(a, k) for _ in range(3)) / k
    print(f"  {'pnp' if obj else 'reach'}-{ctrl}: {ms:.3f} ms/step  {n / ms * 1e3:.3e} env-steps/s", flush=True)
    envs.close()
''' % ROOT
for lib in sys.argv[1:]:
    print(lib, flush=True)
    env = dict(os.environ, MCG_LIB=os.path.abspath(lib))
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
