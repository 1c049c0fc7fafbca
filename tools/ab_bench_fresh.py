#!/usr/bin/env python3
"""A/B of several builds with bench.py's workload (fresh action batch every step), alternating, in one GPU call.

    python tools/ab_bench_fresh.py ab/a.so ab/b.so [--rounds 3]
"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r'''
import sys, os, torch
sys.path.insert(0, %r)
from mycobotgym_amd import MyCobotVecEnv
n = 8192
res = []
for obj, ctrl, k in ((False, "joint", 600), (False, "IK", 120), (False, "mocap", 300), (True, "joint", 300)):
    envs = MyCobotVecEnv(n, has_object=obj, controller_type=ctrl, reward_type="dense")
    envs.reset(seed=0)
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    pool = torch.rand(16, n, envs.action_dim, device="cuda", generator=g) * 2 - 1
    for t in range(100): envs.step_async(pool[t %% 16])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(k): envs.step_async(pool[t %% 16])
    e1.record(); torch.cuda.synchronize()
    res.append(f"{'pnp' if obj else 'reach'}-{ctrl} {e0.elapsed_time(e1) / k:.4f}")
    envs.close()
print("   " + "   ".join(res) + "   ms/step", flush=True)
''' % ROOT
args = sys.argv[1:]
rounds = 2
if "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
libs = args
for r in range(rounds):
    for lib in libs:
        print(lib, flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MCG_LIB=os.path.abspath(lib)), check=True)
