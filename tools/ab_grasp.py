#!/usr/bin/env python3
"""A/B of builds on the coupled (pad-contact) path: every env closing its gripper on the cube (scripted grasp), 8192 envs.

    python tools/ab_grasp.py ab/a.so ab/b.so [--rounds 2]
"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r'''
import sys, os, torch
sys.path.insert(0, %r)
from mycobotgym_amd import MyCobotVecEnv
from mycobotgym_amd.scenarios import grasp_state
n = 8192
envs = MyCobotVecEnv(n, has_object=True, controller_type="joint", reward_type="reward_shaping", max_episode_steps=10 ** 9)
envs.reset(seed=0)
st = grasp_state(n, seed=0)
act = torch.as_tensor(st.pop("action"), device="cuda")
envs.set_state(**st)
for t in range(5): envs.step_async(act)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(40): envs.step_async(act)
e1.record(); torch.cuda.synchronize()
b = envs._buf
print(f"   grasp: {e0.elapsed_time(e1) / 40:.3f} ms/step   ({n * 40 / e0.elapsed_time(e1) / 1e3:.2f} M env-steps/s); "
      f"mean shaped reward {float(b['reward'].mean()):.2f}, holding {float((b['reward'] >= 50).double().mean()):.2f}", flush=True)
envs.close()
''' % ROOT
args = sys.argv[1:]
rounds = 2
if "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
for r in range(rounds):
    for lib in args:
        print(lib, flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MCG_LIB=os.path.abspath(lib)), check=True)
