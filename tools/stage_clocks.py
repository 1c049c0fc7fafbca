#!/usr/bin/env python3
"""Per-stage shader-clock shares of the step kernels (development helper).

    hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -DMCG_STAGE_CLOCKS -Iinclude -Imycobotgym_amd/csrc \
          mycobotgym_amd/csrc/mcg_hip.hip -o ab/clocks.so
    MCG_LIB=ab/clocks.so python tools/stage_clocks.py [--fresh-actions]

Lane 0 of every wave accumulates s_memtime deltas per stage; the table is the sum over waves and launches.
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mycobotgym_amd import MyCobotVecEnv, _abi

NAMES = ["load", "controller", "sincos", "rne", "actuation", "crb->M", "constraint rows", "g0", "newton: other (setup, line search)", "euler: factor M+hB, solve, integrate",
         "collide", "cube solve", "coupled solve", "cube finish", "post (obs/reward/reset/store)",
         "newton: build H", "newton: factor H", "newton: solve", "newton: active-set check", "euler: forces/rhs"]
COUNTS = ["robot sub-steps", "robot Newton iterations", "robot line searches", "cube Newton iterations", "cube line searches"]
fresh = "--fresh-actions" in sys.argv
L = _abi.load()
n = 8192
for obj, ctrl, k in ((False, "joint", 200), (False, "IK", 50), (True, "joint", 100), (True, "IK", 20)):
    envs = MyCobotVecEnv(n, has_object=obj, controller_type=ctrl, reward_type="dense")
    envs.reset(seed=0)
    a = torch.rand(n, envs.action_dim, device="cuda") * 2 - 1
    for _ in range(60): envs.step(a)
    out = (C.c_ulonglong * (len(NAMES) + len(COUNTS)))()
    assert L.mcg_debug_stage_clocks(out, 1) == 0
    for _ in range(k):
        if fresh: a = torch.rand(n, envs.action_dim, device="cuda") * 2 - 1
        envs.step(a)
    assert L.mcg_debug_stage_clocks(out, 1) == 0
    cnt = list(out)[len(NAMES):]; out = list(out)[:len(NAMES)]
    tot = sum(out); waves = n // (32 if obj else 64)
    sub = (20 if ctrl == "joint" else 100)
    print(f"{'pnp' if obj else 'reach'}-{ctrl}: {tot / waves / k / sub:.0f} clocks per wave per sub-step (all stages / sub-steps)")
    for nm, v in zip(NAMES, out):
        if v: print(f"   {nm:32s} {100.0 * v / tot:5.1f} %   {v / waves / k / sub:8.0f} clk/sub-step")
    print("   per wave-sub-step: " + ", ".join(f"{nm} {v / max(cnt[0], 1):.2f}" for nm, v in zip(COUNTS[1:], cnt[1:])))
    envs.close()
