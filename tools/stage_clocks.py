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

NAMES = ["load", "controller", "sincos", "rne", "actuation", "crb->M", "rows: weld / rest", "g0", "newton: other (setup, line search)", "euler: factor M+hB, solve, integrate",
         "cube wave: mesh phase", "cube wave: list scan, zero fill", "coupled solve", "cube wave: waiting at S1b", "post (obs/reward/reset/store)",
         "newton: build H", "newton: factor H", "newton: solve", "newton: active-set check", "euler: forces/rhs",
         "rows: arm axes in link6 frame", "rows: connects + coupling", "rows: limits",
         "M / RNE waves: own share (CRB / bias)", "M / RNE waves: arm meshes' broad phase", "(unused)", "M / RNE waves: waiting at S2",
         "M / RNE waves: waiting for q (S1)", "M / RNE waves: S2 -> S4",
         "cube wave: waiting for q (S1)", "cube wave: collision (rest)", "cube wave: solve + finish", "cube wave: waiting at S4",
         "robot wave: waiting at S2",
         "collision: cube frame, pair numbers", "collision: arm chain + arm meshes on table / ground", "collision: pad frames, ground plane",
         "collision: table - pads", "collision: table - cube, pads - cube", "collision: park the mesh phase's slots",
         "coop: env data, twist columns", "coop: rows", "coop: H0, g0", "coop: residuals, active set", "coop: assembly (LDS window)",
         "coop: gradient + LDL", "coop: solves + transpose", "coop: consistency check", "coop: line search", "coop: hand back", "coop: idle at S5",
         "cube wave: waiting at S1c", "cube wave: flags + solver numbers", "cube wave: waiting at S2", "M / RNE waves: waiting at S1b", "M / RNE waves: mesh phase",
         "M / RNE waves: waiting at S1c", "M / RNE waves: solver numbers"]
COUNTS = ["robot sub-steps", "robot Newton iterations", "robot line searches", "cube Newton iterations", "cube line searches",
          "coupled solves", "coupled Newton iterations", "coupled line searches", "wave-max contacts (per collision pass)",
          "coop active rows (sum over iterations)", "coop line-search evaluations", "coop solves whose carried active set was confirmed at once", "coop solves at the 50-iteration cap", "coop solves that started from a carried active set",
          "pair solves whose carried set was wrong", "wrong rows: joint limits", "wrong rows: static geom - robot contacts", "wrong rows: contacts of the cube",
          "wrong rows: missing from the carried set", "wrong rows: surplus in the carried set",
          "pair solves whose final set = the set of the last sub-step", "... = the set of two sub-steps ago", "... = the set of two sub-steps ago and not the last one's",
          "mesh pairs in the narrow phase", "mesh pairs that touch", "... decided by the face shortcut", "mesh pairs separated by a box axis", "... by a polytope face", "... by an edge axis",
          "contacts on an edge axis", "contacts on a polytope face", "mesh pairs against the cube"]
fresh = "--fresh-actions" in sys.argv
mocap = "--pnp-mocap" in sys.argv
grasp = "--grasp" in sys.argv           # PickAndPlace joint with every env holding the cube (scripted grasp state)
L = _abi.load()
n = 8192
pnpik = "--pnp-ik" in sys.argv         # PickAndPlace, IK controller, random policy only
pnpj = "--pnp-joint" in sys.argv       # PickAndPlace, joint controller, cube resting
for obj, ctrl, k in (((True, "mocap", 40),) if mocap else ((True, "joint", 20),) if grasp else ((True, "IK", 20),) if pnpik else ((True, "joint", 100),) if pnpj else ((False, "joint", 200), (False, "IK", 50), (True, "joint", 100), (True, "IK", 20))):
    envs = MyCobotVecEnv(n, has_object=obj, controller_type=ctrl, reward_type="dense", max_episode_steps=10 ** 9 if grasp else 50)
    envs.reset(seed=0)
    a = torch.rand(n, envs.action_dim, device="cuda") * 2 - 1
    if grasp:
        from mycobotgym_amd.scenarios import grasp_state
        st = grasp_state(n, seed=0)
        a = torch.as_tensor(st.pop("action"), device="cuda")
        envs.set_state(**st)
    for _ in range(5 if grasp else 60): envs.step(a)
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
