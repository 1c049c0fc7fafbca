#!/usr/bin/env python3
"""Development helper: time a bench.py workload and, when MCG_LIB is a -DMCG_STAGE_CLOCKS build, print the cooperative-solve counters.

    [MCG_COOP_PAIR=0] [MCG_LIB=ab/clocks.so] python tools/coop_probe.py pnp-IK [steps] [warmup]
"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from mycobotgym_amd import MyCobotVecEnv, _abi

case = sys.argv[1] if len(sys.argv) > 1 else "pnp-IK"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 10
task, controller, dr, grasp = bench.CASES[case]
n = 8192
envs = MyCobotVecEnv(n, has_object=task == "pnp", controller_type=controller, reward_type="dense", seed=0,
                     max_episode_steps=10 ** 9 if grasp else 50)
envs.reset(seed=0)
g = torch.Generator(device="cuda"); g.manual_seed(1234)
if not grasp:
    envs.set_state(elapsed=torch.randint(0, 50, (n,), device="cuda", generator=g, dtype=torch.int32))
pool = torch.rand(16, n, envs.action_dim, device="cuda", generator=g) * 2 - 1
if grasp:
    from mycobotgym_amd.scenarios import grasp_state
    st = grasp_state(n, seed=0); act = torch.as_tensor(st.pop("action"), device="cuda"); envs.set_state(**st)
    pool = act.unsqueeze(0).repeat(16, 1, 1).contiguous()
L = _abi.load()
has_clk = hasattr(L, "mcg_debug_stage_clocks")
for t in range(warm): envs.step_async(pool[t % 16])
torch.cuda.synchronize()
if has_clk:
    out = (C.c_ulonglong * 256)(); L.mcg_debug_stage_clocks(out, 1)
    if hasattr(L, 'mcg_debug_wg_stat'): L.mcg_debug_wg_stat(None, 0, 1)
import subprocess, threading, json as _json
samples = []; stop = False
def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = _json.loads(o); c = list(d.values())[0]
            samples.append((c.get("sclk clock speed:", "?"), c.get("Average Graphics Package Power (W)", c.get("Current Socket Graphics Package Power (W)", "?"))))
        except Exception as ex:
            samples.append((str(ex)[:40], "?"))
        time.sleep(0.05)
th = threading.Thread(target=poll); th.start()
per = []
for t in range(steps):
    t0 = time.perf_counter(); envs.step_async(pool[t % 16]); torch.cuda.synchronize(); per.append((time.perf_counter() - t0) * 1e3)
stop = True; th.join()
print('   smi samples (sclk, power):', samples[:3], '...', samples[-6:])
print(f"{case} coop_pair={os.environ.get('MCG_COOP_PAIR', '1')} lib={os.path.basename(os.environ.get('MCG_LIB', 'plain'))}: "
      f"mean {sum(per) / len(per):.3f} ms/step; first {per[0]:.2f} min {min(per):.2f} max {max(per):.2f}; every 5th: " + " ".join(f"{x:.1f}" for x in per[::5]))
if has_clk:
    import re
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage_clocks.py")).read()
    nst = len(eval(re.search(r"NAMES = (\[.*?\])\nCOUNTS", src, re.S).group(1)))
    cn = eval(re.search(r"COUNTS = (\[.*?\])\n", src, re.S).group(1))
    L.mcg_debug_stage_clocks(out, 1)
    cnt = list(out)[nst:nst + len(cn)]
    wg_sub = (n // 32) * steps * (100 if controller == "IK" else 20)
    print("   per workgroup-sub-step: " + ", ".join(f"{a} {v / wg_sub:.2f}" for a, v in zip(cn, cnt) if v))
    if hasattr(L, "mcg_debug_wg_stat"):
        import numpy as np
        w = (C.c_ulonglong * (256 * 4))(); L.mcg_debug_wg_stat(w, 256 * 4, 1)
        w = np.array(list(w), dtype=np.float64).reshape(256, 4)
        order = np.argsort(-w[:, 0])
        print("   per workgroup (clocks/launch, coop solves/launch, cube iterations/launch, cube line searches/launch): mean",
              " ".join(f"{x:.0f}" for x in w.mean(0) / steps))
        for k in list(order[:5]) + list(order[-2:]):
            print(f"      wg {k:3d}: " + " ".join(f"{x:.0f}" for x in w[k] / steps))
    print("   raw counts: " + ", ".join(f"{a} {v}" for a, v in zip(cn, cnt) if v and ("9+" in a or "cap" in a or "solves" in a)))
