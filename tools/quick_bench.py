"""Quick on-GPU timing of the step kernel (development helper; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from mycobotgym_amd import MyCobotVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for ctrl in ("joint", "IK"):
    envs = MyCobotVecEnv(n, has_object=False, controller_type=ctrl, reward_type="dense")
    envs.reset(seed=0)
    a = torch.rand(n, envs.action_dim, device="cuda") * 2 - 1
    for _ in range(5): envs.step(a)
    torch.cuda.synchronize()
    ms = envs.time_steps(a, 20)
    print(f"{ctrl}: n={n} {ms/20:.3f} ms/step  {n/(ms/20)*1e3:.3e} env-steps/s  {n/(ms/20)*1e3*(20 if ctrl=='joint' else 100):.3e} substeps/s")
    envs.close()
