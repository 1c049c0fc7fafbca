import sys, time, torch
sys.path.insert(0, "/root/repo")
from mycobotgym_amd import MyCobotVecEnv
for n in (64, 8192):
    envs = MyCobotVecEnv(n, has_object=False, controller_type="joint", reward_type="dense")
    envs.reset(seed=0)
    pool = torch.rand(16, n, envs.action_dim, device="cuda") * 2 - 1
    for t in range(2000): envs.step_async(pool[t % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(3000): envs.step_async(pool[t % 16])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: host enqueue {1e3*(t1-t0)/3000:.4f} ms/step, incl. drain {1e3*(t2-t0)/3000:.4f} ms/step")
    ms = envs.time_steps(pool[0], 500) / 500
    print(f"   time_steps (C loop of launches, one action batch): {ms:.4f} ms/step")
    envs.close()
