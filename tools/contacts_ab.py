#!/usr/bin/env python3
"""Development helper: do two builds of the library see the same contacts, and step the same, from the same states?

    MCG_LIB=ab/a.so python tools/contacts_ab.py dump  gpurun_out/x.pt [case] [warmup]    # states after a warm-up + contact lists + one more step
    MCG_LIB=ab/b.so python tools/contacts_ab.py check gpurun_out/x.pt [case]             # same states in the other build: compare
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from mycobotgym_amd import MyCobotVecEnv

mode, path = sys.argv[1], sys.argv[2]
case = sys.argv[3] if len(sys.argv) > 3 else "pnp-IK"
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 100
task, controller, dr, grasp = bench.CASES[case]
n = 8192
envs = MyCobotVecEnv(n, has_object=True, controller_type=controller, reward_type="dense", seed=0)
envs.reset(seed=0)
g = torch.Generator(device="cuda"); g.manual_seed(1234)
pool = torch.rand(16, n, envs.action_dim, device="cuda", generator=g) * 2 - 1
if mode == "dump":
    envs.set_state(elapsed=torch.randint(0, 50, (n,), device="cuda", generator=g, dtype=torch.int32))
    for t in range(warm): envs.step_async(pool[t % 16])
    torch.cuda.synchronize()
    st = {k: v.cpu() for k, v in envs.get_state().items()}
    kc = {k: v.cpu() for k, v in envs.debug_contacts().items()}
    obs = envs.step(pool[3])[0]
    obs = {k: v.cpu() for k, v in obs.items()} if isinstance(obs, dict) else obs.cpu()
    torch.save({"state": st, "contacts": kc, "obs": obs}, path)
    print("dumped", path, "contacts per env (mean)", kc["count"].double().mean().item())
else:
    d = torch.load(path)
    envs.set_state(**d["state"])
    kc = {k: v.cpu() for k, v in envs.debug_contacts().items()}
    ref = d["contacts"]
    bad = (kc["count"] != ref["count"]).nonzero().flatten()
    print(f"{case}: environments whose list length differs: {len(bad)} of {n}")
    for i in bad[:5].tolist():
        print("  env", i, "ref", ref["count"][i].item(), ref["type"][i][:ref["count"][i]].tolist(), "this", kc["count"][i].item(), kc["type"][i][:kc["count"][i]].tolist())
    same = kc["count"] == ref["count"]
    for k in ("dist", "pos", "normal", "D"):
        a, b = kc[k][same], ref[k][same]
        print(f"  {k}: max abs difference {(a - b).abs().max().item():.3e}")
    obs = envs.step(pool[3])[0]
    o1 = obs["observation"].cpu() if isinstance(obs, dict) else obs.cpu()
    o0 = d["obs"]["observation"] if isinstance(d["obs"], dict) else d["obs"]
    err = (o1 - o0).abs().max(dim=1).values
    print(f"  one env-step from the same state: max obs difference {err.max().item():.3e}, envs above 1e-9: {(err > 1e-9).sum().item()}")
    # who differs: environments whose list (at the start of the step) held an arm-mesh entry, against the others
    ty = ref["type"]; cnt = ref["count"]
    idx = torch.arange(ty.shape[1])[None, :] < cnt[:, None]
    arm = ((ty >= 5) & (ty < 13) & idx).any(dim=1)           # PAIR_TABLE_LINK0 .. + 7 (mcg_cube.hpp)
    for name, m in (("arm-mesh entry in the list", arm), ("none", ~arm)):
        e = err[m]
        if len(e): print(f"    {name}: {len(e)} envs, above 1e-9: {(e > 1e-9).sum().item()}, above 1e-12: {(e > 1e-12).sum().item()}, above 0: {(e > 0).sum().item()}, median {e.median().item():.2e}")
    q = torch.tensor([0.5, 0.9, 0.99, 0.999]).to(err.dtype)
    print("    quantiles of the difference (50/90/99/99.9 %):", [f"{v:.2e}" for v in torch.quantile(err, q).tolist()])
    for i in (err > 1e-9).nonzero().flatten()[:6].tolist():
        print("    env", i, "err %.3e" % err[i].item(), "types", ty[i][:cnt[i]].tolist())
