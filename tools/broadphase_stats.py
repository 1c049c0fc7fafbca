#!/usr/bin/env python3
"""How often does the arm-mesh broad phase of the collision pass let a mesh through?  (CPU, oracle states; development helper.)

Rolls the CPU oracle under a uniformly random policy (PickAndPlace, IK controller) and evaluates, per env and per arm mesh, the
predicates the kernel's collision pass uses (mcg_cube.hpp: prepare / hull) and a candidate replacement; prints per-env and per-wave
(32 envs, any lane) pass rates.  The kernel pays a mesh's exact test for the whole wave whenever ONE lane passes.
"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from tests.common import make_oracle, load_json, table_name
from mycobotgym_amd.model.mjcf import _np_model
from mycobotgym_amd.model.specialize import specialize

n, steps = 256, 30
tab = load_json(table_name(True))
spec = specialize(_np_model(tab))
box = np.asarray(spec["link_hull_box"]); rad = np.asarray([b[15] for b in np.asarray(spec["body"])])[:6] if False else None
body = np.asarray(spec["body"])
hull_rad = body[:6, 15]
tp, th = np.asarray(spec["table_pos"]), np.asarray(spec["table_half"])
ora = make_oracle(n, has_object=True, controller_type="IK", reward_type="dense", seed=0)
ora.reset(seed=0)
names = ["link1", "link2", "link3", "link4", "link5", "link6"]
bid = [tab["body_name"].index(b) for b in names]
rng = np.random.default_rng(0)
cnt = {k: np.zeros(8) for k in ("sphere", "bbox_z", "aabb", "sphere_w", "bbox_z_w", "aabb_w")}
tot = 0
for t in range(steps):
    ora.step(rng.uniform(-1, 1, (n, ora.act_dim)).astype(np.float32))
    xp = np.stack([ora.data(i).get("xpos", (tab["nbody"], 3)) for i in range(n)])
    xm = np.stack([ora.data(i).get("xmat", (tab["nbody"], 9)) for i in range(n)]).reshape(n, -1, 3, 3)
    res = {k: np.zeros((n, 8), bool) for k in ("sphere", "bbox_z", "aabb")}
    for p in range(8):
        b = min(p, 5)
        P, R = xp[:, bid[b]], xm[:, bid[b]]
        res["sphere"][:, p] = P[:, 2] - hull_rad[b] < tp[2] + th[2]
        c = P + np.einsum("nij,j->ni", R, box[p, :3]); e = np.einsum("nij,j->ni", np.abs(R), box[p, 3:])
        res["bbox_z"][:, p] = res["sphere"][:, p] & (c[:, 2] - e[:, 2] < tp[2] + th[2])
        over = np.all(np.abs(c - tp) <= th + e, axis=1)
        res["aabb"][:, p] = res["sphere"][:, p] & (over | (c[:, 2] - e[:, 2] < 0))
    for k in res:
        cnt[k] += res[k].sum(0)
        cnt[k + "_w"] += res[k].reshape(n // 32, 32, 8).any(1).sum(0)
    tot += 1
print("mesh:                 " + " ".join(f"{m:>8s}" for m in names + ["flange", "gripbase"]))
for k in ("sphere", "bbox_z", "aabb"):
    print(f"{k:8s} per env      " + " ".join(f"{v / (tot * n):8.3f}" for v in cnt[k]))
    print(f"{k:8s} per 32-wave  " + " ".join(f"{v / (tot * n / 32):8.3f}" for v in cnt[k + "_w"]))
