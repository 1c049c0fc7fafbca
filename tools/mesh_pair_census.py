#!/usr/bin/env python3
"""Which mesh pairs reach the exact narrow phase, and how many of them touch?  (CPU, oracle states; development helper.)

Rolls the CPU oracle under a uniformly random policy (PickAndPlace, IK controller), evaluates the kernels' per-lane broad phase
(csrc/mcg_cube.hpp: mesh_broad -- the polytope's bounding box in its body's frame against the ground, the table (6 axes) and the cube
(6 axes)) on the oracle's geom poses, and reads the oracle's contact list for the pairs that do touch.  Prints, per mesh and target,
candidates and contacts per environment; the narrow phase costs one wave-pass per candidate.

    python tools/mesh_pair_census.py [n_envs] [env_steps]
"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from tests.common import make_oracle, load_json, table_name
from mycobotgym_amd.model.mjcf import _np_model
from mycobotgym_amd.model.specialize import specialize
from mycobotgym_amd.model import polytope as pt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
tab = load_json(table_name(True))
spec = specialize(_np_model(tab))
mbox = np.asarray(spec["mesh_box"])
tp, th = np.asarray(spec["table_pos"]), np.asarray(spec["table_half"])
hc = np.asarray(spec["cube_half"])
body_polys = pt.unpack(np.asarray(spec["polytopes"])); stl_polys = pt.unpack(pt.load_asset()[0])
frames = []                                     # x_body = R x_geom + pw, recovered from the two vertex tables
for B, S in zip(body_polys, stl_polys):
    A = np.hstack([S["verts"], np.ones((len(S["verts"]), 1))])
    X = np.linalg.lstsq(A, B["verts"], rcond=None)[0]
    frames.append((X[:3].T, X[3]))
gname = tab["geom_name"]; gmesh = tab["geom_mesh"]; gtype = tab["geom_type"]
mesh_names = tab.get("mesh_name")
geoms = []
for nm in pt.MESH_NAMES:
    gs = [g for g in range(tab["ngeom"]) if gtype[g] == 7 and tab["geom_contype"][g] and tab["geom_conaffinity"][g]
          and (mesh_names[gmesh[g]] if mesh_names else gmesh[g]) == nm]
    geoms.append(gs)
gcube = gname.index("object0")
ora = make_oracle(n, has_object=True, controller_type="IK", reward_type="dense", seed=0)
ora.reset(seed=0)
rng = np.random.default_rng(0)
cand = np.zeros((pt.NMESH, 3)); hit = np.zeros((pt.NMESH, 3)); per_env = []; tot = 0
ngeom = tab["ngeom"]
for t in range(steps):
    ora.step(rng.uniform(-1, 1, (n, ora.act_dim)).astype(np.float32))
    c_env = np.zeros(n)
    for i in range(n):
        d = ora.data(i)
        xp = d.get("geom_xpos", (ngeom, 3)); xm = d.get("geom_xmat", (ngeom, 9)).reshape(ngeom, 3, 3)
        Rc, pc = xm[gcube], xp[gcube]
        for mi in range(pt.NMESH):
            g = geoms[mi][0]; Rf, pw = frames[mi]
            R = xm[g] @ Rf.T; p = xp[g] - R @ pw                      # world <- engine body
            bx = mbox[mi]
            c = p + R @ bx[:3]; e = np.abs(R) @ bx[3:]
            ground = c[2] - e[2] < 0
            table = np.all(np.abs(c - tp) <= th + e)
            rel = R.T @ (c - tp); rad = np.abs(R.T) @ th
            table = table and np.all(np.abs(rel) <= bx[3:] + rad)
            tt = pc - c
            M = R.T @ Rc                                               # box axes x cube axes
            near = np.all(np.abs(R.T @ tt) <= bx[3:] + np.abs(M) @ hc) and np.all(np.abs(Rc.T @ tt) <= hc + np.abs(M.T) @ bx[3:])
            cand[mi] += (ground, table, near); c_env[i] += ground + table + near
        ncon = int(d.get("ncon", (1,), np.int32)[0]); raw = d.get("contact", (64, 28))
        seen = set()
        for k in range(ncon):
            ints = raw[k, 26:28].copy().view(np.int32); g1, g2 = int(ints[1]), int(ints[2])
            for mi in range(pt.NMESH):
                for (a, b) in ((g1, g2), (g2, g1)):
                    if a == geoms[mi][0]:
                        tg = 2 if b == gcube else (0 if gtype[b] == 0 else 1)
                        if (mi, tg) not in seen: seen.add((mi, tg)); hit[mi, tg] += 1
    per_env.append(c_env); tot += n
per_env = np.concatenate(per_env)
print(f"{tot} environment states (random IK policy, ends of env-steps)")
print(f"{'mesh':20s} {'cand ground':>12s} {'table':>8s} {'cube':>8s} | {'hit ground':>11s} {'table':>8s} {'cube':>8s}")
for mi, nm in enumerate(pt.MESH_NAMES):
    print(f"{nm:20s} {cand[mi,0]/tot:12.4f} {cand[mi,1]/tot:8.4f} {cand[mi,2]/tot:8.4f} | {hit[mi,0]/tot:11.4f} {hit[mi,1]/tot:8.4f} {hit[mi,2]/tot:8.4f}")
print(f"candidates per environment {cand.sum()/tot:.3f} (per 32: {32*cand.sum()/tot:.1f}); contacts per environment {hit.sum()/tot:.3f}")
print("environments by candidate count:", {int(k): int(v) for k, v in zip(*np.unique(per_env, return_counts=True))})
