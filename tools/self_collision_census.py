#!/usr/bin/env python3
"""How often would the mesh <-> mesh pairs that this build does NOT collide touch?  (CPU, oracle states.)

Rolls the CPU oracle under a uniformly random policy (PickAndPlace, IK controller) and tests, per sampled state, every pair of mesh
geoms that MuJoCo's filter lets through (oracle/mco_collision.c: filtered -- same body, weld-parent, the eight <exclude>s of
mycobot280_main.xml:27-37, contype / conaffinity) for overlap of their collision polytopes: bounding spheres first, then the exact
answer from a linear programme (is there a point inside both sets of face planes, and how deep is the deepest one).

    python tools/self_collision_census.py [n_envs] [env_steps] [IK|joint|mocap]
"""
import os, sys
import numpy as np
from scipy.optimize import linprog
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from tests.common import make_oracle, load_json, table_name
from mycobotgym_amd.model import polytope as pt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
controller = sys.argv[3] if len(sys.argv) > 3 else "IK"
tab = load_json(table_name(True))
polys = pt.unpack(pt.load_asset()[0])                      # geom (STL) coordinates
gtype, gbody = tab["geom_type"], tab["geom_body"]
mesh_names = tab.get("mesh_name"); gmesh = tab["geom_mesh"]
geom_of = []
for nm in pt.MESH_NAMES:
    gs = [g for g in range(tab["ngeom"]) if gtype[g] == 7 and tab["geom_contype"][g] and tab["geom_conaffinity"][g]
          and (mesh_names[gmesh[g]] if mesh_names else gmesh[g]) == nm]
    geom_of.append(gs[0])
weld, parent = tab["body_weldid"], tab["body_parent"]
excl = {tuple(sorted(e)) for e in tab["excludes"]}


def filtered(g1, g2):
    b1, b2 = gbody[g1], gbody[g2]
    w1, w2 = weld[b1], weld[b2]
    if (w1 == 0 and w2 == 0) or w1 == w2: return True
    p1, p2 = weld[parent[w1]], weld[parent[w2]]
    if w1 != 0 and w2 != 0 and (p1 == w2 or p2 == w1): return True
    return tuple(sorted((b1, b2))) in excl


pairs = [(a, b) for a in range(pt.NMESH) for b in range(a + 1, pt.NMESH) if not filtered(geom_of[a], geom_of[b])]
print(f"{len(pairs)} of {pt.NMESH * (pt.NMESH - 1) // 2} mesh pairs pass the filter:",
      ", ".join(f"{pt.MESH_NAMES[a]}-{pt.MESH_NAMES[b]}" for a, b in pairs))
cen = [0.5 * (P["verts"].max(0) + P["verts"].min(0)) for P in polys]
rad = [np.linalg.norm(P["verts"] - c, axis=1).max() for P, c in zip(polys, cen)]
ora = make_oracle(n, has_object=True, controller_type=controller, reward_type="dense", seed=0)
ora.reset(seed=0)
rng = np.random.default_rng(0)
ngeom = tab["ngeom"]
near = np.zeros(len(pairs), int); touch = np.zeros(len(pairs), int); deep = np.zeros(len(pairs)); states = 0; any_touch = 0
for t in range(steps):
    ora.step(rng.uniform(-1, 1, (n, ora.act_dim)).astype(np.float32))
    for i in range(n):
        d = ora.data(i)
        xp = d.get("geom_xpos", (ngeom, 3)); xm = d.get("geom_xmat", (ngeom, 9)).reshape(ngeom, 3, 3)
        hit = False
        for k, (a, b) in enumerate(pairs):
            ga, gb = geom_of[a], geom_of[b]
            ca = xp[ga] + xm[ga] @ cen[a]; cb = xp[gb] + xm[gb] @ cen[b]
            if np.linalg.norm(ca - cb) > rad[a] + rad[b]: continue
            near[k] += 1
            # max s  s.t.  n_f . x + s <= d_f for every face of both polytopes (world frame): s > 0 <=> the interiors overlap, s = the
            # radius of the largest ball inside the intersection
            rows, rhs = [], []
            for g, P in ((ga, polys[a]), (gb, polys[b])):
                N = P["faces"][:, :3] @ xm[g].T                                    # world normals
                rows.append(np.hstack([N, np.ones((len(N), 1))])); rhs.append(P["faces"][:, 3] + N @ xp[g])
            r = linprog([0, 0, 0, -1], A_ub=np.vstack(rows), b_ub=np.concatenate(rhs), bounds=[(None, None)] * 3 + [(None, 1.0)], method="highs")
            if r.status == 0 and -r.fun > 1e-9:
                touch[k] += 1; deep[k] = max(deep[k], -r.fun); hit = True
        any_touch += hit; states += 1
print(f"{states} environment states (random {controller} policy, ends of env-steps): {any_touch} ({100.0 * any_touch / states:.2f} %) hold at least one "
      f"overlapping mesh pair")
print(f"{'pair':44s} {'spheres overlap':>16s} {'polytopes overlap':>18s} {'largest inscribed ball':>24s}")
for k in np.argsort(-touch):
    if near[k] == 0: continue
    a, b = pairs[k]
    print(f"{pt.MESH_NAMES[a] + ' - ' + pt.MESH_NAMES[b]:44s} {near[k] / states:16.4f} {touch[k] / states:18.4f} {deep[k] * 1e3:21.2f} mm")
