"""The collision rules restated in the oracle, checked against an independent exact computation (tests/indep_collision.py: scipy
ConvexHull of the Minkowski difference, vertex enumeration) at random and near-degenerate poses.  CPU only."""
import numpy as np
import pytest

from tests.common import load_json
from tests import indep_collision as ic


def _oracle_scene(tab, spec, d, q):
    d.set_state(qpos=q, qvel=np.zeros(18)); d.forward()
    nb, ng = tab["nbody"], tab["ngeom"]
    sc = ic.Scene(tab, spec, d.get("xpos", (nb, 3)), d.get("xmat", (nb, 9)), d.get("geom_xpos", (ng, 3)), d.get("geom_xmat", (ng, 9)))
    n = int(d.get("ncon", (1,), np.int32)[0])
    return sc, ic.oracle_contacts(tab, d.get("contact", (64, 28)), n), n


@pytest.fixture(scope="module")
def setup():
    from oracle import pyoracle as po
    from mycobotgym_amd.model.mjcf import _np_model
    from mycobotgym_amd.model.specialize import specialize
    tab = load_json("mycobot280")
    spec = specialize(_np_model(tab))
    d = po.OracleData(po.OracleModel(tab, enable_contact=True, scope_geom=tab["geom_name"].index("object0")))
    return tab, spec, d


def _quat(rng, small=None):
    if small is None:
        q = rng.normal(size=4)
    else:
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = small
        q = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
    return q / np.linalg.norm(q)


def test_cube_on_table_and_ground_random_and_degenerate(setup):
    """Cube against the table (box-box) and the ground (plane-box): random orientations, faces parallel to the table's (the state in
    which round 2 found the shared edge-axis fault), parallel edges, a vertex over an edge, hanging over the table's rim."""
    tab, spec, d = setup
    rng = np.random.default_rng(0)
    stats = {}; seen = 0
    q0 = np.array(tab["qpos0"], float)
    for k in range(600):
        q = q0.copy()
        mode = k % 6
        if mode == 0:   q[12:15] = [rng.uniform(-0.15, 0.15), rng.uniform(-0.1, 0.2), 0.2 + rng.uniform(0.004, 0.0175)]; q[15:19] = _quat(rng)
        elif mode == 1: q[12:15] = [rng.uniform(-0.15, 0.15), rng.uniform(-0.1, 0.2), 0.21 - rng.uniform(0, 2e-3)]; q[15:19] = _quat(rng, rng.choice([0, 1e-9, 1e-6, 3e-4]))
        elif mode == 2: q[12:15] = [0.2 + rng.uniform(-0.012, 0.012), rng.uniform(-0.1, 0.2), 0.2 + rng.uniform(0.002, 0.012)]; q[15:19] = _quat(rng, rng.uniform(0, 0.5))   # over the rim
        elif mode == 3: q[12:15] = [rng.uniform(0.3, 0.5), rng.uniform(-0.3, 0.3), rng.uniform(0.004, 0.017)]; q[15:19] = _quat(rng)                                       # on the ground
        elif mode == 4: q[12:15] = [0.2 + 0.01 - rng.uniform(0, 2e-3), rng.uniform(-0.1, 0.1), rng.uniform(0.009, 0.02)]; q[15:19] = _quat(rng, rng.choice([0, 1e-7]))      # ground and the table's side
        else:           q[12:15] = [rng.uniform(-0.15, 0.15), rng.uniform(-0.1, 0.2), 0.2 + rng.uniform(0.009, 0.0172)]; q[15:19] = _quat(rng, np.pi / 4 + rng.normal(0, 1e-3))   # edge down
        sc, con, n = _oracle_scene(tab, spec, d, q)
        if int(d.get("ndrop", (1,), np.int32)[0]) > 0: continue                       # (the cap)
        ic.check_scene(sc, con, stats, f"pose {k} mode {mode}")
        seen += n > 0
    print("\n" + ic.summarize(stats) + f"; poses with contacts {seen}")
    assert seen > 300


def test_arm_and_gripper_poses(setup):
    """Random arm poses near the table and gripper poses around the cube: pads on the table / the ground / the cube (box-box), and the
    fourteen mesh polytopes against the table, the ground and the cube: zero false contacts, zero missed overlaps, depths within the rule."""
    tab, spec, d = setup
    from mycobotgym_amd.scenarios import grasp_state
    rng = np.random.default_rng(1)
    stats = {}; seen = 0
    q0 = np.array(tab["qpos0"], float)
    g = np.asarray(grasp_state(64, seed=0)["qpos"]); g = g.T if g.shape[0] == 19 else g
    for k in range(1500):
        if k % 3 == 0:
            q = g[rng.integers(len(g))].copy()
            q[:6] += rng.normal(0, 0.05, 6); q[6] = q[8] = np.clip(q[6] + rng.normal(0, 0.15), 0, 0.7)
        else:
            q = q0.copy(); q[:6] = rng.uniform(-2.5, 2.5, 6); q[6] = q[8] = rng.uniform(0, 0.7)
        sc, con, n = _oracle_scene(tab, spec, d, q)
        if int(d.get("ndrop", (1,), np.int32)[0]) > 0: continue
        if any(k[0].startswith("other") or k[1].startswith("other") for k in con): continue
        ic.check_scene(sc, con, stats, f"pose {k}")
        seen += n > 4
    print("\n" + ic.summarize(stats) + f"; poses with robot contacts {seen}")
    assert seen > 100
    assert len(stats.get("poly_exact", [])) > 200


def test_face_bound_of_the_mesh_narrow_phase():
    """The bound that lets the kernels skip the P and E families (csrc/mcg_mesh.hpp: mesh_box): when the polytope's
    deepest vertex along the box axis of least penetration keeps clearances >= that depth to the box's faces along the other two axes,
    the box axis IS the axis of least penetration of the two shapes.  Checked on the collision polytopes of the fourteen meshes against
    table-sized and cube-sized boxes with the independent exact depth (qhull of the Minkowski difference), in the middle of a face and
    across the rim, where the bound must not be claimed."""
    from mycobotgym_amd.model import polytope as pt
    polys = pt.unpack(pt.load_asset()[0])
    rng = np.random.default_rng(7)
    held = not_held = 0
    for trial in range(420):
        m = trial % pt.NMESH
        V = polys[m]["verts"] @ ic_rot(rng).T                                   # the mesh at the origin, rotated
        half = np.array([0.2, 0.25, 0.2]) if trial % 3 else np.array([0.02, 0.02, 0.02])
        Rb = ic_rot(rng) if trial % 2 else np.eye(3)
        # the box's +z face a little under the polytope's lowest point along the box's z axis, shifted sideways by up to a box width
        z = Rb[:, 2]; low = V[np.argmin(V @ z)]
        depth_in = rng.uniform(1e-4, 6e-3)
        shift = rng.uniform(-1.1, 1.1, 2) * half[:2]
        pb = low + Rb[:, 0] * shift[0] + Rb[:, 1] * shift[1] - z * (half[2] - depth_in)
        # B family: depth along each of the six face axes, the deepest vertex of the winner
        loc = (V - pb) @ Rb                                                    # box coordinates
        dep = np.concatenate([half - loc.min(0), loc.max(0) + half])           # axis +j: polytope's min against the + face; -j: its max against the - face
        if dep.min() <= 0: continue                                            # separated by a box axis
        k = int(np.argmin(dep)); j = k % 3; d = float(dep[k])
        p = loc[np.argmin(loc[:, j])] if k < 3 else loc[np.argmax(loc[:, j])]
        g = min(half[i] - abs(p[i]) for i in range(3) if i != j)
        exact, _ = ic.mtd(ic.box_vertices(pb, Rb, half), V)
        if g >= d + 1e-9:
            assert abs(exact - d) < 2e-7, (pt.MESH_NAMES[m], trial, d, exact, g)     # no other axis is shallower, none separates
            held += 1
        else:
            assert exact <= d + 2e-7                                           # (a box axis is always an upper bound of the depth)
            not_held += 1
    assert held > 100 and not_held > 50, (held, not_held)


def ic_rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x)], [2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)]])
