"""An INDEPENDENT checker of the collision rules the CPU oracle and the HIP kernels share (test infrastructure).

Nothing here is shared with oracle/mco_collision.c or csrc/mcg_cube.hpp: convex shapes are vertex sets, and the only tools are
scipy's ConvexHull and numpy.  For two convex polytopes A, B the Minkowski difference B - A = conv{b - a} contains the origin iff
they overlap; then the distance from the origin to its nearest facet is the penetration depth (the smallest translation that
separates them) and that facet's normal the direction of least penetration.  For two boxes the facet normals of B - A are exactly
the 15 separating-axis candidates of mjc_BoxBox, so this is the quantity MuJoCo's axis search minimises.

What is checked for a list of reported contacts (oracle's or kernels'), pair by pair:
  * presence: a pair reports contacts iff the exact test says the shapes overlap (beyond a touching tolerance);
  * normal: the overlap of the two shapes ALONG the reported normal is within 5 % (the rule's face-axis preference) of the exact depth,
    and equals the deepest reported point's depth;
  * points: every reported point lies in both shapes (inflated by the depth), at most 8 per pair, dist < 0.
The mesh geoms' collision polytopes against a box (all fourteen meshes on the table / the cube) are held to the same exact rule since
round 4: one contact iff the shapes overlap, its depth the overlap along its normal and within the 5 % rule of the exact depth.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import ConvexHull, QhullError

TOUCH = 1e-9          # depths below this are "touching": either answer is accepted


def box_vertices(pos, mat, half):
    s = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64) * np.asarray(half)
    return np.asarray(pos) + s @ np.asarray(mat).reshape(3, 3).T


def poly_vertices(pos, mat, verts):
    return np.asarray(pos) + np.asarray(verts) @ np.asarray(mat).reshape(3, 3).T


def overlap_along(VA, VB, n):
    """Overlap of the projections on n (n from A to B): > 0 iff they overlap along n."""
    return float((VA @ n).max() - (VB @ n).min())


def mtd(VA, VB):
    """(depth, normal A->B).  depth > 0: overlapping, the least translation of B along +normal that separates; depth <= 0: separated
    (-depth is a lower bound of the gap: the largest facet distance)."""
    D = (VB[None, :, :] - VA[:, None, :]).reshape(-1, 3)
    try:
        hull = ConvexHull(D)
    except QhullError:                      # degenerate (flat) difference: jiggle
        hull = ConvexHull(D, qhull_options="QJ")
    off = hull.equations[:, 3]              # n.x + off <= 0 inside
    nrm = hull.equations[:, :3]
    if (off > 0).any():                     # origin outside: separated
        k = int(np.argmax(off))
        return -float(off[k]), -nrm[k]
    k = int(np.argmin(-off))
    # the facet's outward normal n points from the origin to the facet of B - A: moving B by -depth*n ... separation direction A->B is -n
    return float(-off[k]), -nrm[k]


def in_box(p, pos, mat, half, slack):
    loc = np.asarray(mat).reshape(3, 3).T @ (np.asarray(p) - np.asarray(pos))
    return bool(np.all(np.abs(loc) <= np.asarray(half) + slack))


def ground_box(V):
    """mjc_PlaneBox on z = 0: every vertex below the plane -> (dist, pos) with pos midway between vertex and plane."""
    out = []
    for v in V:
        if v[2] < 0:
            out.append((float(v[2]), np.array([v[0], v[1], 0.5 * v[2]])))
    return out


def check_box_pair(VA, VB, boxA, boxB, contacts, what=""):
    """contacts: list of (dist, pos[3], normal[3]) the rule reported for this pair (normal from A to B).  Returns a dict of measurements."""
    depth, n_star = mtd(VA, VB)
    res = {"exact_depth": depth, "n": len(contacts)}
    if not contacts:
        assert depth < TOUCH, f"{what}: shapes overlap by {depth:.3e} but no contact was reported"
        return res
    assert depth > -TOUCH, f"{what}: {len(contacts)} contacts reported but the shapes are {-depth:.3e} apart"
    assert len(contacts) <= 8, what
    n = np.asarray(contacts[0][2], dtype=np.float64)
    assert abs(np.linalg.norm(n) - 1) < 1e-9, what
    for c in contacts:
        assert np.allclose(c[2], n, atol=1e-12), f"{what}: one pair, two normals"
        assert c[0] < 0, what
    along = overlap_along(VA, VB, n)
    deepest = max(-c[0] for c in contacts)
    res.update(along=along, deepest=deepest)
    assert along <= 1.05 * max(depth, 0) + 1e-9, f"{what}: overlap along the reported normal {along:.6e} against the exact depth {depth:.6e}"
    # (the deepest vertex of the incident face may be clipped away by the reference face's outline: then the points are shallower)
    assert deepest <= along + 1e-9, f"{what}: deepest point {deepest:.6e} beyond the overlap along the normal {along:.6e}"
    for c in contacts:
        slack = along + 1e-9
        assert in_box(c[1], *boxA, slack) and in_box(c[1], *boxB, slack), f"{what}: contact point outside the shapes"
    return res


# ----------------------------------------------------------------------------------------------- one environment's scene
from mycobotgym_amd.model import polytope as _pt

MESHES = _pt.MESH_NAMES
_POLYS = None


def polytopes():
    global _POLYS
    if _POLYS is None:
        _POLYS = _pt.unpack(_pt.load_asset()[0])
    return _POLYS


class Scene:
    """World-frame shapes of one environment from geom poses (kinematics are pinned separately: known answers, Appendix E).  The mesh
    geoms' polytopes come from the asset in geom (= STL) coordinates; every mesh geom sits at its body's origin."""

    def __init__(self, tab, spec, xpos, xmat, geom_xpos, geom_xmat):
        gi = tab["geom_name"].index
        def box(g):
            return (np.asarray(geom_xpos[g]), np.asarray(geom_xmat[g]).reshape(3, 3), np.asarray(tab["geom_size"][g], dtype=np.float64))
        self.table = box(1); assert tab["geom_type"][1] == 6
        self.cube = box(gi("object0"))
        self.pad = [box(gi("right_finger_layer")), box(gi("left_finger_layer"))]
        self.V = {"table": box_vertices(*self.table), "cube": box_vertices(*self.cube),
                  "pad0": box_vertices(*self.pad[0]), "pad1": box_vertices(*self.pad[1])}
        P = polytopes()
        for m, name in enumerate(MESHES):
            g = [k for k in range(tab["ngeom"]) if tab["geom_type"][k] == 7 and tab["geom_mesh"][k] == name][0]
            self.V[f"mesh{m}"] = poly_vertices(geom_xpos[g], geom_xmat[g], P[m]["verts"])
        self.boxes = {"table": self.table, "cube": self.cube, "pad0": self.pad[0], "pad1": self.pad[1]}


def check_scene(sc: Scene, contacts: dict, stats: dict, what=""):
    """contacts: {(a, b): [(dist, pos, normal a->b), ...]} with a in {"ground", "table", "pad0", "pad1", "mesh0".."mesh13"} and b the
    other shape's key in Scene.V.  Every pair is asserted against the exact rule."""
    movers_static = ["cube", "pad0", "pad1"] + [f"mesh{m}" for m in range(len(MESHES))]
    for b in movers_static:
        # ground plane: exact by enumeration
        got = contacts.get(("ground", b), [])
        V = sc.V[b]
        if b in ("cube", "pad0", "pad1"):
            want = ground_box(V)
            assert len(got) == len(want), f"{what} ground-{b}: {len(got)} contacts reported, {len(want)} vertices below the plane"
            for (d, p, n) in got:
                assert np.allclose(n, [0, 0, 1]) and any(abs(d - w[0]) < 1e-12 and np.allclose(p, w[1], atol=1e-12) for w in want), f"{what} ground-{b}"
        else:
            lo = float(V[:, 2].min())
            assert (len(got) > 0) == (lo < 0) or abs(lo) < TOUCH, f"{what} ground-{b}: lowest vertex {lo:.3e}, {len(got)} contacts"
            assert len(got) <= 1, f"{what} ground-{b}"
            for (d, p, n) in got:
                assert abs(d - lo) < 1e-12 and np.allclose(n, [0, 0, 1]), f"{what} ground-{b}"
                k = int(np.argmin(V[:, 2])); assert np.allclose(p, [V[k, 0], V[k, 1], 0.5 * lo], atol=1e-12), f"{what} ground-{b}: point"
        # table
        got = contacts.get(("table", b), [])
        if b in ("cube", "pad0", "pad1"):
            r = check_box_pair(sc.V["table"], V, sc.table, sc.boxes[b], got, f"{what} table-{b}")
            stats.setdefault("box_pairs", []).append((r["exact_depth"], r.get("along", 0.0), r["n"]))
        else:
            check_polytope(sc.V["table"], V, got, stats, f"{what} table-{b}", sc.table)
    for sd in (0, 1):
        got = contacts.get((f"pad{sd}", "cube"), [])
        r = check_box_pair(sc.V[f"pad{sd}"], sc.V["cube"], sc.pad[sd], sc.cube, got, f"{what} pad{sd}-cube")
        stats.setdefault("box_pairs", []).append((r["exact_depth"], r.get("along", 0.0), r["n"]))
    for m in range(len(MESHES)):
        check_polytope(sc.V[f"mesh{m}"], sc.V["cube"], contacts.get((f"mesh{m}", "cube"), []), stats, f"{what} {MESHES[m]}-cube", sc.cube)


def check_polytope(VA, VB, got, stats, what, box):
    """A polytope pair: ONE contact iff the shapes overlap; its depth is the overlap along its normal and IS the exact depth (no face-axis
    preference for these pairs); its point lies within a depth of the box and of the polytope's hull."""
    # POLY_TOL: the STL vertices are float32 and the triangles of one flat CAD face differ by ~1e-6 rad; the tables merge them into one face
    # whose plane is the mean normal at the vertices' support (polytope.py: faces_and_edges), up to 1e-6 rad x a few cm off a hull facet
    POLY_TOL = 2e-7
    depth, _ = mtd(VA, VB)
    assert len(got) <= 1, what
    if not got:
        assert depth < POLY_TOL, f"{what}: the shapes overlap by {depth:.3e} but no contact was reported"
        stats.setdefault("poly_none", []).append(depth)
        return
    assert depth > -POLY_TOL, f"{what}: a contact of depth {-got[0][0]:.3e} reported but the shapes are {-depth:.3e} apart"
    rep = -got[0][0]
    n = np.asarray(got[0][2])
    along = overlap_along(VA, VB, n)
    assert rep > 0 and abs(np.linalg.norm(n) - 1) < 1e-9, what
    assert abs(rep - along) < POLY_TOL, f"{what}: reported depth {rep:.6e}, overlap along the reported normal {along:.6e}"
    assert abs(rep - depth) < POLY_TOL, f"{what}: reported depth {rep:.6e} against the exact depth {depth:.6e}"
    assert in_box(got[0][1], *box, rep + POLY_TOL), f"{what}: contact point outside the box"
    poly = VB if box is not None and len(VA) == 8 and np.allclose(VA, box_vertices(*box)) else VA
    hull = ConvexHull(poly)
    assert (hull.equations[:, :3] @ np.asarray(got[0][1]) + hull.equations[:, 3]).max() <= rep + POLY_TOL, f"{what}: contact point outside the polytope"
    stats.setdefault("poly_exact", []).append((rep, depth))


def summarize(stats):
    out = []
    bp = np.array(stats.get("box_pairs", [])).reshape(-1, 3)
    if len(bp):
        touching = bp[bp[:, 2] > 0]
        out.append(f"box pairs checked {len(bp)} ({len(touching)} in contact; overlap along the reported normal / exact depth: max "
                   f"{(touching[:, 1] / np.maximum(touching[:, 0], 1e-300)).max() if len(touching) else 0:.4f})")
    ex = np.array(stats.get("poly_exact", [])).reshape(-1, 2)
    out.append(f"polytope pairs in contact {len(ex)} (reported / exact depth: max {(ex[:, 0] / np.maximum(ex[:, 1], 1e-300)).max() if len(ex) else 0:.4f}, "
               f"deepest {ex[:, 0].max() if len(ex) else 0:.2e}); false contacts 0, missed overlaps 0 (asserted); separated pairs confirmed {len(stats.get('poly_none', []))}")
    return "; ".join(out)


def oracle_contacts(tab, raw, ncon):
    """The oracle's contact list (28 doubles per contact) -> the dict check_scene takes.  Duplicate mesh geoms give duplicate contacts: one kept."""
    gname, gbody, bname = tab["geom_name"], tab["geom_body"], tab["body_name"]
    out = {}
    for c in range(ncon):
        ints = raw[c, 26:28].copy().view(np.int32)
        g1, g2 = int(ints[1]), int(ints[2])
        key = (_shape_key(tab, g1), _shape_key(tab, g2))
        item = (float(raw[c, 0]), raw[c, 1:4].copy(), raw[c, 4:7].copy())
        lst = out.setdefault(key, [])
        if not any(abs(item[0] - o[0]) < 1e-15 and np.allclose(item[1], o[1], atol=1e-15) for o in lst):
            lst.append(item)
    return out


def _shape_key(tab, g):
    name = tab["geom_name"][g]
    if tab["geom_type"][g] == 0: return "ground"
    if g == 1: return "table"
    if name == "object0": return "cube"
    if name == "right_finger_layer": return "pad0"
    if name == "left_finger_layer": return "pad1"
    mesh = tab["geom_mesh"][g]
    if mesh in MESHES: return f"mesh{MESHES.index(mesh)}"
    return f"other:{name or mesh}"


# pair types of the kernels' list (csrc/mcg_cube.hpp)
def kernel_contacts(count, dist, pos, normal, typ):
    """One env's entries of MyCobotVecEnv.debug_contacts() -> the dict check_scene takes.  Pair types (csrc/mcg_cube.hpp): 0 static-cube,
    1 / 2 right / left pad-cube, 3 / 4 static-right / left pad, 5 + m static-mesh m, 19 + m mesh m-cube."""
    out = {}
    for c in range(int(count)):
        t = int(typ[c]); n = np.asarray(normal[c], dtype=np.float64); p = np.asarray(pos[c], dtype=np.float64)
        if t in (1, 2): key = (f"pad{t - 1}", "cube")
        elif t >= 19: key = (f"mesh{t - 19}", "cube")
        else:
            mover = "cube" if t == 0 else (f"pad{t - 3}" if t in (3, 4) else f"mesh{t - 5}")
            ground = np.allclose(n, [0, 0, 1]) and p[2] < 0.1          # the ground plane's contacts sit at z ~ 0, the table top's at 0.2
            key = ("ground" if ground else "table", mover)
        out.setdefault(key, []).append((float(dist[c]), p, n))
    return out
