"""Collision polytopes of the mesh geoms (reference geoms: mycobotgym/envs/assets/mycobot280_main.xml:105-247).

MuJoCo collides the CONVEX HULL of a mesh geom (qhull at compile time: 500-1500 hull vertices for the arm links, 48-61 for the gripper's
small parts) through libccd, one contact per pair.  This build collides, per mesh, a polytope spanned by a SUBSET of the hull's vertices:

  * a small mesh (at most ``full_below`` hull vertices: the gear, finger and hinge links) keeps its whole hull -- Hausdorff distance 0;
  * a large one starts from the hull's support points in the 26 directions of a cube's faces, edges and corners and takes, one at a
    time, the hull vertex FARTHEST from the current polytope (exact point-to-polytope distance) until none is farther than ``tol``
    (1 mm): an inner approximation within ``tol`` of the hull everywhere (the Hausdorff distance, measured and stored).

For the exact separating-axis test of a polytope against a box (csrc/mcg_mesh.hpp, oracle/mco_collision.c) the polytope carries

  verts [V, 3]
  faces [F, 4]   outward unit normal n and offset d (n.x <= d inside); coplanar hull triangles merged into one face
  edges [E, 13]  an end point p of the edge, its unit direction e, u1 = n1 - (n1.n2) n2 and u2 = n2 - (n1.n2) n1 for the two adjacent
                 face normals: a direction x perpendicular to e lies in the edge's normal cone iff x.u1 >= 0 and x.u2 >= 0
                 (x = a n1 + b n2 with a, b >= 0), i.e. iff the edge is the polytope's support set along x; and the edge's length.

This is an OFFLINE tool (scipy's qhull); the tables travel as mycobotgym_amd/assets/polytopes.npz and as the generated header
csrc/polytopes_gen.h.  Nothing at run time reads the reference tree.
"""
from __future__ import annotations

import itertools
import os

import numpy as np

# order = the kernels' / the oracle's mesh index (pair types are derived from it); body = the engine body the geom rides on
MESH_NAMES = ("link1", "link2", "link3", "link4", "link5", "link6", "flange", "gripper_base",
              "right_gear_link", "right_finger_link", "left_gear_link", "left_finger_link", "right_hinge_link", "left_hinge_link")
MESH_BODY = (0, 1, 2, 3, 4, 5, 5, 5, 6, 7, 8, 9, 10, 11)
NMESH = len(MESH_NAMES)
VPAD = 64            # every table is padded to a multiple of the wave width: lane = vertex / face / edge
TOL = 1.0e-3
FULL_BELOW = 64
ASSET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets", "polytopes.npz")


def point_hull_distance(P: np.ndarray, tri: np.ndarray) -> np.ndarray:
    """Distance from each point of P [n, 3] to the surface made of the triangles tri [m, 3, 3] (closest point on each triangle,
    Ericson's region walk, vectorised over the triangles)."""
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    ab, ac = b - a, c - a
    out = np.empty(len(P))
    for i, p in enumerate(P):
        ap, bp, cp = p - a, p - b, p - c
        d1 = (ab * ap).sum(1); d2 = (ac * ap).sum(1)
        d3 = (ab * bp).sum(1); d4 = (ac * bp).sum(1)
        d5 = (ab * cp).sum(1); d6 = (ac * cp).sum(1)
        vc = d1 * d4 - d3 * d2; vb = d5 * d2 - d1 * d6; va = d3 * d6 - d5 * d4
        q = np.empty_like(a)
        done = np.zeros(len(a), bool)

        def put(mask, val):
            m = mask & ~done
            q[m] = val[m]; done[m] = True

        with np.errstate(all="ignore"):
            put((d1 <= 0) & (d2 <= 0), a)
            put((d3 >= 0) & (d4 <= d3), b)
            put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + (d1 / (d1 - d3))[:, None] * ab)
            put((d6 >= 0) & (d5 <= d6), c)
            put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + (d2 / (d2 - d6))[:, None] * ac)
            put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b + ((d4 - d3) / ((d4 - d3) + (d5 - d6)))[:, None] * (c - b))
            den = 1.0 / (va + vb + vc)
            put(np.ones(len(a), bool), a + ab * (vb * den)[:, None] + ac * (vc * den)[:, None])
        out[i] = np.sqrt(((q - p) ** 2).sum(1).min())
    return out


def select_vertices(hv: np.ndarray, tol: float = TOL, full_below: int = FULL_BELOW, vmax: int = 128):
    """Indices into the hull vertices hv, and the Hausdorff distance of their polytope to the hull."""
    from scipy.spatial import ConvexHull
    if len(hv) <= full_below:
        return list(range(len(hv))), 0.0
    dirs = np.array([d for d in itertools.product((-1.0, 0.0, 1.0), repeat=3) if any(d)])
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    sel = []
    for k in np.argmax(hv @ dirs.T, axis=0):          # first occurrence per direction, direction order
        if int(k) not in sel:
            sel.append(int(k))
    while True:
        h = ConvexHull(hv[sel])
        exc = (hv @ h.equations[:, :3].T + h.equations[:, 3]).max(1)      # > 0: outside the polytope (a lower bound of the distance)
        cand = np.where(exc > 1e-9)[0]
        if len(cand) == 0:
            return sel, 0.0
        d = point_hull_distance(hv[cand], hv[sel][h.simplices])
        k = int(np.argmax(d))
        if d[k] <= tol or len(sel) >= vmax:
            return sel, float(d[k])
        sel.append(int(cand[k]))


def faces_and_edges(V: np.ndarray, ang_tol: float = 1e-4, off_tol: float = 1e-5):
    """Merged faces [F, 4] and edges [E, 13] of conv(V), and the indices of the vertices that lie on an edge (a hull vertex whose
    triangles all merge into one face -- float32 noise of a flat CAD face -- is no corner).  Every row of V must be a hull vertex.
    Tolerances: the STL files carry float32 coordinates, so the triangles of one flat face differ by ~1e-6 rad; the face counts are
    the same for any ang_tol between 2e-5 and 1e-3."""
    from scipy.spatial import ConvexHull
    h = ConvexHull(V)
    assert len(h.vertices) == len(V), "a polytope vertex lies inside the hull of the others"
    nt = len(h.simplices)
    eq = h.equations
    parent = list(range(nt))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]; x = parent[x]
        return x

    edge_tris = {}
    for t, s in enumerate(h.simplices):
        for i, j in ((0, 1), (1, 2), (0, 2)):
            edge_tris.setdefault((min(s[i], s[j]), max(s[i], s[j])), []).append(t)
    for (i, j), ts in edge_tris.items():
        assert len(ts) == 2, "non-manifold hull"
        t0, t1 = ts
        if np.linalg.norm(np.cross(eq[t0, :3], eq[t1, :3])) < ang_tol and eq[t0, :3] @ eq[t1, :3] > 0 and abs(eq[t0, 3] - eq[t1, 3]) < off_tol:
            parent[find(t0)] = find(t1)
    roots = sorted({find(t) for t in range(nt)})
    fid = {r: k for k, r in enumerate(roots)}
    faces = np.zeros((len(roots), 4))
    for r in roots:
        members = [t for t in range(nt) if find(t) == r]
        n = eq[members, :3].mean(0); n /= np.linalg.norm(n)
        faces[fid[r], :3] = n
        faces[fid[r], 3] = float((V @ n).max())          # the support value: no vertex lies outside the face's plane
    edges = []; used = set()
    for (i, j), (t0, t1) in sorted(edge_tris.items()):
        f0, f1 = fid[find(t0)], fid[find(t1)]
        if f0 == f1:
            continue
        used.update((int(i), int(j)))
        e = V[j] - V[i]; ln = float(np.linalg.norm(e)); e /= ln
        n1, n2 = faces[f0, :3], faces[f1, :3]
        g = float(n1 @ n2)
        edges.append(np.concatenate([V[i], e, n1 - g * n2, n2 - g * n1, [ln]]))
    return faces, np.array(edges), sorted(used)


def collision_polytope(tris: np.ndarray, tol: float = TOL, full_below: int = FULL_BELOW) -> dict:
    """tris [ntri, 3, 3] (STL coordinates) -> the mesh's collision polytope and how far it is from the hull."""
    from scipy.spatial import ConvexHull
    pts = np.unique(tris.reshape(-1, 3), axis=0)
    hull = ConvexHull(pts)
    hv = pts[hull.vertices]
    sel, hd = select_vertices(hv, tol, full_below)
    V = hv[sel]
    faces, edges, used = faces_and_edges(V)
    V = V[used]
    return {"verts": V, "faces": faces, "edges": edges, "hausdorff": hd, "hull_nvert": int(len(hv)),
            "hull_volume": float(hull.volume), "volume": float(ConvexHull(V).volume)}


# ------------------------------------------------------------------------------------------------ the flat table
# One blob of doubles for the device and the oracle (struct-of-arrays per mesh, every array padded to a multiple of VPAD so that a wave
# reads one element per lane):  meta[NMESH][8] = {V, F, E, offset of the mesh's arrays, Vpad, Fpad, Epad, 0}, then per mesh
#   vx[Vpad] vy vz | fnx[Fpad] fny fnz fd | ep[3][Epad] ee[3][Epad] eu1[3][Epad] eu2[3][Epad] elen[Epad]
# padding: vertices repeat vertex 0; faces are (0, 0, 0 | +1e30) (never the axis of least penetration, never separating); edges have a
# zero direction (skipped by the |e x b| test).
META = 8
NEF = 13           # numbers per edge


def _pad(n):
    return ((n + VPAD - 1) // VPAD) * VPAD


def pack(polys: list) -> np.ndarray:
    assert len(polys) == NMESH
    parts = []; meta = np.zeros((NMESH, META)); off = NMESH * META
    for m, P in enumerate(polys):
        V, F, E = np.asarray(P["verts"]), np.asarray(P["faces"]), np.asarray(P["edges"])
        nv, nf, ne = len(V), len(F), len(E)
        vp, fp, ep = _pad(nv), _pad(nf), _pad(ne)
        meta[m] = [nv, nf, ne, off, vp, fp, ep, 0]
        vv = np.repeat(V[:1], vp, 0); vv[:nv] = V
        ff = np.zeros((fp, 4)); ff[:, 3] = 1e30; ff[:nf] = F
        ee = np.zeros((ep, NEF)); ee[:ne] = E
        blk = np.concatenate([vv.T.ravel(), ff.T.ravel(), ee.T.ravel()])
        parts.append(blk); off += len(blk)
    return np.concatenate([meta.ravel()] + parts)


def unpack(blob: np.ndarray) -> list:
    blob = np.asarray(blob, dtype=np.float64)
    meta = blob[:NMESH * META].reshape(NMESH, META)
    out = []
    for m in range(NMESH):
        nv, nf, ne, off, vp, fp, ep = (int(x) for x in meta[m, :7])
        v = blob[off:off + 3 * vp].reshape(3, vp).T[:nv]
        f = blob[off + 3 * vp:off + 3 * vp + 4 * fp].reshape(4, fp).T[:nf]
        e = blob[off + 3 * vp + 4 * fp:off + 3 * vp + 4 * fp + NEF * ep].reshape(NEF, ep).T[:ne]
        out.append({"verts": v.copy(), "faces": f.copy(), "edges": e.copy()})
    return out


def build_asset(meshdir: str, path: str = ASSET) -> dict:
    """Offline: STL files of the reference -> polytopes.npz (the blob in STL coordinates + per-mesh statistics)."""
    from .mjcf import load_stl
    polys, stats = [], {}
    for name in MESH_NAMES:
        P = collision_polytope(load_stl(os.path.join(meshdir, name + ".STL")))
        polys.append(P)
        stats[name] = dict(V=len(P["verts"]), F=len(P["faces"]), E=len(P["edges"]), hausdorff=P["hausdorff"], hull_nvert=P["hull_nvert"],
                           volume_ratio=P["volume"] / P["hull_volume"])
    blob = pack(polys)
    np.savez_compressed(path, blob=blob, names=np.array(MESH_NAMES),
                        stats=np.array([[stats[n][k] for k in ("V", "F", "E", "hausdorff", "hull_nvert", "volume_ratio")] for n in MESH_NAMES]))
    return stats


def load_asset(path: str = ASSET):
    z = np.load(path)
    return np.asarray(z["blob"], dtype=np.float64), np.asarray(z["stats"], dtype=np.float64)


def transform(polys: list, frames: list) -> list:
    """Polytopes in STL (= geom) coordinates -> in the frame of the engine body each rides on: frames[m] = (R, p) of the geom there."""
    out = []
    for P, (R, p) in zip(polys, frames):
        R = np.asarray(R, dtype=np.float64); p = np.asarray(p, dtype=np.float64)
        V = p + P["verts"] @ R.T
        F = P["faces"].copy(); F[:, :3] = P["faces"][:, :3] @ R.T; F[:, 3] = P["faces"][:, 3] + F[:, :3] @ p
        E = P["edges"].copy()
        E[:, 0:3] = p + P["edges"][:, 0:3] @ R.T
        for k in (3, 6, 9):
            E[:, k:k + 3] = P["edges"][:, k:k + 3] @ R.T
        out.append({"verts": V, "faces": F, "edges": E})
    return out
