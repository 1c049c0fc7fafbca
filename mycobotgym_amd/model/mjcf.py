"""MJCF -> compiled-model tables for the MyCobot-280 scene.

This is the build's stand-in for the model compilation that the reference delegates to
MuJoCo inside ``MujocoEnv.__init__`` (reference ``mycobotgym/envs/mycobot.py:60-75``:
``MjModel.from_xml_path``).  It is an *offline* tool: it reads the reference's MJCF + STL
files where they lie (``/root/reference/mycobotgym/envs/assets``) and writes a plain JSON
table (``mycobotgym_amd/assets/*.json``) that travels with this repo; nothing at run time
reads the reference tree.

Only the MJCF subset that the MyCobot scene uses is implemented: includes, default classes
with ``childclass``, bodies / inertials / hinge + free joints, box / plane / mesh geoms,
sites, fixed tendons, ``general`` actuators, connect / joint / weld equalities, contact
excludes, keyframes and mocap bodies.

Semantics follow MuJoCo 2.3.2 as recalled (all [RECALL]):
  * bodies are numbered depth-first in document order, world = 0;
  * bodies without an ``<inertial>`` get mass/inertia from their geoms at density 1000;
  * mesh volume/inertia use the *legacy* rule by default (area-weighted face-centroid apex,
    |volume| per pyramid); ``mesh_inertia="exact"`` switches to signed volumes.
"""
from __future__ import annotations

import json
import os
import struct
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional

import numpy as np

# MuJoCo enum values (mjtJoint, mjtGeom, mjtEq) kept so tables read like an mjModel dump
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_BOX, GEOM_MESH = 0, 6, 7
EQ_CONNECT, EQ_WELD, EQ_JOINT = 0, 1, 2

DEFAULT_SOLREF = [0.02, 1.0]
DEFAULT_SOLIMP = [0.9, 0.95, 0.001, 0.5, 2.0]
DEFAULT_FRICTION = [1.0, 0.005, 0.0001]
DEFAULT_DENSITY = 1000.0


# ----------------------------------------------------------------------------- quaternions
def quat_normalize(q):
    q = np.asarray(q, dtype=np.float64)
    return q / np.linalg.norm(q)


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def quat_conj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def mat_to_quat(R):
    """Rotation matrix -> unit quaternion (w,x,y,z), largest-component branch."""
    t = np.trace(R)
    if t > 0:
        w = 0.5 * np.sqrt(1 + t)
        q = [w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        x = 0.5 * np.sqrt(1 + R[0, 0] - R[1, 1] - R[2, 2])
        q = [(R[2, 1] - R[1, 2]) / (4 * x), x, (R[0, 1] + R[1, 0]) / (4 * x), (R[0, 2] + R[2, 0]) / (4 * x)]
    elif R[1, 1] > R[2, 2]:
        y = 0.5 * np.sqrt(1 - R[0, 0] + R[1, 1] - R[2, 2])
        q = [(R[0, 2] - R[2, 0]) / (4 * y), (R[0, 1] + R[1, 0]) / (4 * y), y, (R[1, 2] + R[2, 1]) / (4 * y)]
    else:
        z = 0.5 * np.sqrt(1 - R[0, 0] - R[1, 1] + R[2, 2])
        q = [(R[1, 0] - R[0, 1]) / (4 * z), (R[0, 2] + R[2, 0]) / (4 * z), (R[1, 2] + R[2, 1]) / (4 * z), z]
    return quat_normalize(q)


def euler_to_quat(e):
    """MJCF ``euler`` attribute, default sequence 'xyz' (intrinsic)."""
    q = np.array([1.0, 0, 0, 0])
    for ax, ang in enumerate(e):
        h = 0.5 * ang
        r = np.array([np.cos(h), 0, 0, 0])
        r[1 + ax] = np.sin(h)
        q = quat_mul(q, r)
    return q


# ----------------------------------------------------------------------------------- meshes
def load_stl(path: str) -> np.ndarray:
    """Binary STL -> float64 array [ntri, 3 vertices, 3 coords]."""
    with open(path, "rb") as f:
        data = f.read()
    ntri = struct.unpack_from("<I", data, 80)[0]
    if 84 + 50 * ntri != len(data):
        raise ValueError(f"{path}: not a binary STL (size mismatch)")
    rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]),
                        count=ntri, offset=84)
    return rec["v"].astype(np.float64)


def support_polytope(tris: np.ndarray) -> dict:
    """Collision proxy of a mesh geom (first stage): MuJoCo collides the CONVEX HULL of a mesh (qhull at compile time,
    500-1500 vertices for the arm links); this build collides the polytope spanned by the hull's support points in the 26 directions
    of a cube's faces, edges and corners -- an inner approximation of the hull (81-89 % of its volume for the arm links, within
    2 mm of its surface) that a per-sub-step kernel can afford.  Vertices in the STL's own coordinates."""
    import itertools
    from scipy.spatial import ConvexHull
    pts = np.unique(tris.reshape(-1, 3), axis=0)
    hull = ConvexHull(pts)
    hv = pts[hull.vertices]
    dirs = np.array([d for d in itertools.product((-1.0, 0.0, 1.0), repeat=3) if any(d)])
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    idx = []
    for k in np.argmax(hv @ dirs.T, axis=0):          # one support point per direction, first occurrence kept, order = direction order
        if int(k) not in idx: idx.append(int(k))
    sv = hv[idx]
    return {"hull_nvert": int(len(hv)), "hull_volume": float(hull.volume), "support": sv,
            "support_volume": float(ConvexHull(sv).volume) if len(sv) >= 4 else 0.0}


def mesh_inertial(tris: np.ndarray, rule: str = "legacy"):
    """Volume, centre of mass and inertia tensor (about the CoM, unit density) of a mesh.

    [RECALL MuJoCo 2.3.2 user_mesh.cc]  ``legacy``: the apex of every face pyramid is the
    area-weighted mean of the face centroids and each pyramid counts with |volume|
    (over-counts non-convex meshes); ``exact``: signed volumes about the origin.
    """
    v0, v1, v2 = tris[:, 0], tris[:, 1], tris[:, 2]
    nrm = np.cross(v1 - v0, v2 - v0)
    area2 = np.linalg.norm(nrm, axis=1)
    keep = area2 > 1e-30
    v0, v1, v2, nrm, area2 = v0[keep], v1[keep], v2[keep], nrm[keep], area2[keep]
    area = 0.5 * area2
    unit = nrm / area2[:, None]
    cen = (v0 + v1 + v2) / 3.0
    if rule == "legacy":
        apex = (area[:, None] * cen).sum(0) / area.sum()
    elif rule == "exact":
        apex = np.zeros(3)
    else:
        raise ValueError(rule)
    vol = np.einsum("ij,ij->i", cen - apex, unit) * area / 3.0
    if rule == "legacy":
        vol = np.abs(vol)
    volume = vol.sum()
    com = (vol[:, None] * (0.75 * cen + 0.25 * apex)).sum(0) / volume
    # second moments of each (com, v0, v1, v2) tetrahedron, apex at the CoM
    d0, d1, d2 = v0 - com, v1 - com, v2 - com
    volc = np.einsum("ij,ij->i", cen - com, unit) * area / 3.0
    if rule == "legacy":
        volc = np.abs(volc)
    P = np.zeros((3, 3))
    for a in range(3):
        for b in range(a, 3):
            s = (2 * (d0[:, a] * d0[:, b] + d1[:, a] * d1[:, b] + d2[:, a] * d2[:, b])
                 + d0[:, a] * d1[:, b] + d0[:, b] * d1[:, a]
                 + d0[:, a] * d2[:, b] + d0[:, b] * d2[:, a]
                 + d1[:, a] * d2[:, b] + d1[:, b] * d2[:, a])
            P[a, b] = P[b, a] = (volc * s).sum() / 20.0
    I = np.array([
        [P[1, 1] + P[2, 2], -P[0, 1], -P[0, 2]],
        [-P[0, 1], P[0, 0] + P[2, 2], -P[1, 2]],
        [-P[0, 2], -P[1, 2], P[0, 0] + P[1, 1]],
    ])
    return volume, com, I


def principal_axes(I: np.ndarray):
    """Symmetric inertia tensor -> (iquat, diag) with diag descending and det(R)=+1."""
    w, V = np.linalg.eigh(I)
    order = np.argsort(-w)
    w, V = w[order], V[:, order]
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    return mat_to_quat(V), w


# ------------------------------------------------------------------------- default classes
class _Defaults:
    def __init__(self, parent: Optional["_Defaults"] = None):
        self.tags: Dict[str, Dict[str, str]] = {}
        if parent is not None:
            self.tags = {k: dict(v) for k, v in parent.tags.items()}

    def update(self, tag: str, attrs: Dict[str, str]):
        self.tags.setdefault(tag, {}).update(attrs)

    def resolve(self, tag: str, attrs: Dict[str, str]) -> Dict[str, str]:
        out = dict(self.tags.get(tag, {}))
        out.update(attrs)
        return out


def _floats(s: str) -> List[float]:
    return [float(x) for x in s.split()]


def _bool(s: str) -> bool:
    return s.strip().lower() == "true"


# ---------------------------------------------------------------------------------- compile
class MjcfCompiler:
    def __init__(self, xml_path: str, mesh_inertia: str = "legacy", drop_bodies=(),
                 missing_mesh_ok=("base_link",)):
        self.xml_path = os.path.abspath(xml_path)
        self.dir = os.path.dirname(self.xml_path)
        self.mesh_rule = mesh_inertia
        self.drop_bodies = set(drop_bodies)
        self.missing_mesh_ok = set(missing_mesh_ok)
        self.root = self._load(self.xml_path)

    # -- XML loading with <include> expansion
    def _load(self, path: str) -> ET.Element:
        root = ET.parse(path).getroot()
        self._expand(root, os.path.dirname(path))
        return root

    def _expand(self, elem: ET.Element, base: str):
        i = 0
        while i < len(elem):
            child = elem[i]
            if child.tag == "include":
                inc = ET.parse(os.path.join(base, child.attrib["file"])).getroot()
                self._expand(inc, base)
                elem.remove(child)
                for k, sub in enumerate(list(inc)):
                    elem.insert(i + k, sub)
                i += len(inc)
            else:
                self._expand(child, base)
                i += 1

    def _sections(self, tag: str) -> List[ET.Element]:
        return [e for e in self.root if e.tag == tag]

    # -- defaults
    def _read_defaults(self):
        self.classes: Dict[str, _Defaults] = {"main": _Defaults()}

        def walk(elem: ET.Element, cls: _Defaults):
            for ch in elem:
                if ch.tag == "default":
                    sub = _Defaults(cls)
                    self.classes[ch.attrib["class"]] = sub
                    walk(ch, sub)
                else:
                    cls.update(ch.tag, dict(ch.attrib))

        for d in self._sections("default"):
            walk(d, self.classes["main"])

    def _attrs(self, elem: ET.Element, childclass: Optional[str]) -> Dict[str, str]:
        cname = elem.attrib.get("class", childclass or "main")
        a = {k: v for k, v in elem.attrib.items() if k != "class"}
        return self.classes[cname].resolve(elem.tag, a)

    # -- assets
    def _read_meshes(self):
        comp = self._sections("compiler")
        meshdir = comp[0].attrib.get("meshdir", "") if comp else ""
        self.meshes: Dict[str, dict] = {}
        for asset in self._sections("asset"):
            for m in asset.findall("mesh"):
                name = m.attrib["name"]
                path = os.path.join(self.dir, meshdir, m.attrib["file"])
                if not os.path.exists(path):
                    if name in self.missing_mesh_ok:
                        # base_link.STL is absent from the reference checkout (.MISSING_LARGE_BLOBS);
                        # it is only attached to a static body, so it carries no dynamics.
                        self.meshes[name] = {"missing": True, "volume": 0.0, "com": np.zeros(3),
                                             "inertia": np.zeros((3, 3)), "ntri": 0}
                        continue
                    raise FileNotFoundError(path)
                tris = load_stl(path)
                vol, com, I = mesh_inertial(tris, self.mesh_rule)
                vol_e, _, _ = mesh_inertial(tris, "exact")
                self.meshes[name] = {"missing": False, "volume": vol, "com": com, "inertia": I,
                                     "ntri": int(tris.shape[0]), "volume_exact": vol_e, **support_polytope(tris)}

    # -- main entry
    def compile(self) -> dict:
        self._read_defaults()
        self._read_meshes()
        opt = {"timestep": 0.002, "gravity": [0.0, 0.0, -9.81], "impratio": 1.0,
               "tolerance": 1e-8, "iterations": 100, "cone": "pyramidal", "integrator": "Euler",
               "solver": "Newton", "refsafe": True, "eulerdamp": True}
        for o in self._sections("option"):
            if "timestep" in o.attrib:
                opt["timestep"] = float(o.attrib["timestep"])
            if "gravity" in o.attrib:
                opt["gravity"] = _floats(o.attrib["gravity"])

        B = {k: [] for k in ("name", "parent", "pos", "quat", "ipos", "iquat", "mass", "inertia",
                             "mocap", "explicit_inertial")}
        J = {k: [] for k in ("name", "type", "body", "pos", "axis", "qposadr", "dofadr", "limited",
                             "range", "solref", "solimp", "armature", "damping", "ref")}
        G = {k: [] for k in ("name", "type", "body", "pos", "quat", "size", "condim", "friction",
                             "solref", "solimp", "contype", "conaffinity", "mesh", "density",
                             "mass_attr", "group")}
        S = {k: [] for k in ("name", "body", "pos", "quat")}
        B["name"].append("world"); B["parent"].append(0)
        B["pos"].append(np.zeros(3)); B["quat"].append(np.array([1.0, 0, 0, 0]))
        B["ipos"].append(np.zeros(3)); B["iquat"].append(np.array([1.0, 0, 0, 0]))
        B["mass"].append(0.0); B["inertia"].append(np.zeros(3)); B["mocap"].append(False)
        B["explicit_inertial"].append(True)
        nq = nv = 0

        def pose(attrs):
            pos = np.array(_floats(attrs["pos"])) if "pos" in attrs else np.zeros(3)
            if "quat" in attrs:
                q = quat_normalize(_floats(attrs["quat"]))
            elif "euler" in attrs:
                q = euler_to_quat(_floats(attrs["euler"]))
            else:
                q = np.array([1.0, 0, 0, 0])
            return pos, q

        def add_geom(e, bid, cc):
            a = self._attrs(e, cc)
            pos, q = pose(a)
            gtype = {"plane": GEOM_PLANE, "box": GEOM_BOX, "mesh": GEOM_MESH}[a.get("type", "sphere")]
            G["name"].append(a.get("name", "")); G["type"].append(gtype); G["body"].append(bid)
            G["pos"].append(pos); G["quat"].append(q)
            G["size"].append(np.array((_floats(a["size"]) + [0, 0, 0])[:3]) if "size" in a else np.zeros(3))
            G["condim"].append(int(a.get("condim", 3)))
            G["friction"].append(np.array(_floats(a["friction"])) if "friction" in a else np.array(DEFAULT_FRICTION))
            G["solref"].append(np.array(_floats(a["solref"])) if "solref" in a else np.array(DEFAULT_SOLREF))
            si = list(DEFAULT_SOLIMP)
            if "solimp" in a:
                v = _floats(a["solimp"]); si[:len(v)] = v
            G["solimp"].append(np.array(si))
            G["contype"].append(int(a.get("contype", 1))); G["conaffinity"].append(int(a.get("conaffinity", 1)))
            G["mesh"].append(a.get("mesh", "")); G["density"].append(float(a.get("density", DEFAULT_DENSITY)))
            G["mass_attr"].append(float(a["mass"]) if "mass" in a else None)
            G["group"].append(int(a.get("group", 0)))

        def add_site(e, bid, cc):
            a = self._attrs(e, cc)
            pos, q = pose(a)
            S["name"].append(a.get("name", "")); S["body"].append(bid); S["pos"].append(pos); S["quat"].append(q)

        def add_body(e: ET.Element, parent: int, cc: Optional[str]):
            nonlocal nq, nv
            if e.attrib.get("name") in self.drop_bodies:
                return
            cc = e.attrib.get("childclass", cc)
            bid = len(B["name"])
            pos, q = pose(e.attrib)
            B["name"].append(e.attrib.get("name", f"body{bid}")); B["parent"].append(parent)
            B["pos"].append(pos); B["quat"].append(q)
            B["mocap"].append(_bool(e.attrib.get("mocap", "false")))
            inert = e.find("inertial")
            if inert is not None:
                ipos, iq = pose(inert.attrib)
                B["ipos"].append(ipos); B["iquat"].append(iq)
                B["mass"].append(float(inert.attrib["mass"]))
                B["inertia"].append(np.array(_floats(inert.attrib["diaginertia"])))
                B["explicit_inertial"].append(True)
            else:
                B["ipos"].append(None); B["iquat"].append(None); B["mass"].append(None)
                B["inertia"].append(None); B["explicit_inertial"].append(False)
            for ch in e:
                if ch.tag in ("joint", "freejoint"):
                    a = self._attrs(ch, cc) if ch.tag == "joint" else dict(ch.attrib)
                    jt = {"hinge": JNT_HINGE, "free": JNT_FREE, "slide": JNT_SLIDE, "ball": JNT_BALL}[
                        "free" if ch.tag == "freejoint" else a.get("type", "hinge")]
                    if jt not in (JNT_HINGE, JNT_FREE):
                        raise NotImplementedError("only hinge and free joints occur in the MyCobot scene")
                    J["name"].append(a.get("name", "")); J["type"].append(jt); J["body"].append(bid)
                    J["pos"].append(np.array(_floats(a["pos"])) if "pos" in a else np.zeros(3))
                    ax = np.array(_floats(a["axis"])) if "axis" in a else np.array([0.0, 0, 1])
                    J["axis"].append(ax / np.linalg.norm(ax))
                    J["qposadr"].append(nq); J["dofadr"].append(nv)
                    has_range = "range" in a
                    lim = a.get("limited", "auto")
                    J["limited"].append(_bool(lim) if lim != "auto" else False)
                    J["range"].append(np.array(_floats(a["range"])) if has_range else np.zeros(2))
                    J["solref"].append(np.array(_floats(a["solreflimit"])) if "solreflimit" in a else np.array(DEFAULT_SOLREF))
                    si = list(DEFAULT_SOLIMP)
                    if "solimplimit" in a:
                        v = _floats(a["solimplimit"]); si[:len(v)] = v
                    J["solimp"].append(np.array(si))
                    J["armature"].append(float(a.get("armature", 0.0)))
                    J["damping"].append(float(a.get("damping", 0.0)))
                    J["ref"].append(float(a.get("ref", 0.0)))
                    nq += 7 if jt == JNT_FREE else 1
                    nv += 6 if jt == JNT_FREE else 1
                elif ch.tag == "geom":
                    add_geom(ch, bid, cc)
                elif ch.tag == "site":
                    add_site(ch, bid, cc)
            for ch in e:
                if ch.tag == "body":
                    add_body(ch, bid, cc)

        for wb in self._sections("worldbody"):
            for ch in wb:
                if ch.tag == "geom":
                    add_geom(ch, 0, None)
                elif ch.tag == "site":
                    add_site(ch, 0, None)
            for ch in wb:
                if ch.tag == "body":
                    add_body(ch, 0, None)

        nbody = len(B["name"])
        # geom volumes / inertias (geom frame for primitives; body frame tensor for meshes)
        for b in range(1, nbody):
            if B["explicit_inertial"][b]:
                continue
            mass = 0.0; mc = np.zeros(3); parts = []
            for g in range(len(G["name"])):
                if G["body"][g] != b:
                    continue
                R = quat_to_mat(G["quat"][g])
                if G["type"][g] == GEOM_BOX:
                    s = G["size"][g]
                    vol = 8 * s[0] * s[1] * s[2]
                    m = G["mass_attr"][g] if G["mass_attr"][g] is not None else G["density"][g] * vol
                    Ig = m / 3.0 * np.diag([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2])
                    c = G["pos"][g]; Ib = R @ Ig @ R.T
                elif G["type"][g] == GEOM_MESH:
                    me = self.meshes[G["mesh"][g]]
                    m = G["mass_attr"][g] if G["mass_attr"][g] is not None else G["density"][g] * me["volume"]
                    scale = (m / me["volume"]) if me["volume"] > 0 else 0.0
                    c = G["pos"][g] + R @ me["com"]; Ib = scale * (R @ me["inertia"] @ R.T)
                else:
                    continue
                if m <= 0:
                    continue
                mass += m; mc += m * c; parts.append((m, c, Ib))
            if mass > 0:
                com = mc / mass
                I = np.zeros((3, 3))
                for m, c, Ib in parts:
                    d = c - com
                    I += Ib + m * (d @ d * np.eye(3) - np.outer(d, d))
                iq, diag = principal_axes(I)
                B["ipos"][b] = com; B["iquat"][b] = iq; B["mass"][b] = mass; B["inertia"][b] = diag
            else:
                B["ipos"][b] = np.zeros(3); B["iquat"][b] = np.array([1.0, 0, 0, 0])
                B["mass"][b] = 0.0; B["inertia"][b] = np.zeros(3)

        # dof tables
        dof_body, dof_jnt, dof_parent, dof_arm, dof_damp = [], [], [], [], []
        body_dofadr = [-1] * nbody; body_dofnum = [0] * nbody
        for j in range(len(J["name"])):
            n = 6 if J["type"][j] == JNT_FREE else 1
            b = J["body"][j]
            if body_dofadr[b] < 0:
                body_dofadr[b] = J["dofadr"][j]
            body_dofnum[b] += n
            for k in range(n):
                dof_body.append(b); dof_jnt.append(j)
                dof_arm.append(J["armature"][j]); dof_damp.append(J["damping"][j])
        # dof_parentid: previous dof in the same body, else last dof of the nearest ancestor with dofs
        for d in range(nv):
            b = dof_body[d]
            if d > body_dofadr[b]:
                dof_parent.append(d - 1)
                continue
            p = B["parent"][b]
            while p != 0 and body_dofnum[p] == 0:
                p = B["parent"][p]
            dof_parent.append(-1 if p == 0 else body_dofadr[p] + body_dofnum[p] - 1)

        qpos0 = np.zeros(nq)
        for j in range(len(J["name"])):
            a = J["qposadr"][j]
            if J["type"][j] == JNT_FREE:
                b = J["body"][j]
                qpos0[a:a + 3] = B["pos"][b]; qpos0[a + 3:a + 7] = B["quat"][b]
            else:
                qpos0[a] = J["ref"][j]

        # body_rootid / weldid
        rootid = [0] * nbody; weldid = [0] * nbody
        for b in range(1, nbody):
            p = B["parent"][b]
            rootid[b] = b if p == 0 else rootid[p]
            weldid[b] = b if body_dofnum[b] > 0 else weldid[p]

        bname = {n: i for i, n in enumerate(B["name"])}
        jname = {n: i for i, n in enumerate(J["name"])}

        # tendons (fixed)
        tendons = []
        for sec in self._sections("tendon"):
            for t in sec.findall("fixed"):
                tendons.append({"name": t.attrib.get("name", ""),
                                "joints": [jname[j.attrib["joint"]] for j in t.findall("joint")],
                                "coefs": [float(j.attrib["coef"]) for j in t.findall("joint")]})
        tname = {t["name"]: i for i, t in enumerate(tendons)}

        # actuators
        acts = []
        for sec in self._sections("actuator"):
            for e in sec:
                if e.tag != "general":
                    raise NotImplementedError(e.tag)
                a = self._attrs(e, None)
                gain = (_floats(a.get("gainprm", "1")) + [0, 0, 0])[:3]
                bias = (_floats(a.get("biasprm", "0")) + [0, 0, 0])[:3]
                if a.get("biastype", "none") == "none":
                    bias = [0.0, 0.0, 0.0]
                if a.get("dyntype", "none") != "none" or a.get("gaintype", "fixed") != "fixed":
                    raise NotImplementedError("actuator dynamics / non-fixed gain")
                acts.append({
                    "name": a.get("name", ""),
                    "trntype": "joint" if "joint" in a else "tendon",
                    "trnid": jname[a["joint"]] if "joint" in a else tname[a["tendon"]],
                    "gear": float((_floats(a.get("gear", "1")))[0]),
                    "gainprm": gain, "biasprm": bias,
                    "ctrllimited": _bool(a.get("ctrllimited", "false")),
                    "ctrlrange": _floats(a.get("ctrlrange", "0 0")),
                    "forcelimited": _bool(a.get("forcelimited", "false")),
                    "forcerange": _floats(a.get("forcerange", "0 0")),
                })

        model = {
            "source": os.path.basename(self.xml_path),
            "mesh_inertia": self.mesh_rule,
            "dropped_bodies": sorted(self.drop_bodies),
            "opt": opt,
            "nbody": nbody, "nq": nq, "nv": nv, "njnt": len(J["name"]), "ngeom": len(G["name"]),
            "nsite": len(S["name"]), "nu": len(acts), "ntendon": len(tendons),
            "body_name": B["name"], "body_parent": B["parent"], "body_rootid": rootid, "body_weldid": weldid,
            "body_mocap": B["mocap"],
            "body_pos": B["pos"], "body_quat": B["quat"], "body_ipos": B["ipos"], "body_iquat": B["iquat"],
            "body_mass": B["mass"], "body_inertia": B["inertia"],
            "body_dofadr": body_dofadr, "body_dofnum": body_dofnum,
            "jnt_name": J["name"], "jnt_type": J["type"], "jnt_body": J["body"], "jnt_pos": J["pos"],
            "jnt_axis": J["axis"], "jnt_qposadr": J["qposadr"], "jnt_dofadr": J["dofadr"],
            "jnt_limited": J["limited"], "jnt_range": J["range"], "jnt_solref": J["solref"],
            "jnt_solimp": J["solimp"],
            "dof_body": dof_body, "dof_jnt": dof_jnt, "dof_parent": dof_parent,
            "dof_armature": dof_arm, "dof_damping": dof_damp, "qpos0": qpos0,
            "geom_name": G["name"], "geom_type": G["type"], "geom_body": G["body"], "geom_pos": G["pos"],
            "geom_quat": G["quat"], "geom_size": G["size"], "geom_condim": G["condim"],
            "geom_friction": G["friction"], "geom_solref": G["solref"], "geom_solimp": G["solimp"],
            "geom_contype": G["contype"], "geom_conaffinity": G["conaffinity"], "geom_mesh": G["mesh"],
            "site_name": S["name"], "site_body": S["body"], "site_pos": S["pos"], "site_quat": S["quat"],
            "tendons": tendons, "actuators": acts,
            "meshes": {k: {"missing": v["missing"], "ntri": v["ntri"], "volume": v["volume"],
                           "volume_exact": v.get("volume_exact", 0.0), "com": v["com"],
                           "hull_nvert": v.get("hull_nvert", 0), "hull_volume": v.get("hull_volume", 0.0),
                           "support": v.get("support", np.zeros((0, 3))), "support_volume": v.get("support_volume", 0.0)}
                       for k, v in self.meshes.items()},
        }

        # kinematics at qpos0 for equality anchors / weld relposes
        from .refdyn import kinematics  # local import: refdyn only needs the tables built so far
        kin = kinematics(_np_model(model), qpos0)
        eqs = []
        for sec in self._sections("equality"):
            for e in sec:
                si = list(DEFAULT_SOLIMP)
                if "solimp" in e.attrib:
                    v = _floats(e.attrib["solimp"]); si[:len(v)] = v
                sr = _floats(e.attrib["solref"]) if "solref" in e.attrib else list(DEFAULT_SOLREF)
                if e.tag == "connect":
                    if e.attrib["body1"] not in bname or e.attrib["body2"] not in bname:
                        continue
                    b1, b2 = bname[e.attrib["body1"]], bname[e.attrib["body2"]]
                    a1 = np.array(_floats(e.attrib["anchor"]))
                    w = kin["xpos"][b1] + kin["xmat"][b1] @ a1
                    a2 = kin["xmat"][b2].T @ (w - kin["xpos"][b2])
                    eqs.append({"type": EQ_CONNECT, "obj1": b1, "obj2": b2,
                                "data": list(a1) + list(a2) + [0.0] * 5, "solref": sr, "solimp": si})
                elif e.tag == "joint":
                    j1 = jname[e.attrib["joint1"]]; j2 = jname[e.attrib["joint2"]]
                    pc = (_floats(e.attrib.get("polycoef", "0 1 0 0 0")) + [0] * 5)[:5]
                    eqs.append({"type": EQ_JOINT, "obj1": j1, "obj2": j2, "data": pc + [0.0] * 6,
                                "solref": sr, "solimp": si})
                elif e.tag == "weld":
                    if e.attrib["body1"] not in bname or e.attrib["body2"] not in bname:
                        continue
                    b1, b2 = bname[e.attrib["body1"]], bname[e.attrib["body2"]]
                    # [RECALL 2.3.2] data = anchor(3, in body2) | relpos(3) | relquat(4) | torquescale
                    anchor = np.array(_floats(e.attrib.get("anchor", "0 0 0")))
                    q1, q2 = kin["xquat"][b1], kin["xquat"][b2]
                    relpos = kin["xmat"][b1].T @ (kin["xpos"][b2] - kin["xpos"][b1])
                    relquat = quat_mul(quat_conj(q1), q2)
                    eqs.append({"type": EQ_WELD, "obj1": b1, "obj2": b2,
                                "data": list(anchor) + list(relpos) + list(relquat) + [float(e.attrib.get("torquescale", 1.0))],
                                "solref": sr, "solimp": si})
        model["eq"] = eqs
        model["neq"] = len(eqs)

        excl = []
        for sec in self._sections("contact"):
            for e in sec.findall("exclude"):
                if e.attrib["body1"] in bname and e.attrib["body2"] in bname:
                    excl.append([bname[e.attrib["body1"]], bname[e.attrib["body2"]]])
        model["excludes"] = excl

        keys = []
        for sec in self._sections("keyframe"):
            for k in sec.findall("key"):
                kq = _floats(k.attrib["qpos"]) if "qpos" in k.attrib else list(qpos0)
                if len(kq) != nq and self.drop_bodies:
                    kq = kq[:nq]   # dropped trailing free body (the cube is the last joint)
                kv = _floats(k.attrib["qvel"])[:nv] if "qvel" in k.attrib else [0.0] * nv
                keys.append({"name": k.attrib.get("name", ""), "qpos": kq, "qvel": kv,
                             "ctrl": _floats(k.attrib["ctrl"]) if "ctrl" in k.attrib else [0.0] * len(acts),
                             "mpos": _floats(k.attrib["mpos"]) if "mpos" in k.attrib else [],
                             "mquat": _floats(k.attrib["mquat"]) if "mquat" in k.attrib else []})
        model["keys"] = keys
        return _to_jsonable(model)


def _to_jsonable(o):
    if isinstance(o, dict):
        return {k: _to_jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_to_jsonable(v) for v in o]
    if isinstance(o, np.ndarray):
        return o.tolist()
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.bool_,)):
        return bool(o)
    return o


_ARRAY_KEYS = ("body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia", "jnt_pos",
               "jnt_axis", "jnt_range", "jnt_solref", "jnt_solimp", "dof_armature", "dof_damping", "qpos0",
               "geom_pos", "geom_quat", "geom_size", "geom_friction", "geom_solref", "geom_solimp",
               "site_pos", "site_quat")


def _np_model(model: dict) -> dict:
    """Copy of a model table with the numeric fields as float64 arrays."""
    m = dict(model)
    for k in _ARRAY_KEYS:
        if k in m:
            m[k] = np.asarray(m[k], dtype=np.float64)
    return m


def load_model(path: str) -> dict:
    with open(path) as f:
        return _np_model(json.load(f))


def save_model(model: dict, path: str):
    with open(path, "w") as f:
        json.dump(_to_jsonable(model), f, indent=1)
        f.write("\n")
