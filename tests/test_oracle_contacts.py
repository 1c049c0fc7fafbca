"""Oracle P4 (collision) and contact rows: box-box / plane-box geometry, the resting cube, friction, a scripted grasp.

Known answer NOT reproduced: the reference keyframes (mycobot280.xml:6, mycobot280_mocap.xml:7) store the resting cube
at z = 0.209981, i.e. 1.9e-5 below 0.21; the restated contact model settles 9.59e-6 below -- its own analytic value
(checked here), a factor 1.98 ~ (1 + mu^2) stiffer.  One of the [RECALL] regularisation rules for pyramidal contacts is
therefore off by that factor; which one cannot be decided without MuJoCo.
"""
import ctypes as C
import json

import numpy as np
import pytest

from tests.common import load_json, make_oracle


@pytest.fixture(scope="module")
def po(built):
    from oracle import pyoracle
    return pyoracle


def _scene(po, cube_pos, cube_quat=(1, 0, 0, 0), qvel=None):
    tab = load_json("mycobot280")
    om = po.OracleModel(tab, enable_contact=True); d = po.OracleData(om)
    q = np.asarray(tab["qpos0"], dtype=float).copy(); q[12:15] = cube_pos; q[15:19] = cube_quat
    d.set_state(qpos=q, qvel=np.zeros(18) if qvel is None else qvel)
    return tab, om, d


def _contacts(d):
    n = int(d.get("ncon", (1,), np.int32)[0])
    nefc = int(d.get("nefc", (1,), np.int32)[0])
    return n, nefc


def test_cube_on_table_four_face_contacts(po):
    tab, om, d = _scene(po, [-0.05, 0.0, 0.2099])
    d.forward()
    n, nefc = _contacts(d)
    assert n == 4 and nefc == 7 + 4 * 6                      # 7 equality rows + 4 condim-4 pyramids
    pos = d.get("efc_pos", (416,))[7:31]
    assert np.allclose(pos, -1e-4, atol=1e-12)               # dist = -penetration on every pyramid row
    # separated: no contact; cube over the table edge: fewer vertices inside
    tab, om, d = _scene(po, [-0.05, 0.0, 0.2101]); d.forward(); assert _contacts(d)[0] == 0
    # half over the x = 0.2 edge: the incident face is clipped to the table top (2 vertices + 2 clip points)
    tab, om, d = _scene(po, [0.2, 0.0, 0.2099]); d.forward(); assert _contacts(d)[0] == 4
    tab, om, d = _scene(po, [0.2105, 0.0, 0.2099]); d.forward(); assert _contacts(d)[0] == 0
    # on the ground plane next to the table
    tab, om, d = _scene(po, [0.5, 0.0, 0.0099]); d.forward(); assert _contacts(d)[0] == 4


def test_edge_contact_and_tilted_cube(po):
    # cube rotated 45 deg about x and lowered: one bottom edge touches the table -> 2 vertices
    c, s = np.cos(np.pi / 8), np.sin(np.pi / 8)
    tab, om, d = _scene(po, [-0.05, 0.0, 0.2 + 0.01 * np.sqrt(2) - 1e-4], (c, s, 0, 0))
    d.forward()
    assert _contacts(d)[0] == 2


def test_nearly_parallel_edges_never_beat_the_faces(po):
    """A cube rocking on the table by 1e-9 .. 1e-3 rad, with rounding-level noise in its quaternion: always the four face contacts
    (normal +z), never an edge-edge axis made of the rounding noise of two parallel edges (the fault
    tests/golden/cube_parallel_edge_state.npz records; EDGE_MIN_SIN in mco_collision.c)."""
    rng = np.random.default_rng(0)
    tab, om, d = _scene(po, [-0.05, 0.0, 0.21])
    q = np.asarray(tab["qpos0"], dtype=float).copy()
    for trial in range(400):
        ang = 10.0 ** rng.uniform(-9, -3.3)
        axis = rng.normal(size=3); axis[2] = 0; axis /= np.linalg.norm(axis)
        if trial % 3 == 0: axis = np.array([0.0, 1.0, 0.0])
        quat = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis]) * (1 + rng.uniform(-1, 1) * 1e-8)
        quat = quat + rng.uniform(-1, 1, 4) * 2e-16
        q[12:15] = [rng.uniform(-0.15, 0.15), rng.uniform(-0.2, 0.2), 0.21 - 2e-5]
        q[15:19] = quat
        d.set_state(qpos=q, qvel=np.zeros(18)); d.forward()
        n, nefc = _contacts(d)
        assert n == 4 and nefc == 7 + 24, (trial, ang, n)
        J = d.get("efc_J", (416, 24))[7:31, 12:15]
        assert np.abs(J[0::6, 2] - 1.0).max() < 1e-6, (trial, ang)       # first pyramid row: n + mu t1, z-component = n_z = 1
        assert d.get("efc_pos", (416,))[7:31].max() < -1e-6


def _rest_penetration(po, rules=None):
    tab, om, d = _scene(po, [-0.05, 0.0, 0.21])
    if rules:
        r = np.zeros(8, dtype=np.int32)
        for k, v in rules.items(): r[k] = v
        om._set_i("rule", r)
    d.step(1500)
    assert abs(d.qvel[14]) < 1e-10 and np.abs(d.qvel[12:18]).max() < 1e-9
    return 0.21 - d.qpos[14]


def test_rest_height_is_the_closed_form_of_the_restated_rules(po):
    """The settled cube sits where the restated rules say: m g = 4 contacts x 6 pyramid edges x D K imp pen with
    D = 1 / (2 mu^2 R), R = (1 - imp) / imp * tran (1 + mu^2) (all [RECALL])."""
    pen = _rest_penetration(po)
    K, imp_d0, imp_d1, width = 9551.195, 0.9495, 0.9745, 0.001
    def residual(p):
        x = p / width; imp = imp_d0 + 2 * x * x * (imp_d1 - imp_d0)
        D = 1 / (2 * (1 - imp) / imp * 125 * 2)
        return 24 * D * K * imp * p - 0.008 * 9.81
    lo, hi = 1e-7, 1e-4
    for _ in range(80):
        mid = 0.5 * (lo + hi); lo, hi = (mid, hi) if residual(mid) < 0 else (lo, mid)
    assert abs(pen - lo) < 2e-9


REF_PEN = (1.85e-5, 1.95e-5)      # both keyframes store z = 0.209981 (mycobot280.xml:6, mycobot280_mocap.xml:7)


@pytest.mark.xfail(strict=True, reason="known answer NOT reproduced: the restated [RECALL] contact rules give 9.59e-6, the reference's "
                   "keyframes 1.9e-5 (2x softer); oracle/RULE_STUDY.md lists the single-rule changes that would close it, none of "
                   "which the recalled MuJoCo source supports")
def test_rest_height_reference_keyframe(po):
    pen = _rest_penetration(po)
    assert REF_PEN[0] <= pen <= REF_PEN[1]


def test_rest_height_candidate_rule_reproduces_the_keyframe(po):
    """oracle/RULE_STUDY.md, K1: of the enumerated constraint-rule variants exactly one lands in the keyframes' window."""
    hits = [(r2, r3) for r2 in (0, 1) for r3 in (0, 1, 2) if REF_PEN[0] <= _rest_penetration(po, {2: r2, 3: r3}) <= REF_PEN[1]]
    assert hits == [(0, 2)]


def test_rest_height_under_contact_rule_keyframe(po):
    """contact_rule="keyframe" (include/mcg.h: contact_rpy = 4; oracle rule[3] = 2; MyCobotVecEnv(contact_rule="keyframe")): the
    selectable variant under which the reference's one contact datum -- the cube resting at z = 0.209981 in both keyframes
    (mycobot280.xml:6, mycobot280_mocap.xml:7) -- IS reproduced.  The default stays the rule as recalled (the xfail above)."""
    from mycobotgym_amd.model.mjcf import _np_model
    from mycobotgym_amd.model.specialize import specialize
    tab = load_json("mycobot280")
    assert specialize(_np_model(tab), contact_rule="keyframe")["contact_rpy"] == 4.0 and specialize(_np_model(tab))["contact_rpy"] == 2.0
    pen = _rest_penetration(po, {3: 2})
    assert REF_PEN[0] <= pen <= REF_PEN[1], pen


def test_friction_holds_then_slides(po):
    """Tangential push below mu * N sticks (soft: creeps), above it slides: Coulomb behaviour of the pyramid."""
    tab = load_json("mycobot280")
    om = po.OracleModel(tab, enable_contact=True); d = po.OracleData(om)
    d.step(600)                                        # settle
    v = d.qvel.copy(); v[12] = 0.05; d.set_state(qvel=v)
    d.step(200)
    assert abs(d.qvel[12]) < 5e-3                      # friction (mu = 1) stops a 5 cm/s slide within 0.4 s
    assert abs(d.qpos[14] - 0.20999) < 2e-5


def test_scripted_grasp_pinches_the_cube(po):
    tab = load_json("mycobot280")
    om = po.OracleModel(tab, enable_contact=True); d = po.OracleData(om)
    key = tab["keys"][0]
    d.set_state(qpos=key["qpos"], ctrl=key["ctrl"]); d.forward()
    gn = tab["geom_name"]; gr, gl, gc = gn.index("right_finger_layer"), gn.index("left_finger_layer"), gn.index("object0")
    gx = d.get("geom_xpos", (48, 3))
    q = np.asarray(key["qpos"], dtype=float).copy(); q[12:15] = 0.5 * (gx[gr] + gx[gl]); q[15:19] = [1, 0, 0, 0]
    ctrl = np.asarray(key["ctrl"], dtype=float).copy(); ctrl[6] = 1.0
    d.set_state(qpos=q, qvel=np.zeros(18), ctrl=ctrl)
    z0 = q[14]
    d.step(20 * 12)
    assert d.qpos[6] > 0.6                              # gripper closed on the cube
    ncon = int(d.get("ncon", (1,), np.int32)[0])
    g1 = d.get("efc_id", (224,), np.int32)
    assert ncon >= 2
    # the cube is held: it has dropped far less than free fall would (0.5 g t^2 = 1.1 m in 0.48 s)
    assert z0 - d.qpos[14] < 0.05
    assert int(d.get("warning_badstate", (1,), np.int32)[0]) == 0


def test_pickandplace_env_obs_layout_and_shaping(built):
    """25-number observation (Appendix A.6) and the staged reward with the stale target site (Appendix D-8)."""
    ora = make_oracle(4, has_object=True, controller_type="joint", reward_type="reward_shaping", seed=0)
    obs, ag, dg = ora.reset(seed=0)
    assert obs.shape == (4, 25)
    grip, objp, rel = obs[:, 0:3], obs[:, 3:6], obs[:, 6:9]
    assert np.allclose(rel, objp - grip) and np.allclose(ag, objp)
    assert np.allclose(objp[:, 2], 0.21) and np.all(np.hypot(objp[:, 0] - grip[:, 0], objp[:, 1] - grip[:, 1]) >= 0.1)
    assert np.all(np.hypot(dg[:, 0] - objp[:, 0], dg[:, 1] - objp[:, 1]) >= 0.1)          # mycobot.py:232
    o = ora.step(np.zeros((4, 7), np.float32))
    d = np.linalg.norm(o["obs"][:, 0:3] - o["obs"][:, 3:6], axis=1)
    assert np.allclose(o["reward"], 100 * 0.2 * (1 - np.tanh(d)), atol=1e-12)             # reach stage only


def test_domain_randomisation_changes_only_the_cube(built):
    ora = make_oracle(16, has_object=True, controller_type="joint", seed=1,
                      domain_randomization={"mass": (0.5, 2.0), "friction": (0.5, 1.5)})
    ora.reset(seed=1)
    o1 = ora.step(np.zeros((16, 7), np.float32))
    base = make_oracle(16, has_object=True, controller_type="joint", seed=1); base.reset(seed=1)
    o2 = base.step(np.zeros((16, 7), np.float32))
    # same goals and cube placement (DR draws use their own stream); arm identical, cube rest height differs with mass
    assert np.array_equal(o1["desired"], o2["desired"]) and np.allclose(o1["obs"][:, 0:3], o2["obs"][:, 0:3], atol=1e-12)
    assert np.abs(o1["obs"][:, 5] - o2["obs"][:, 5]).max() > 1e-7


def test_finger_pads_never_reach_each_other(po):
    """The last primitive pair the kernels leave out: right pad <-> left pad (mycobot280_main.xml:195-199,222-225).  With EVERY
    primitive pair enabled in the oracle, closing the empty gripper fully stops at the gear joints' upper limit (0.7 rad) with the
    two 2 mm thick pads still 2.6 mm apart: the pair cannot collide inside the joint range, so leaving it out changes nothing."""
    tab = load_json("mycobot280")
    om = po.OracleModel(tab, enable_contact=True, scope_geom=-1); d = po.OracleData(om)
    q = np.asarray(tab["qpos0"], dtype=float).copy(); q[12] = 0.15                      # the cube out of the gripper's way
    d.set_state(qpos=q, qvel=np.zeros(18), ctrl=[0, 0, 0, 0, 0, 0, 1.0])
    gn = tab["geom_name"]; gr, gl = gn.index("right_finger_layer"), gn.index("left_finger_layer")
    gaps = []
    for _ in range(30):
        d.step(20)
        gx = d.get("geom_xpos", (48, 3))
        gaps.append(np.linalg.norm(gx[gr] - gx[gl]) - 2 * tab["geom_size"][gr][2])
        assert int(d.get("ncon", (1,), np.int32)[0]) == 4                               # the cube on the table, nothing else
    assert d.qpos[6] > 0.699 and 0.0024 < gaps[-1] < 0.0028 and min(gaps) > 0.0024


def _contact_table(d):
    """The oracle's contact list: (geom1, geom2, dim, dist, pos, normal, efc_address) per contact (mco_contact, exported raw)."""
    n = int(d.get("ncon", (1,), np.int32)[0])
    raw = d.get("contact", (64, 28))
    out = []
    for c in range(n):
        ints = raw[c, 26:28].copy().view(np.int32)           # dim, geom1, geom2, efc_address
        out.append((int(ints[1]), int(ints[2]), int(ints[0]), float(raw[c, 0]), raw[c, 1:4].copy(), raw[c, 4:7].copy(), int(ints[3])))
    return out


def test_finger_pad_on_the_table_is_a_contact(po):
    """Scoped set (what the kernels implement): cube pairs + finger pad <-> table / ground plane (+ the mesh polytopes, behind them)."""
    tab = load_json("mycobot280")
    om = po.OracleModel(tab, enable_contact=True, scope_geom=tab["geom_name"].index("object0")); d = po.OracleData(om)
    q = np.asarray(tab["qpos0"], dtype=float).copy()
    q[:7] = [0.417, -2.299, 1.057, 0.345, 1.63, 0.161, 0.569]; q[8] = q[6]            # a pose that presses a pad on the table top
    d.set_state(qpos=q, qvel=np.zeros(18)); d.forward()
    gn = tab["geom_name"]; pads = {gn.index("right_finger_layer"), gn.index("left_finger_layer")}
    t = _contact_table(d)
    padc = [c for c in t if c[1] in pads]
    assert padc and all(c[0] == 1 and c[2] == 4 for c in padc)                          # the table is geom1; condim 4 (the pad's)
    # pair order: the primitive pairs first -- (table, pads) before (table, cube) -- then the meshes
    kinds = ["pad" if c[1] in pads else ("cube" if c[1] == gn.index("object0") and c[0] == 1 else "mesh") for c in t]
    assert kinds == sorted(kinds, key=["pad", "cube", "mesh"].index)
    J = d.get("efc_J", (416, 24))
    for c in padc:
        rows = J[c[6]:c[6] + 6, :18]
        assert np.abs(rows[:, 12:18]).max() == 0.0 and np.abs(rows[:, :10]).max() > 0  # rows of the pad contacts: robot dofs only


def test_arm_mesh_on_the_table_is_a_contact(po):
    """the arm's mesh geoms collide with the table / ground through their collision polytopes, one contact per geom pair;
    every mesh is attached twice in the reference (mycobot280_main.xml:105-175: a visual copy with density 0 and a default one, both
    colliding), so contacts come in identical pairs (one ENTRY of the list); default geoms are condim 3: four pyramid rows, in the arm's
    dofs up to the link."""
    tab = load_json("mycobot280")
    om = po.OracleModel(tab, enable_contact=True, scope_geom=tab["geom_name"].index("object0")); d = po.OracleData(om)
    q = np.asarray(tab["qpos0"], dtype=float).copy()
    d.set_state(qpos=q, qvel=np.zeros(18)); d.forward()
    assert int(d.get("ncon", (1,), np.int32)[0]) == 4                    # upright arm: only the cube on the table
    # fold the arm forward until exactly one link presses on the table top
    rng = np.random.default_rng(3)
    found = None
    for _ in range(20000):
        q[:6] = rng.uniform(-2.5, 2.5, 6)
        d.set_state(qpos=q, qvel=np.zeros(18)); d.forward()
        n = int(d.get("ncon", (1,), np.int32)[0])
        nefc = int(d.get("nefc", (1,), np.int32)[0])
        if n == 6 and nefc == 7 + 2 * 4 + 4 * 6:                         # exactly one mesh pair (2 x condim 3) + the cube's four (condim 4)
            found = q.copy(); break
    assert found is not None
    assert int(d.get("nentry", (1,), np.int32)[0]) == 5                  # the twins are one entry
    t = _contact_table(d)
    assert [c[2] for c in t] == [4, 4, 4, 4, 3, 3] and t[5][0] == t[4][0] and t[5][1] == t[4][1] + 1
    J = d.get("efc_J", (416, 24))[t[4][6]:t[4][6] + 8, :18]
    pos = d.get("efc_pos", (416,))[t[4][6]:t[4][6] + 8]
    assert np.allclose(J[:4], J[4:8]) and np.allclose(pos[:4], pos[4:8]) and pos[0] < 0        # the two copies: identical contacts
    assert np.abs(J[:, 6:]).max() == 0.0 and np.abs(J[:, :6]).max() > 0                         # arm dofs only


def test_collision_polytopes_are_within_a_millimetre_of_the_hulls():
    """mycobotgym_amd/assets/polytopes.npz: every mesh's polytope within 1 mm (Hausdorff) of its convex hull, the gripper's small parts
    exact; Euler's formula holds for the merged faces and edges; every vertex inside every face plane; the edges' cone vectors are
    perpendicular to the edges."""
    from mycobotgym_amd.model import polytope as pt
    blob, stats = pt.load_asset()
    polys = pt.unpack(blob)
    assert len(polys) == 14
    for name, P, st in zip(pt.MESH_NAMES, polys, stats):
        V, F, E = P["verts"], P["faces"], P["edges"]
        assert (len(V), len(F), len(E)) == tuple(int(x) for x in st[:3])
        assert len(V) - len(E) + len(F) == 2, name
        assert st[3] <= 1.0e-3 and (st[3] == 0.0 or st[4] > 64), (name, st[3])
        assert (V @ F[:, :3].T - F[:, 3]).max() < 1e-12
        assert np.abs(np.linalg.norm(F[:, :3], axis=1) - 1).max() < 1e-12 and np.abs(np.linalg.norm(E[:, 3:6], axis=1) - 1).max() < 1e-12
        assert np.abs((E[:, 3:6] * E[:, 6:9]).sum(1)).max() < 1e-5 and np.abs((E[:, 3:6] * E[:, 9:12]).sum(1)).max() < 1e-5      # (merged faces: float32 STL noise)
        assert st[5] > 0.95 or st[3] == 0.0                                 # volume against the hull's
    assert np.array_equal(pt.pack(polys), blob)


def test_gripper_meshes_against_the_cube(po):
    """The mesh geoms against the cube (oracle side).  None of the scripted-grasp START states holds such a contact; with the gripper
    pushed sideways they appear: condim 4, the mesh as geom1, one contact per geom, the twin geom's copy right after (the reference
    attaches every mesh twice), behind the primitive pairs' contacts, which are what they are without the meshes."""
    from mycobotgym_amd.scenarios import grasp_state
    tab = load_json("mycobot280")
    scope = tab["geom_name"].index("object0")
    d1 = po.OracleData(po.OracleModel(tab, enable_contact=True, scope_geom=scope))
    d0 = po.OracleData(po.OracleModel(tab, enable_contact=True, scope_geom=scope, mesh_collision=False))
    meshes = {g for g in range(tab["ngeom"]) if tab["geom_type"][g] == 7}
    q0 = np.asarray(grasp_state(64, seed=0)["qpos"]); q0 = q0.T if q0.shape[0] == 19 else q0
    for q in q0:
        d1.set_state(qpos=q, qvel=np.zeros(18)); d1.forward()
        assert not any(c[0] in meshes or c[1] in meshes for c in _contact_table(d1))
    rng = np.random.default_rng(0)
    seen = 0; which = set()
    for trial in range(200):
        q = q0[rng.integers(64)].copy()
        q[:6] += rng.normal(0, 0.06, 6); q[6] = q[8] = np.clip(q[6] + rng.normal(0, 0.15), 0, 0.7)
        for d in (d0, d1): d.set_state(qpos=q, qvel=np.zeros(18)); d.forward()
        t0, t1 = _contact_table(d0), _contact_table(d1)
        if int(d1.get("ndrop", (1,), np.int32)[0]): continue
        assert [c[:4] for c in t1[:len(t0)]] == [c[:4] for c in t0]                         # the primitive pairs are untouched, and in front
        mc = [c for c in t1[len(t0):] if c[1] == scope]
        assert all(c[0] in meshes for c in t1[len(t0):])
        if not mc: continue
        seen += 1
        assert all(c[2] == 4 and c[3] < 0 for c in mc) and len(mc) % 2 == 0
        for a, b in zip(mc[0::2], mc[1::2]):
            assert b[0] == a[0] + 1 and a[3] == b[3] and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
            which.add(tab["geom_mesh"][a[0]])
    print("\nmeshes seen on the cube:", sorted(which))
    assert seen > 50 and len(which) >= 3
