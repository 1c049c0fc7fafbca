"""Pins the CPU oracle (mini-MuJoCo restatement) on everything that can be pinned without MuJoCo:
reference-internal known answers, an independent numpy formulation, analytic physics
invariants and published known-answer vectors (Philox / Random123).

PARITY UNPINNED against real MuJoCo 2.3.2: the package is absent here and the reference ships no golden vectors.
"""
import ctypes as C
import json

import numpy as np
import pytest

from tests.common import load_json


@pytest.fixture(scope="module")
def po(built):
    from oracle import pyoracle
    return pyoracle


def _zero_gains(tab, damping=False, eq=False, limits=False):
    tab = json.loads(json.dumps(tab))
    for a in tab["actuators"]:
        a["gainprm"] = [0, 0, 0]; a["biasprm"] = [0, 0, 0]
    if not damping:
        tab["dof_damping"] = [0.0] * tab["nv"]
    if not eq:
        tab["eq"] = []; tab["neq"] = 0
    if not limits:
        tab["jnt_limited"] = [False] * tab["njnt"]
    return tab


def test_fk_known_answers(po):
    tab = load_json("mycobot280")
    d = po.OracleData(po.OracleModel(tab))
    s = tab["site_name"].index("EEF")
    d.forward()
    assert np.allclose(d.get("site_xpos", (8, 3))[s], [0.0138673, 0.01864658, 0.61236], atol=1e-8)
    d.set_state(qpos=tab["keys"][0]["qpos"]); d.forward()
    assert np.allclose(d.get("site_xpos", (8, 3))[s], [-0.05154491, 0.01053502, 0.3448586], atol=2e-7)
    # quaternion of the EEF at the keyframe ~ the fetch target [0, -0.707, 0, 0.707] (mycobot.py:140)
    q = np.zeros(4); po.lib().mco_mat2quat.argtypes = [C.c_void_p, C.c_void_p]
    mat = np.ascontiguousarray(d.get("site_xmat", (8, 9))[s])
    po.lib().mco_mat2quat(q.ctypes.data_as(C.c_void_p), mat.ctypes.data_as(C.c_void_p))
    assert np.allclose(np.abs(q), [0.0176, 0.7053, 0.0078, 0.7087], atol=2e-3)


def test_mass_matrix_vs_jacobian_sum_and_invweight(po):
    from mycobotgym_amd.model.mjcf import _np_model
    from mycobotgym_amd.model.refdyn import kinematics, mass_matrix, invweight0
    for name in ("mycobot280", "mycobot280_reach"):
        tab = load_json(name); m = _np_model(tab)
        om = po.OracleModel(tab); d = po.OracleData(om)
        rng = np.random.default_rng(1)
        for _ in range(5):
            q = m["qpos0"].copy(); q[:12] = rng.uniform(-1.5, 1.5, 12)
            if om.nq == 19:
                q[12:15] = rng.normal(size=3) * 0.1; q[15:19] = rng.normal(size=4)
            d.set_state(qpos=q, qvel=rng.normal(size=om.nv)); d.forward()
            Mref = mass_matrix(m, kinematics(m, d.qpos.copy()))
            assert np.abs(d.dense("qM") - Mref).max() < 1e-15 + 1e-13 * np.abs(Mref).max()
        biw, diw = invweight0(m)
        assert np.allclose(om.get("dof_invweight0", om.nv), diw, rtol=1e-12)
        got = om.get("body_invweight0", 2 * m["nbody"]).reshape(-1, 2)
        assert np.allclose(got, biw, rtol=1e-11, atol=1e-300)


def test_bias_force_equals_lagrangian_derivative(po):
    """qfrc_bias = d/dt(dL/dqd) - dL/dq at qacc = 0, by central differences of the oracle's own M(q) and V(q)."""
    tab = _zero_gains(load_json("mycobot280_reach"))
    om = po.OracleModel(tab); d = po.OracleData(om)
    rng = np.random.default_rng(2)
    q0 = rng.uniform(-1, 1, 12); v = rng.normal(size=12)
    def MV(q):
        d.set_state(qpos=q, qvel=np.zeros(12)); d.forward()
        return d.dense("qM").copy() - np.diag(tab["dof_armature"]), d.energy()[0]
    eps = 1e-6
    dM = np.zeros((12, 12, 12)); dV = np.zeros(12)
    for k in range(12):
        e = np.zeros(12); e[k] = eps
        Mp, Vp = MV(q0 + e); Mm, Vm = MV(q0 - e)
        dM[k] = (Mp - Mm) / (2 * eps); dV[k] = (Vp - Vm) / (2 * eps)
    # c_i = sum_jk (dM_ij/dq_k - 0.5 dM_jk/dq_i) v_j v_k + dV/dq_i
    c = np.einsum("kij,j,k->i", dM, v, v) - 0.5 * np.einsum("ijk,j,k->i", dM, v, v) + dV
    d.set_state(qpos=q0, qvel=v); d.forward()
    assert np.abs(d.vec("qfrc_bias") - c).max() < 1e-6 * max(1.0, np.abs(c).max())


def test_energy_conservation_free_space(po):
    """Gravity on; damping, actuators, equalities and limits off: total energy drifts only at O(h) (semi-implicit Euler)."""
    tab = _zero_gains(load_json("mycobot280_reach"))
    tab["opt"]["timestep"] = 1e-4
    om = po.OracleModel(tab); d = po.OracleData(om)
    rng = np.random.default_rng(3)
    q = rng.uniform(-0.5, 0.5, 12); q[6:] *= 0.1
    d.set_state(qpos=q, qvel=np.zeros(12)); d.forward()
    e0 = sum(d.energy())
    d.step(2000); d.forward()
    pe, ke = d.energy()
    assert ke > 1e-5                       # it really moved
    assert abs(pe + ke - e0) < 2e-3 * ke


def test_free_cube_ballistic(po):
    """Cube alone (contacts off): exact free fall, constant horizontal and angular momentum (symmetric inertia)."""
    tab = load_json("mycobot280"); tab["dof_damping"][12:] = [0.0] * 6
    om = po.OracleModel(tab, enable_contact=False); d = po.OracleData(om)
    v = np.zeros(18); v[12:18] = [0.1, -0.2, 0.3, 1.0, -2.0, 0.5]
    d.set_state(qvel=v)
    n, h = 200, tab["opt"]["timestep"]
    d.step(n)
    assert np.allclose(d.qvel[12:14], [0.1, -0.2], atol=1e-12) and np.isclose(d.qvel[14], 0.3 - 9.81 * n * h, atol=1e-10)
    assert np.allclose(d.qvel[15:18], [1.0, -2.0, 0.5], atol=1e-9)
    assert np.isclose(np.linalg.norm(d.qpos[15:19]), 1.0, atol=1e-12)
    assert np.isclose(d.qpos[12], -0.05 + 0.1 * n * h, atol=1e-12)


def test_constraint_solver_kkt_and_minimum(po):
    """At the solver's answer: M (a - a_smooth) = J^T f, f = -D (J a - aref) on active rows, f >= 0 on limit rows,
    and the cost does not decrease under random perturbations (unique minimiser of a strictly convex problem)."""
    tab = load_json("mycobot280_reach")
    om = po.OracleModel(tab); d = po.OracleData(om)
    rng = np.random.default_rng(4)
    for trial in range(20):
        q = rng.uniform(-1, 1, 12); q[6] = rng.uniform(-0.05, 0.75); q[8] = rng.uniform(-0.05, 0.75)
        if trial % 3 == 0:
            q[2] = 2.97 + rng.uniform(0, 0.02)          # arm joint beyond its limit
        d.set_state(qpos=q, qvel=rng.normal(size=12) * 0.5, ctrl=rng.uniform(-1, 1, 7)); d.forward()
        nefc = int(d.get("nefc", (1,), np.int32)[0]); ne = int(d.get("ne", (1,), np.int32)[0])
        J = d.get("efc_J", (224, 24))[:nefc, :12]; D = d.vec("efc_D", nefc); aref = d.vec("efc_aref", nefc)
        f = d.vec("efc_force", nefc); a = d.vec("qacc"); a_s = d.vec("qacc_smooth"); M = d.dense("qM")
        assert np.abs(M @ (a - a_s) - J.T @ f).max() < 1e-7 * max(1.0, np.abs(J.T @ f).max())
        r = J @ a - aref
        assert np.all(f[ne:] >= 0)
        active = np.concatenate([np.ones(ne, bool), r[ne:] < 0])
        assert np.allclose(f[active], -D[active] * r[active], rtol=1e-9, atol=1e-12) and np.all(f[~active] == 0)
        def cost(x):
            rr = J @ x - aref
            act = np.concatenate([np.ones(ne, bool), rr[ne:] < 0])
            return 0.5 * (x - a_s) @ M @ (x - a_s) + 0.5 * np.sum(D[act] * rr[act] ** 2)
        c0 = cost(a)
        for _ in range(10):
            assert cost(a + rng.normal(size=12) * 1e-3 * (1 + np.abs(a))) >= c0 - 1e-9 * abs(c0)


def test_gripper_loop_stays_closed(po):
    """The soft connects hold the four-bar loops: anchor mismatch stays far below a millimetre while the gripper moves."""
    tab = load_json("mycobot280_reach")
    om = po.OracleModel(tab); d = po.OracleData(om)
    d.set_state(ctrl=[0, 0, 0, 0, 0, 0, 1.0])
    worst = 0.0
    for _ in range(30):
        d.step(20)
        worst = max(worst, np.abs(d.vec("efc_pos", 7)[:6]).max())
    assert d.qpos[6] > 0.3 and abs(d.qpos[6] - d.qpos[8]) < 5e-3       # closed and symmetric (joint coupling)
    assert worst < 5e-4


def test_philox_known_answers(po):
    """Random123 philox4x32-10 known-answer vectors."""
    L = po.lib()
    def ph(ctr, key):
        c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
        L.mco_philox4x32_10(c, k, o); return list(o)
    assert ph([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ph([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ph([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_quaternion_helpers_vs_scipy(po):
    from scipy.spatial.transform import Rotation as R
    L = po.lib()
    for f in ("mco_mat2quat", "mco_mulquat", "mco_quat2mat"):
        getattr(L, f).argtypes = [C.c_void_p] * (3 if f == "mco_mulquat" else 2)
    L.mco_quat2vel.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
    rng = np.random.default_rng(5)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for _ in range(50):
        rot = R.random(random_state=int(rng.integers(1 << 30)))
        mat = np.ascontiguousarray(rot.as_matrix().ravel()); q = np.zeros(4)
        L.mco_mat2quat(p(q), p(mat))
        xyzw = rot.as_quat(); ref = np.array([xyzw[3], *xyzw[:3]])
        assert min(np.abs(q - ref).max(), np.abs(q + ref).max()) < 1e-12
        back = np.zeros(9); L.mco_quat2mat(p(back), p(q)); assert np.abs(back - mat).max() < 1e-12
        v = np.zeros(3); L.mco_quat2vel(p(v), p(q), 50.0)
        rv = rot.as_rotvec() / 50.0
        assert np.abs(v - rv).max() < 1e-12


def test_mocap_weld_keyframe_is_an_equilibrium(po):
    """Reference data known answer for the mocap variant: the `fetch_env` keyframe of mycobot280_mocap.xml:6-9
    (qpos + mocap pose) was saved with the arm hanging on the weld.  It is a wrist-singular pose: the weld cannot be met,
    the residual (1.2 mm) is parallel to the null direction of J^T and stays there only if the six weld rows are equally
    stiff.  The restated weld must hold the keyframe: residual unchanged to 2 %, joints within the keyframe's rounding
    along the singular direction."""
    tab = load_json("mycobot280_mocap")
    m = po.OracleModel(tab)
    d = po.OracleData(m)
    key = tab["keys"][0]
    d.set_state(qpos=key["qpos"], qvel=key["qvel"], ctrl=key["ctrl"])
    d.set_mocap(key["mpos"], key["mquat"])
    d.forward()
    assert int(d.get("nefc", (1,), np.int32)[0]) == 13                 # weld 6 + two connects 6 + joint 1
    r0 = d.get("efc_pos", (224,))[:6].copy()
    J = d.get("efc_J", (224, 24))[:6, :6]
    n = np.linalg.svd(J.T)[2][-1]
    assert abs(r0 @ n) / np.linalg.norm(r0) > 0.999                    # the reference's residual lies in the singular direction
    assert 1.0e-3 < np.linalg.norm(r0[:3]) < 1.3e-3                    # "about 1 mm below mpos" 
    q0 = np.array(d.qpos[:6])
    d.step(4000)
    d.forward()
    r = d.get("efc_pos", (224,))[:6]
    assert np.abs(d.qvel[:12]).max() < 1e-8                            # settled
    assert np.linalg.norm(r - r0) < 0.02 * np.linalg.norm(r0)
    assert np.abs(np.array(d.qpos[:6]) - q0).max() < 5e-3


def test_mocap_weld_rest_pose_has_no_residual(po):
    tab = load_json("mycobot280_mocap")
    d = po.OracleData(po.OracleModel(tab))
    d.forward()
    assert np.abs(d.get("efc_pos", (224,))[:6]).max() < 1e-12           # FK(EEF; qpos0) == mocap rest pose, mocap.xml:3
    tcp = tab["body_name"].index("gripper_tcp")
    assert np.allclose(d.get("xpos", (32, 3))[tcp], tab["body_pos"][tab["body_name"].index("robot0:mocap")], atol=1e-8)


@pytest.mark.parametrize("variant", ["mycobot280_mocap", "mycobot280_mocap_exactmesh"])
def test_mocap_keyframe_gripper_deflections_are_an_equilibrium(po, variant):
    """Second known answer in the same keyframe (mycobot280_mocap.xml:7): its six gripper angles (5e-5 ... 3e-4 rad) are the
    sag of the closed-loop gripper under gravity against the finger actuator's affine bias (-100 len), the gear coupling and the
    two soft connects.  The restated gripper must keep them: the gear joints to the keyframe's rounding (1e-6 relative), the
    finger and hinge joints, which feel the connects' regularisation (invweight0 of the whole arm), to 2e-3."""
    tab = load_json(variant)
    d = po.OracleData(po.OracleModel(tab))
    key = tab["keys"][0]
    d.set_state(qpos=key["qpos"], qvel=key["qvel"], ctrl=key["ctrl"])
    d.set_mocap(key["mpos"], key["mquat"])
    d.forward()
    g0 = np.array(d.qpos[6:12])
    assert np.all(np.abs(g0) > 4e-5) and np.all(np.abs(g0) < 4e-4)
    d.step(6000)
    g1 = np.array(d.qpos[6:12])
    rel = np.abs(g1 - g0) / np.abs(g0)
    assert rel[0] < 5e-6 and rel[2] < 5e-6            # gear R, gear L
    assert rel[[1, 3, 4, 5]].max() < 2e-3             # fingers, hinges
