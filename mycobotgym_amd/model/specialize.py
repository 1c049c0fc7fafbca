"""Compiled-model table -> the numeric parameter block the HIP kernels consume (``mcg_model``).

The kernels hard-code the *structure* of the MyCobot-280 scene (a 6-hinge arm whose joint axes
are coordinate axes, a planar 6-hinge gripper hanging off link6 with two loop closures and one
joint coupling, optionally one free cube) and read every *number* from this block, so model
variants (legacy / exact mesh inertia, fetch keyframe, domain-randomised cube) share one binary.

Joint-less bodies are welded into their nearest jointed ancestor (flange, camera frames,
gripper_base, gripper_tcp -> link6; finger layers -> fingers), giving 12 moving bodies + cube:

    0..5  link1..link6      6 right_gear   7 right_finger   8 left_gear   9 left_finger
    10 right_hinge  11 left_hinge   (12 object0)

All inertial data are expressed in the body's own frame about the body origin (= joint anchor):
``mass``, ``mc`` = mass * com, ``inertia`` = (xx, yy, zz, xy, xz, yz) about the origin.
The structural assumptions are asserted here so that a model that violates them fails loudly.
"""
from __future__ import annotations

import numpy as np

from .mjcf import quat_to_mat
from .refdyn import invweight0

NB_ARM = 6
NB_ROBOT = 12
MOVING = ["link1", "link2", "link3", "link4", "link5", "link6", "right_gear_link", "right_finger_link",
          "left_gear_link", "left_finger_link", "right_hinge_link", "left_hinge_link"]
PARENT = [-1, 0, 1, 2, 3, 4, 5, 6, 5, 8, 5, 5]
AXIS_K = [2, 0, 0, 0, 2, 0, 1, 1, 1, 1, 1, 1]          # must match AXK / AXS in csrc/mcg_dynamics.hpp
AXIS_S = [-1.0, -1.0, 1.0, -1.0, -1.0, -1.0, 1.0, -1.0, -1.0, 1.0, 1.0, 1.0]
MINIMP, MAXIMP, MINVAL = 1e-4, 0.9999, 1e-15


def _axis_code(ax):
    k = int(np.argmax(np.abs(ax)))
    assert abs(abs(ax[k]) - 1.0) < 1e-12, f"joint axis {ax} is not a coordinate axis"
    return k, float(np.sign(ax[k]))


def _solparams(solref, solimp, timestep):
    """(K, B, d0, dmax, width, midpoint, power, 1/width, 1/mid^(p-1), 1/(1-mid)^(p-1)); refsafe + solimp clamps applied."""
    tc, damp = float(solref[0]), float(solref[1])
    d0, dmax, width, mid, power = [float(x) for x in solimp]
    d0 = min(max(d0, MINIMP), MAXIMP); dmax = min(max(dmax, MINIMP), MAXIMP)
    width = max(width, 0.0); mid = min(max(mid, MINIMP), MAXIMP); power = max(power, 1.0)
    if tc > 0:
        tc = max(tc, 2 * timestep)
        K = 1.0 / max(MINVAL, dmax * dmax * tc * tc * damp * damp)
        B = 2.0 / max(MINVAL, dmax * tc)
    else:
        K = -tc / max(MINVAL, dmax * dmax)
        B = -damp / max(MINVAL, dmax)
    invw = 1.0 / width if width > MINVAL else 0.0
    return [K, B, d0, dmax, width, mid, power, invw, 1.0 / mid ** (power - 1), 1.0 / (1 - mid) ** (power - 1)]


def mix_contact(m, g1, g2):
    """[RECALL mj_contactParam] pair parameters of two geoms with equal priority and solmix."""
    condim = max(m["geom_condim"][g1], m["geom_condim"][g2])
    f = np.maximum(np.asarray(m["geom_friction"][g1]), np.asarray(m["geom_friction"][g2]))
    r1, r2 = np.asarray(m["geom_solref"][g1]), np.asarray(m["geom_solref"][g2])
    solref = 0.5 * (r1 + r2) if (r1[0] > 0 and r2[0] > 0) else np.minimum(r1, r2)
    solimp = 0.5 * (np.asarray(m["geom_solimp"][g1]) + np.asarray(m["geom_solimp"][g2]))
    return condim, [f[0], f[0], f[1], f[2], f[2]], solref, solimp


def specialize(m: dict, weld_rule: str = "common", contact_rule: str = "mujoco") -> dict:
    """m: table from ``load_model`` (full scene; the cube may have been dropped).
    contact_rule: "mujoco" = Rpy = 2 mu^2 R for a contact's pyramid rows (the rule as recalled) | "keyframe" = 4 mu^2 R, the one single
    change that reproduces the cube's rest height in the reference's keyframes (oracle/RULE_STUDY.md K1; oracle rule[3] = 2)."""
    name2id = {n: i for i, n in enumerate(m["body_name"])}
    has_cube = "object0" in name2id
    ids = [name2id[n] for n in MOVING] + ([name2id["object0"]] if has_cube else [])
    nb = len(ids)
    h = m["opt"]["timestep"]
    out = {"has_cube": has_cube, "timestep": h}

    # --- base (static) frame
    base = name2id["mycobot"]
    assert m["body_parent"][base] == 0 and m["body_dofnum"][base] == 0
    Rb = quat_to_mat(np.asarray(m["body_quat"][base]))
    out["base_pos"] = np.asarray(m["body_pos"][base], dtype=float)
    out["base_mat"] = Rb
    bq = np.asarray(m["body_quat"][base], dtype=float)
    out["base_quat"] = bq / np.linalg.norm(bq)         # xquat of the arm's chain starts here (mocap weld residual)
    out["gravity_base"] = Rb.T @ (-np.asarray(m["opt"]["gravity"], dtype=float))

    # --- weld joint-less descendants into each moving body
    def fixed_children(b):
        return [c for c in range(m["nbody"]) if m["body_parent"][c] == b and m["body_dofnum"][c] == 0]

    r = np.zeros((13, 3)); mass = np.zeros(13); mc = np.zeros((13, 3)); inertia = np.zeros((13, 6))
    axis_k = []; axis_s = []
    weld_frames = {}   # body id -> (root index, R, p) of every body welded into a moving body
    for i, b in enumerate(ids):
        free = m["body_dofnum"][b] == 6
        if not free:
            j = [jj for jj in range(m["njnt"]) if m["jnt_body"][jj] == b]
            assert len(j) == 1 and np.allclose(m["jnt_pos"][j[0]], 0), "hinge anchored at the body origin expected"
            k, s = _axis_code(np.asarray(m["jnt_axis"][j[0]])); axis_k.append(k); axis_s.append(s)
            assert np.allclose(m["body_quat"][b], [1, 0, 0, 0]), "moving bodies are expected to have identity quat"
        # origin in the parent moving body's frame (walk up through joint-less bodies)
        p = m["body_parent"][b]; off = np.asarray(m["body_pos"][b], dtype=float)
        while p != 0 and m["body_dofnum"][p] == 0 and p != base:
            assert np.allclose(m["body_quat"][p], [1, 0, 0, 0])
            off = off + np.asarray(m["body_pos"][p]); p = m["body_parent"][p]
        if i < NB_ROBOT:
            expect = base if PARENT[i] < 0 else ids[PARENT[i]]
            assert p == expect, f"{m['body_name'][b]}: unexpected tree structure"
        r[i] = off
        # composite of b and its welded descendants
        stack = [(b, np.eye(3), np.zeros(3))]
        I = np.zeros((3, 3))
        while stack:
            k_, R, pos = stack.pop()
            weld_frames[k_] = (i, R, pos)
            mk = m["body_mass"][k_]
            if mk > 0:
                c = pos + R @ np.asarray(m["body_ipos"][k_])
                Rc = R @ quat_to_mat(np.asarray(m["body_iquat"][k_]))
                Ic = Rc @ np.diag(np.asarray(m["body_inertia"][k_])) @ Rc.T
                mass[i] += mk; mc[i] += mk * c
                I += Ic + mk * (c @ c * np.eye(3) - np.outer(c, c))
            for ch in fixed_children(k_):
                Rch = R @ quat_to_mat(np.asarray(m["body_quat"][ch]))
                stack.append((ch, Rch, pos + R @ np.asarray(m["body_pos"][ch])))
        inertia[i] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]
    out.update(r=r, mass=mass, mc=mc, inertia=inertia, axis_k=axis_k, axis_s=axis_s)
    assert axis_k == AXIS_K and axis_s == AXIS_S, "joint axes differ from the structure compiled into the kernels"
    # structural facts the kernels rely on
    assert [axis_k[i] for i in range(6, 12)] == [1] * 6, "gripper joints must all turn about local y"

    # --- dofs
    nv = m["nv"]
    arm = np.zeros(18); damp = np.zeros(18)
    arm[:nv] = m["dof_armature"]; damp[:nv] = m["dof_damping"]
    out["armature"] = arm; out["damping"] = damp
    body = np.zeros((13, 16))         # mcg_body rows: r(3) mass mc(3) inertia(6) armature damping pad
    body[:, 0:3] = r; body[:, 3] = mass; body[:, 4:7] = mc; body[:, 7:13] = inertia
    body[:12, 13] = arm[:12]; body[:12, 14] = damp[:12]
    out["body"] = body
    out["cube_damping"] = damp[12:18]
    rng = np.zeros((12, 2)); limited = []
    lim_par = np.zeros((12, 10))
    for i in range(12):
        j = [jj for jj in range(m["njnt"]) if m["jnt_body"][jj] == ids[i]][0]
        assert m["jnt_dofadr"][j] == i and m["jnt_qposadr"][j] == i
        limited.append(bool(m["jnt_limited"][j])); rng[i] = m["jnt_range"][j]
        lim_par[i] = _solparams(m["jnt_solref"][j], m["jnt_solimp"][j], h)
    assert limited == [True] * 10 + [False] * 2, "joints 0-9 limited, the two couplers unlimited"
    out["jnt_range"] = rng; out["limit_par"] = lim_par

    # --- inverse weights at qpos0 (constraint regularisation)
    biw, diw = invweight0(m)
    out["limit_diag"] = np.concatenate([diw[:12]])

    # --- equalities: connect(right_finger, right_hinge), connect(left_finger, left_hinge), joint(gear R = gear L)
    eq = list(m["eq"])
    weld = [e for e in eq if e["type"] == 1]
    eq = [e for e in eq if e["type"] != 1]
    assert len(weld) <= 1 and [e["type"] for e in eq[:3]] == [0, 0, 2]
    assert (eq[0]["obj1"], eq[0]["obj2"]) == (ids[7], ids[10]) and (eq[1]["obj1"], eq[1]["obj2"]) == (ids[9], ids[11])
    assert (eq[2]["obj1"], eq[2]["obj2"]) == (6, 8) and np.allclose(eq[2]["data"][:5], [0, 1, 0, 0, 0])
    out["eq_anchor1"] = np.array([eq[0]["data"][0:3], eq[1]["data"][0:3]])
    out["eq_anchor2"] = np.array([eq[0]["data"][3:6], eq[1]["data"][3:6]])
    out["eq_par"] = np.array([_solparams(e["solref"], e["solimp"], h) for e in eq[:3]])
    out["eq_diag"] = np.array([biw[ids[7], 0] + biw[ids[10], 0], biw[ids[9], 0] + biw[ids[11], 0], diw[6] + diw[8]])

    # --- actuators: 6 joint servos + tendon servo over (gear R, gear L)
    acts = list(m["actuators"])
    if len(acts) == 1:       # mocap variant (mocap_actuators.xml): no arm servos; zero-gain placeholders keep the layout
        assert weld, "a model without arm actuators is expected to carry the mocap weld"
        acts = [dict(trntype="joint", trnid=i, gear=1.0, gainprm=[0.0, 0, 0], biasprm=[0.0, 0, 0], ctrllimited=True,
                     forcelimited=True, ctrlrange=[-1.0, 1.0], forcerange=[-1.0, 1.0]) for i in range(6)] + acts
    assert len(acts) == 7 and [a["trntype"] for a in acts] == ["joint"] * 6 + ["tendon"]
    assert [a["trnid"] for a in acts[:6]] == list(range(6)) and all(a["gear"] == 1 for a in acts)
    ten = m["tendons"][acts[6]["trnid"]]
    assert ten["joints"] == [6, 8]
    out["act_gain"] = np.array([a["gainprm"][0] for a in acts])
    out["act_bias"] = np.array([a["biasprm"] for a in acts])
    assert all(a["ctrllimited"] and a["forcelimited"] for a in acts)
    out["act_ctrlrange"] = np.array([a["ctrlrange"] for a in acts])
    out["act_forcerange"] = np.array([a["forcerange"] for a in acts])
    out["tendon_coef"] = np.array(ten["coefs"])

    # --- sites
    s = m["site_name"].index("EEF")
    i6, R, p = weld_frames[m["site_body"][s]]
    assert i6 == 5 and np.allclose(R, np.eye(3)) and np.allclose(m["site_quat"][s], [1, 0, 0, 0])
    out["site_eef"] = p + np.asarray(m["site_pos"][s])
    # site target0 hangs on the world body: stage_rewards reads its MJCF position unless the env is rendering (mycobot.py:422, 309-311)
    assert contact_rule in ("mujoco", "keyframe")
    out["contact_rpy"] = 4.0 if contact_rule == "keyframe" else 2.0
    out["target0"] = (np.asarray(m["site_pos"][m["site_name"].index("target0")], dtype=np.float64)
                      if "target0" in m["site_name"] else np.array([-0.15, 0.0, 0.21]))

    # --- mocap weld (mocap.xml:15-20): body1 = the mocap body (static), body2 = gripper_tcp, welded into link6
    out["weld_on"] = 0.0
    out["weld_par"] = np.zeros(10); out["weld_diag"] = np.zeros(2); out["weld_anchor"] = np.zeros(3)
    out["weld_relpos"] = np.zeros(3); out["weld_relquat"] = np.array([1.0, 0, 0, 0]); out["weld_torquescale"] = 1.0
    if weld:
        w = weld[0]
        b1, b2 = w["obj1"], w["obj2"]
        assert m["body_mocap"][b1] and m["body_name"][b2] == "gripper_tcp"
        i6, R, p = weld_frames[b2]
        assert i6 == 5 and np.allclose(R, np.eye(3)), "gripper_tcp: welded into link6 without rotation expected"
        data = np.asarray(w["data"], dtype=float)
        assert np.allclose(data[0:3], 0), "weld anchor at the origin of gripper_tcp expected (xpos[tcp] is the weld point)"
        out["weld_on"] = 1.0
        out["weld_par"] = np.array(_solparams(w["solref"], w["solimp"], h))
        # all six rows carry the translational inverse weights (oracle/mco_physics.c, pinned by the keyframe equilibrium)
        # weld_rule "common": one (translational) weight for the six rows -- the variant the reference's mocap keyframe supports;
        # "mujoco": mj_diagApprox as recalled, the rotational inverse weight on rows 3-5 (oracle/RULE_STUDY.md, K2)
        assert weld_rule in ("common", "mujoco")
        part = 1 if weld_rule == "mujoco" else 0
        out["weld_diag"] = np.array([biw[b1, 0] + biw[b2, 0], biw[b1, part] + biw[b2, part]])
        out["weld_anchor"] = p + data[0:3]               # anchor (body2 frame) in the link6 frame
        out["weld_relpos"] = data[3:6]; out["weld_relquat"] = data[6:10]; out["weld_torquescale"] = float(data[10])
        out["mocap_pose0"] = np.concatenate([np.asarray(m["body_pos"][b1], float), np.asarray(m["body_quat"][b1], float)])

    # --- contact geometry of the PickAndPlace scene: cube, table top, finger pads
    if has_cube:
        gname = {n: i for i, n in enumerate(m["geom_name"]) if n}
        gc, gr, gl = gname["object0"], gname["right_finger_layer"], gname["left_finger_layer"]
        gt = [g for g in range(m["ngeom"]) if m["body_name"][m["geom_body"][g]] == "table"][0]
        out["cube_half"] = np.asarray(m["geom_size"][gc], dtype=float)
        tb = m["geom_body"][gt]
        out["table_pos"] = np.asarray(m["body_pos"][tb]) + np.asarray(m["geom_pos"][gt])
        out["table_half"] = np.asarray(m["geom_size"][gt], dtype=float)
        pads = []
        for g, fing in ((gr, 7), (gl, 9)):
            root, R, p = weld_frames[m["geom_body"][g]]
            assert root == fing and np.allclose(R, np.eye(3)) and np.allclose(m["geom_quat"][g], [1, 0, 0, 0])
            pads.append(np.concatenate([p + np.asarray(m["geom_pos"][g]), np.asarray(m["geom_size"][g])]))
        out["pad_box"] = np.array(pads)      # centre (finger frame) | half sizes
        cp = []
        for (ga, gb) in ((gt, gc), (gr, gc), (gl, gc), (gt, gr), (gt, gl)):
            condim, fri, solref, solimp = mix_contact(m, ga, gb)
            assert condim == 4
            cp.append(np.concatenate([_solparams(solref, solimp, h), fri]))
        out["contact_par"] = np.array(cp)    # rows: table-cube, right pad-cube, left pad-cube, table-right pad, table-left pad
        bt = lambda g: biw[m["geom_body"][g]]
        out["contact_diag"] = np.array([[bt(ga)[0] + bt(gb)[0], bt(ga)[1] + bt(gb)[1]]
                                        for (ga, gb) in ((gt, gc), (gr, gc), (gl, gc), (gt, gr), (gt, gl))])
        out["cube_invweight"] = np.array([diw[12], diw[15]])
        out["geom_friction0"] = np.array([m["geom_friction"][gt][0], m["geom_friction"][gr][0], m["geom_friction"][gc][0]])
        # structural facts the cube kernels rely on: CoM at the body origin, principal axes = body axes
        assert np.allclose(mc[12], 0) and np.allclose(inertia[12][3:], 0), "cube: centred, axis-aligned inertia expected"
        out["geom_ids"] = {"table": gt, "cube": gc, "pad_r": gr, "pad_l": gl}
        # --- convex-mesh collision: the fourteen mesh geoms' collision polytopes (model/polytope.py) in engine body frames
        from . import polytope as pt
        mbox = np.zeros((pt.NMESH, 6)); tran_mesh = np.zeros(pt.NMESH); mult = set(); mesh_sigs = set(); frames = []
        for mi, nm in enumerate(pt.MESH_NAMES):
            gs = [g for g in range(m["ngeom"]) if m["geom_type"][g] == 7 and m["geom_mesh"][g] == nm
                  and m["geom_contype"][g] and m["geom_conaffinity"][g]]
            assert gs, nm
            g = gs[0]
            root, R, pw = weld_frames[m["geom_body"][g]]
            assert root == pt.MESH_BODY[mi], f"{nm}: rides on engine body {root}, the kernels expect {pt.MESH_BODY[mi]}"
            assert np.allclose(m["geom_pos"][g], 0) and np.allclose(m["geom_quat"][g], [1, 0, 0, 0])
            frames.append((R, pw))
            tran_mesh[mi] = biw[m["geom_body"][g]][0]
            mult.add(len(gs))
            # ONE row of contact_par serves all meshes against the static geoms and one against the cube: every mesh geom must mix to the
            # same numbers, else the row of the last mesh would silently win
            c3, fri3, solref3, solimp3 = mix_contact(m, gt, g)
            c4, fri4, solref4, solimp4 = mix_contact(m, g, gc)
            assert c3 == 3 and c4 == 4
            mesh_sigs.add((tuple(np.round(fri3, 15)), tuple(np.round(solref3, 15)), tuple(np.round(solimp3, 15)), tuple(np.round(fri4, 15)),
                           tuple(np.round(solref4, 15)), tuple(np.round(solimp4, 15)), tuple(np.round(m["geom_friction"][g], 15))))
        assert len(mult) == 1
        assert len(mesh_sigs) == 1, "the mesh geoms differ in friction / solref / solimp: one contact_par row per mesh would be needed"
        cp.append(np.concatenate([_solparams(solref3, solimp3, h), fri3]))       # row 5: table - mesh (condim 3)
        cp.append(np.concatenate([_solparams(solref4, solimp4, h), fri4]))       # row 6: mesh - cube (condim 4)
        out["contact_par"] = np.array(cp)
        out["mesh_mult"] = float(mult.pop()); out["mesh_fric"] = float(m["geom_friction"][g][0])
        polys = pt.transform(pt.unpack(pt.load_asset()[0]), frames)
        for mi, P in enumerate(polys):
            v = P["verts"]
            mbox[mi] = np.concatenate([0.5 * (v.max(0) + v.min(0)), 0.5 * (v.max(0) - v.min(0))])
            bi = pt.MESH_BODY[mi]
            out["body"][bi, 15] = max(out["body"][bi, 15], float(np.linalg.norm(v, axis=1).max()))      # mcg_body.hull_rad: the free broad-phase number
        out["mesh_box"] = mbox
        out["polytopes"] = pt.pack(polys)           # the table block of mcg_create (not a field of mcg_model)
        # pair_tran: the summed translational inverse weights of a pair's two bodies (the static geoms' are zero)
        ptn = np.zeros(5 + 2 * pt.NMESH)
        ptn[0:5] = out["contact_diag"][:, 0]
        ptn[5:5 + pt.NMESH] = tran_mesh + bt(gt)[0]
        ptn[5 + pt.NMESH:] = tran_mesh + bt(gc)[0]
        out["pair_tran"] = ptn
    return out


def initial_gripper_xpos(m: dict, qpos) -> np.ndarray:
    """EEF site position at ``qpos`` (reference ``_env_setup``, mycobot.py:464-466)."""
    from .refdyn import kinematics
    kin = kinematics(m, qpos)
    return kin["site_xpos"][m["site_name"].index("EEF")].copy()
