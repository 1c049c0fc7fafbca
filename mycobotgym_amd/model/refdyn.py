"""Plain-numpy rigid-body helpers over the compiled-model tables.

Used (a) by the MJCF compiler for the quantities MuJoCo derives at qpos0 during compilation
(equality anchors; ``body_invweight0`` / ``dof_invweight0`` of ``mj_setConst`` [RECALL]) and
(b) by the HIP model specialiser to precompute welded-body constants.  It is deliberately the
textbook Jacobian-sum formulation (M = sum_b m Jp^T Jp + Jr^T I Jr), i.e. a third, independent
statement of the mass matrix next to the oracle's composite-rigid-body pass and the HIP
kernel's body-local recursion, so the three can be checked against each other in tests.
"""
from __future__ import annotations

import numpy as np

JNT_FREE, JNT_HINGE = 0, 3


def _qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def _q2m(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def kinematics(m: dict, qpos) -> dict:
    """Forward kinematics of every body / joint / site / geom (world frame)."""
    qpos = np.asarray(qpos, dtype=np.float64)
    nb = m["nbody"]
    xpos = np.zeros((nb, 3)); xquat = np.zeros((nb, 4)); xquat[0, 0] = 1
    xmat = np.zeros((nb, 3, 3)); xmat[0] = np.eye(3)
    njnt = m["njnt"]
    xanchor = np.zeros((njnt, 3)); xaxis = np.zeros((njnt, 3))
    body_jnts = [[] for _ in range(nb)]
    for j in range(njnt):
        body_jnts[m["jnt_body"][j]].append(j)
    for b in range(1, nb):
        p = m["body_parent"][b]
        jl = body_jnts[b]
        if jl and m["jnt_type"][jl[0]] == JNT_FREE:
            a = m["jnt_qposadr"][jl[0]]
            xpos[b] = qpos[a:a + 3]
            q = qpos[a + 3:a + 7]
            xquat[b] = q / np.linalg.norm(q)
            xanchor[jl[0]] = xpos[b]; xaxis[jl[0]] = xmat[p] @ np.array([0, 0, 1.0])
        else:
            xpos[b] = xpos[p] + xmat[p] @ np.asarray(m["body_pos"][b])
            xquat[b] = _qmul(xquat[p], np.asarray(m["body_quat"][b]))
            for j in jl:
                R = _q2m(xquat[b])
                xanchor[j] = xpos[b] + R @ np.asarray(m["jnt_pos"][j])
                xaxis[j] = R @ np.asarray(m["jnt_axis"][j])
                ang = qpos[m["jnt_qposadr"][j]] - m["qpos0"][m["jnt_qposadr"][j]]
                ax = np.asarray(m["jnt_axis"][j])
                ql = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
                xquat[b] = _qmul(xquat[b], ql)
                xpos[b] = xanchor[j] - _q2m(xquat[b]) @ np.asarray(m["jnt_pos"][j])
            xquat[b] /= np.linalg.norm(xquat[b])
        xmat[b] = _q2m(xquat[b])
    xipos = np.zeros((nb, 3)); ximat = np.zeros((nb, 3, 3))
    for b in range(nb):
        xipos[b] = xpos[b] + xmat[b] @ np.asarray(m["body_ipos"][b])
        ximat[b] = xmat[b] @ _q2m(np.asarray(m["body_iquat"][b]))
    out = {"xpos": xpos, "xquat": xquat, "xmat": xmat, "xipos": xipos, "ximat": ximat,
           "xanchor": xanchor, "xaxis": xaxis}
    if "site_body" in m:
        ns = len(m["site_body"])
        sp = np.zeros((ns, 3)); sm = np.zeros((ns, 3, 3))
        for s in range(ns):
            b = m["site_body"][s]
            sp[s] = xpos[b] + xmat[b] @ np.asarray(m["site_pos"][s])
            sm[s] = xmat[b] @ _q2m(np.asarray(m["site_quat"][s]))
        out["site_xpos"] = sp; out["site_xmat"] = sm
    if "geom_body" in m:
        ng = len(m["geom_body"])
        gp = np.zeros((ng, 3)); gm = np.zeros((ng, 3, 3))
        for g in range(ng):
            b = m["geom_body"][g]
            gp[g] = xpos[b] + xmat[b] @ np.asarray(m["geom_pos"][g])
            gm[g] = xmat[b] @ _q2m(np.asarray(m["geom_quat"][g]))
        out["geom_xpos"] = gp; out["geom_xmat"] = gm
    return out


def jac_point(m: dict, kin: dict, body: int, point) -> tuple:
    """Translational and rotational Jacobian (3 x nv each) of ``point`` fixed to ``body``."""
    nv = m["nv"]
    jp = np.zeros((3, nv)); jr = np.zeros((3, nv))
    b = body
    while b != 0:
        for j in range(m["njnt"]):
            if m["jnt_body"][j] != b:
                continue
            d = m["jnt_dofadr"][j]
            if m["jnt_type"][j] == JNT_FREE:
                jp[:, d:d + 3] = np.eye(3)
                R = kin["xmat"][b]
                for k in range(3):
                    ax = R[:, k]
                    jr[:, d + 3 + k] = ax
                    jp[:, d + 3 + k] = np.cross(ax, point - kin["xpos"][b])
            else:
                ax = kin["xaxis"][j]
                jr[:, d] = ax
                jp[:, d] = np.cross(ax, point - kin["xanchor"][j])
        b = m["body_parent"][b]
    return jp, jr


def mass_matrix(m: dict, kin: dict) -> np.ndarray:
    nv = m["nv"]
    M = np.diag(np.asarray(m["dof_armature"], dtype=np.float64)) if nv else np.zeros((0, 0))
    for b in range(1, m["nbody"]):
        mass = m["body_mass"][b]
        if mass == 0 and not np.any(np.asarray(m["body_inertia"][b])):
            continue
        jp, jr = jac_point(m, kin, b, kin["xipos"][b])
        Iw = kin["ximat"][b] @ np.diag(np.asarray(m["body_inertia"][b])) @ kin["ximat"][b].T
        M = M + mass * jp.T @ jp + jr.T @ Iw @ jr
    return M


def invweight0(m: dict):
    """[RECALL mj_setConst/set0] inverse weights at qpos0.

    body_invweight0[b] = (mean diag of Jp M^-1 Jp^T, mean diag of Jr M^-1 Jr^T) at the body CoM;
    dof_invweight0[d]  = (M^-1)_dd for hinges; free joints average the 3 translational and the
    3 rotational entries.  Bodies welded to the world get 0.
    """
    kin = kinematics(m, m["qpos0"])
    M = mass_matrix(m, kin)
    Minv = np.linalg.inv(M)
    nb = m["nbody"]
    biw = np.zeros((nb, 2))
    for b in range(1, nb):
        if m["body_weldid"][b] == 0:
            continue
        jp, jr = jac_point(m, kin, b, kin["xipos"][b])
        biw[b, 0] = np.trace(jp @ Minv @ jp.T) / 3.0
        biw[b, 1] = np.trace(jr @ Minv @ jr.T) / 3.0
    diw = np.zeros(m["nv"])
    for j in range(m["njnt"]):
        d = m["jnt_dofadr"][j]
        if m["jnt_type"][j] == JNT_FREE:
            diw[d:d + 3] = np.mean(np.diag(Minv)[d:d + 3])
            diw[d + 3:d + 6] = np.mean(np.diag(Minv)[d + 3:d + 6])
        else:
            diw[d] = Minv[d, d]
    return biw, diw
