"""CPU restatement of the reference's IK back-projection solver -- TEST INFRASTRUCTURE ONLY.

SURVEY 8(f) rank 1: `towr_utils.apply_results` (towr_utils.py:779-857) projects the physics stage's output (COM, feet)
back onto the skeleton with `JacobianInverseKinematicsCK(anim, targets, translate=True, iterations=30,
smoothness=0.001, damping=7.0)` (src/skeleton_fitting/ik/InverseKinematics.py:326-561), a damped-least-squares Jacobian
IK that is independent per frame except for the smoothness term.  This file restates that algorithm with plain numpy
arrays (quaternions as (..., 4) arrays in w, x, y, z order), each function citing the reference lines it follows.

Parity is PINNED: tests/golden/ik_golden.npz holds inputs and outputs of the reference solver itself, generated in the
build container by tests/golden/make_ik_golden.py; tests/test_ik_oracle.py checks this restatement against them.

Only tests/ may import this module.  The product path of this row is include/chd_ik.h + contact-human-dynamics_amd/csrc/
chd_ik* (HIP, dual-form step); it never touches this file.
"""
import numpy as np


# ---- quaternions (Quaternions.py) --------------------------------------------------------------------------------
def quat_mul(q, r):
    """Quaternions.__mul__, Quaternions x Quaternions branch (Quaternions.py:91-105): the Hamilton product q r."""
    q, r = np.broadcast_arrays(q, r)
    q0, q1, q2, q3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    r0, r1, r2, r3 = r[..., 0], r[..., 1], r[..., 2], r[..., 3]
    out = np.empty(q.shape)
    out[..., 0] = r0 * q0 - r1 * q1 - r2 * q2 - r3 * q3
    out[..., 1] = r0 * q1 + r1 * q0 - r2 * q3 + r3 * q2
    out[..., 2] = r0 * q2 + r1 * q3 + r2 * q0 - r3 * q1
    out[..., 3] = r0 * q3 - r1 * q2 + r2 * q1 + r3 * q0
    return out


def quat_conj(q):
    """Quaternions.__neg__ (Quaternions.py:137-139)."""
    return q * np.array([1.0, -1.0, -1.0, -1.0])


def quat_rotate(q, v):
    """Quaternions x vectors branch (Quaternions.py:108-111): imaginary part of q (0, v) q*."""
    q, v4 = np.broadcast_arrays(q, np.concatenate([np.zeros(v.shape[:-1] + (1,)), v], axis=-1))
    return quat_mul(q, quat_mul(v4, quat_conj(q)))[..., 1:]


def quat_from_angle_axis(angles, axis):
    """Quaternions.from_angle_axis (Quaternions.py:401-405), including the 1e-10 in the axis normalisation."""
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / (np.sqrt(np.sum(axis ** 2, axis=-1)) + 1e-10)[..., np.newaxis]
    s = np.sin(angles / 2.0)[..., np.newaxis]
    c = np.cos(angles / 2.0)[..., np.newaxis]
    return np.concatenate([c, axis * s], axis=-1)


def quat_from_euler_xyz_world(es):
    """Quaternions.from_euler(es, order='xyz', world=True) (Quaternions.py:408-420): q_z (q_y q_x)."""
    qx = quat_from_angle_axis(es[..., 0], np.array([1.0, 0.0, 0.0]))
    qy = quat_from_angle_axis(es[..., 1], np.array([0.0, 1.0, 0.0]))
    qz = quat_from_angle_axis(es[..., 2], np.array([0.0, 0.0, 1.0]))
    return quat_mul(qz, quat_mul(qy, qx))


def quat_to_euler_xyz(q):
    """Quaternions.euler(order='xyz') (Quaternions.py:215-227), on the normalised quaternion."""
    q = q / np.sqrt(np.sum(q ** 2, axis=-1))[..., np.newaxis]
    q0, q1, q2, q3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    es = np.zeros(q.shape[:-1] + (3,))
    es[..., 0] = np.arctan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2))
    es[..., 1] = np.arcsin((2 * (q0 * q2 - q3 * q1)).clip(-1, 1))
    es[..., 2] = np.arctan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3))
    return es


def quat_to_matrix(q):
    """Quaternions.transforms (Quaternions.py:301-324)."""
    qw, qx, qy, qz = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    x2 = qx + qx; y2 = qy + qy; z2 = qz + qz
    xx = qx * x2; yy = qy * y2; wx = qw * x2
    xy = qx * y2; yz = qy * z2; wy = qw * y2
    xz = qx * z2; zz = qz * z2; wz = qw * z2
    m = np.empty(q.shape[:-1] + (3, 3))
    m[..., 0, 0] = 1.0 - (yy + zz); m[..., 0, 1] = xy - wz; m[..., 0, 2] = xz + wy
    m[..., 1, 0] = xy + wz; m[..., 1, 1] = 1.0 - (xx + zz); m[..., 1, 2] = yz - wx
    m[..., 2, 0] = xz - wy; m[..., 2, 1] = yz + wx; m[..., 2, 2] = 1.0 - (xx + yy)
    return m


def quat_from_matrix(ts):
    """Quaternions.from_transforms (Quaternions.py:423-465): largest-component branch with the reference's sign rules."""
    d0, d1, d2 = ts[..., 0, 0], ts[..., 1, 1], ts[..., 2, 2]
    q0 = np.sqrt(((d0 + d1 + d2 + 1.0) / 4.0).clip(0, None))
    q1 = np.sqrt(((d0 - d1 - d2 + 1.0) / 4.0).clip(0, None))
    q2 = np.sqrt(((-d0 + d1 - d2 + 1.0) / 4.0).clip(0, None))
    q3 = np.sqrt(((-d0 - d1 + d2 + 1.0) / 4.0).clip(0, None))
    c0 = (q0 >= q1) & (q0 >= q2) & (q0 >= q3)
    c1 = (q1 >= q0) & (q1 >= q2) & (q1 >= q3)
    c2 = (q2 >= q0) & (q2 >= q1) & (q2 >= q3)
    c3 = (q3 >= q0) & (q3 >= q1) & (q3 >= q2)
    # the reference applies the four masked updates one after the other (a tie satisfies more than one mask)
    q1[c0] *= np.sign(ts[c0, 2, 1] - ts[c0, 1, 2]); q2[c0] *= np.sign(ts[c0, 0, 2] - ts[c0, 2, 0]); q3[c0] *= np.sign(ts[c0, 1, 0] - ts[c0, 0, 1])
    q0[c1] *= np.sign(ts[c1, 2, 1] - ts[c1, 1, 2]); q2[c1] *= np.sign(ts[c1, 1, 0] + ts[c1, 0, 1]); q3[c1] *= np.sign(ts[c1, 0, 2] + ts[c1, 2, 0])
    q0[c2] *= np.sign(ts[c2, 0, 2] - ts[c2, 2, 0]); q1[c2] *= np.sign(ts[c2, 1, 0] + ts[c2, 0, 1]); q3[c2] *= np.sign(ts[c2, 2, 1] + ts[c2, 1, 2])
    q0[c3] *= np.sign(ts[c3, 1, 0] - ts[c3, 0, 1]); q1[c3] *= np.sign(ts[c3, 2, 0] + ts[c3, 0, 2]); q2[c3] *= np.sign(ts[c3, 2, 1] + ts[c3, 1, 2])
    return np.stack([q0, q1, q2, q3], axis=-1)


# ---- forward kinematics (Animation.py) -----------------------------------------------------------------------------
def transforms_global(rot, pos, parents):
    """Animation.transforms_local / transforms_global (Animation.py:294-323, 379-414): per joint [R(rot) | pos; 0 0 0 1],
    composed parent-first (joint order is topological).  rot: (F, J, 4), pos: (F, J, 3) -> (F, J, 4, 4)."""
    F, J = rot.shape[:2]
    loc = np.zeros((F, J, 4, 4))
    loc[:, :, :3, :3] = quat_to_matrix(rot)
    loc[:, :, :3, 3] = pos
    loc[:, :, 3, 3] = 1.0
    glob = np.zeros((F, J, 4, 4))
    glob[:, 0] = loc[:, 0]
    for i in range(1, J):
        glob[:, i] = np.matmul(glob[:, parents[i]], loc[:, i])
    return glob


def positions_global(rot, pos, parents):
    """Animation.positions_global (Animation.py:416-438)."""
    g = transforms_global(rot, pos, parents)[:, :, :, 3]
    return g[:, :, :3] / g[:, :, 3, np.newaxis]


def descendants_mask(parents):
    """AnimationStructure.descendants_mask (AnimationStructure.py:129-150, 217): mask[i, j] = j is a descendant of i."""
    J = len(parents)
    m = np.zeros((J, J), dtype=bool)
    for j in range(J):
        p = parents[j]
        while p != -1:
            m[p, j] = True
            p = parents[p]
    return m


# ---- the solver (InverseKinematics.py:326-561) ---------------------------------------------------------------------
def _jacobian(x, gp, gr, parents, target_joints, dsc, tdsc, translate):
    """JacobianInverseKinematicsCK.jacobian (InverseKinematics.py:411-449).  x: (F, 3J [+3J]); gp: (F, J, 3) global
    positions; gr: (F, J, 4) global rotations; dsc / tdsc: (3J, T) integer masks.  Returns (F, 3T, 3J [+3J])."""
    F, J = gr.shape[:2]
    T = len(target_joints)
    prs = gr[:, parents].copy()
    prs[:, 0] = np.array([1.0, 0.0, 0.0, 0.0])
    tps = gp[:, target_joints]
    qys = quat_from_angle_axis(x[:, 1:J * 3:3], np.array([[[0.0, 1.0, 0.0]]]))
    qzs = quat_from_angle_axis(x[:, 2:J * 3:3], np.array([[[0.0, 0.0, 1.0]]]))
    es = np.empty((F, J * 3, 3))
    es[:, 0::3] = quat_rotate(quat_mul(quat_mul(prs, qzs), qys), np.array([[[1.0, 0.0, 0.0]]]))
    es[:, 1::3] = quat_rotate(quat_mul(prs, qzs), np.array([[[0.0, 1.0, 0.0]]]))
    es[:, 2::3] = quat_rotate(prs, np.array([[[0.0, 0.0, 1.0]]]))
    j = gp.repeat(3, axis=1)
    j = dsc[np.newaxis, :, :, np.newaxis] * (tps[:, np.newaxis, :] - j[:, :, np.newaxis])
    j = np.cross(es[:, :, np.newaxis, :], j)
    j = np.swapaxes(j.reshape((F, J * 3, T * 3)), 1, 2)
    if translate:
        es = np.empty((F, J * 3, 3))
        es[:, 0::3] = quat_rotate(prs, np.array([[[1.0, 0.0, 0.0]]]))
        es[:, 1::3] = quat_rotate(prs, np.array([[[0.0, 1.0, 0.0]]]))
        es[:, 2::3] = quat_rotate(prs, np.array([[[0.0, 0.0, 1.0]]]))
        jt = tdsc[np.newaxis, :, :, np.newaxis] * es[:, :, np.newaxis, :].repeat(T, axis=2)
        jt = np.swapaxes(jt.reshape((F, J * 3, T * 3)), 1, 2)
        j = np.concatenate([j, jt], axis=-1)
    return j


def ik_ck(rot, pos, parents, target_joints, targets, iterations=30, damping=7.0, smoothness=0.001, translate=True, gamma=1.0, dual=False):
    """JacobianInverseKinematicsCK.__call__ with the arguments of apply_results (no references, no angle limits, unit
    weights).  rot: (F, J, 4) local rotations, pos: (F, J, 3) local positions, targets: (T, F, 3) in the order of
    `target_joints`.  Returns the updated (rot, pos).

    dual=True solves the same step in its dual form, dx = J^T (J J^T + lambda^2 I)^-1 e -- identical in exact arithmetic
    because apply_results uses unit weights, so lambda is the same for every unknown -- a 3T x 3T system (T = number of
    targets, ~13) instead of the reference's 6J x 6J (J ~ 31 joints) LU per frame.  This is the form the HIP path will
    use; tests/test_ik_oracle.py checks it against the reference's vectors as well."""
    rot = np.array(rot, dtype=np.float64); pos = np.array(pos, dtype=np.float64)
    parents = np.asarray(parents); target_joints = np.asarray(target_joints)
    F, J = rot.shape[:2]
    desc = descendants_mask(parents)
    tdesc = np.eye(J) + desc
    first_desc = desc[:, target_joints].repeat(3, axis=0).astype(int)
    first_tdesc = tdesc[:, target_joints].repeat(3, axis=0).astype(int)
    endeff = np.swapaxes(np.asarray(targets, dtype=np.float64), 0, 1)          # (F, T, 3)
    for _ in range(iterations):
        gt = transforms_global(rot, pos, parents)
        gp = gt[:, :, :, 3]
        gp = gp[:, :, :3] / gp[:, :, 3, np.newaxis]
        gr = quat_from_matrix(gt)
        x = quat_to_euler_xyz(rot).reshape(F, -1)
        w = np.ones(J).repeat(3)
        if translate:
            x = np.hstack([x, pos.reshape(F, -1)])
            w = np.hstack([w, np.ones(J).repeat(3)])
        jac = _jacobian(x, gp, gr, parents, target_joints, first_desc, first_tdesc, translate)
        lam = damping * (1.0 / (w + 0.001))
        d = (lam * lam) * np.eye(x.shape[1])
        e = gamma * (endeff.reshape(F, -1) - gp[:, target_joints].reshape(F, -1))
        if dual:
            l2 = float(lam[0] * lam[0])
            dx1 = np.array([jf.T.dot(np.linalg.solve(jf.dot(jf.T) + l2 * np.eye(jf.shape[0]), ef)) for jf, ef in zip(jac, e)])
        else:
            dx1 = np.array([np.linalg.solve(jf.T.dot(jf) + d, jf.T.dot(ef)) for jf, ef in zip(jac, e)])
        xp = np.vstack((x[0], x[:F - 1]))
        xa = np.vstack((x[1:], x[F - 1]))
        dx2 = smoothness * (xp + xa - 2 * x)
        x = x + dx1 + dx2
        rot = quat_from_euler_xyz_world(x[:, :J * 3].reshape((F, J, 3)))
        if translate:
            pos = x[:, J * 3:].reshape((F, J, 3))
    return rot, pos
