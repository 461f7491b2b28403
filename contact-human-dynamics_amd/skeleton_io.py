"""Skeleton animation container, BVH reader / writer and forward kinematics for the stages either side of the physics
hot path (``apply_results`` and ``prepare_input`` both start with ``BVH.load`` and the former
ends in ``BVH.save``).

Host-side Python, as in the reference.  Behaviour follows the reference's motion library, cited per function
(``src/skeleton_fitting/ik/{BVH,Animation,Quaternions}.py``); the code is written from the file format, not from that
library: the reader is a token-stream parser of the HIERARCHY / MOTION grammar, the writer builds the joint tree once and
emits lines depth-first.  Quaternions are plain ``(..., 4)`` arrays ``(w, x, y, z)``.

Pinned by ``tests/golden/apply_golden.npz`` (files written here and read by the reference's ``BVH.load``; files written
by the reference's ``BVH.save`` compared byte for byte).
"""
from dataclasses import dataclass

import numpy as np

_AXIS = {'x': 0, 'y': 1, 'z': 2}
_ROT_CHANNEL = {'Xrotation': 'x', 'Yrotation': 'y', 'Zrotation': 'z'}
_CHANNEL_OF = {v: k for k, v in _ROT_CHANNEL.items()}


# ----------------------------------------------------------------------------------------------------------------------
# quaternion helpers (conventions of Quaternions.py)
# ----------------------------------------------------------------------------------------------------------------------
def quat_mul(a, b):
    """Hamilton product a * b (Quaternions.__mul__, Quaternions.py:91-105)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    aw, av = a[..., :1], a[..., 1:]
    bw, bv = b[..., :1], b[..., 1:]
    return np.concatenate([aw * bw - np.sum(av * bv, axis=-1, keepdims=True), aw * bv + bw * av + np.cross(av, bv)], axis=-1)


def quat_from_angle_axis(angles, axis):
    """Quaternions.from_angle_axis (Quaternions.py:394-399), including its ``+ 1e-10`` in the axis normalisation
    (results are a hair short of unit length; downstream code depends on the exact values)."""
    angles = np.asarray(angles, dtype=np.float64)
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / (np.sqrt(np.sum(axis * axis, axis=-1)) + 1e-10)[..., None]
    half = 0.5 * angles[..., None]
    return np.concatenate([np.cos(half), np.sin(half) * axis], axis=-1)


def quat_from_euler(es, order='xyz', world=False):
    """Quaternions.from_euler (Quaternions.py:401-414): es[..., k] is the angle (radians) about axis order[k];
    world=False composes q0 (q1 q2), world=True composes q2 (q1 q0)."""
    es = np.asarray(es, dtype=np.float64)
    qs = []
    for k in range(3):
        ax = np.zeros(3)
        ax[_AXIS[order[k]]] = 1.0
        qs.append(quat_from_angle_axis(es[..., k], ax))
    return quat_mul(qs[2], quat_mul(qs[1], qs[0])) if world else quat_mul(qs[0], quat_mul(qs[1], qs[2]))


def quat_normalized(q):
    q = np.asarray(q, dtype=np.float64)
    return q / np.sqrt(np.sum(q * q, axis=-1))[..., None]


def quat_to_euler_xyz(q):
    """Quaternions.euler(order='xyz') (Quaternions.py:215-227): angles (x, y, z) with q = qz (qy qx)."""
    w, x, y, z = np.moveaxis(quat_normalized(q), -1, 0)
    return np.stack([np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y)),
                     np.arcsin(np.clip(2 * (w * y - z * x), -1, 1)),
                     np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))], axis=-1)


def quat_angle_axis(q):
    """Quaternions.angle_axis (Quaternions.py:289-298), including the 0.001 stand-in for a zero sine."""
    n = quat_normalized(q)
    s = np.sqrt(1 - n[..., 0] ** 2.0)
    s[s == 0] = 0.001
    return 2.0 * np.arccos(n[..., 0]), n[..., 1:] / s[..., None]


def quat_to_matrix(q):
    """Quaternions.transforms (Quaternions.py:301-324); no normalisation, like the reference."""
    w, x, y, z = np.moveaxis(np.asarray(q, dtype=np.float64), -1, 0)
    x2, y2, z2 = x + x, y + y, z + z
    m = np.empty(w.shape + (3, 3))
    m[..., 0, 0] = 1.0 - (y * y2 + z * z2); m[..., 0, 1] = x * y2 - w * z2; m[..., 0, 2] = x * z2 + w * y2
    m[..., 1, 0] = x * y2 + w * z2; m[..., 1, 1] = 1.0 - (x * x2 + z * z2); m[..., 1, 2] = y * z2 - w * x2
    m[..., 2, 0] = x * z2 - w * y2; m[..., 2, 1] = y * z2 + w * x2; m[..., 2, 2] = 1.0 - (x * x2 + y * y2)
    return m


# ----------------------------------------------------------------------------------------------------------------------
# animation container + forward kinematics
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class Motion:
    """What the reference's ``Animation`` holds (Animation.py:9-60): local joint rotations / translations per frame and
    the rest skeleton.  ``parents[0] == -1``; joints are in file (depth-first) order, so ``parents[j] < j``."""
    rotations: np.ndarray     # F x J x 4 quaternions (w, x, y, z)
    positions: np.ndarray     # F x J x 3 local translations
    orients: np.ndarray       # J x 4 (identity for files read here)
    offsets: np.ndarray       # J x 3
    parents: np.ndarray       # J ints

    @property
    def n_frames(self):
        return self.rotations.shape[0]

    @property
    def n_joints(self):
        return self.rotations.shape[1]

    def copy(self):
        return Motion(self.rotations.copy(), self.positions.copy(), self.orients.copy(), self.offsets.copy(), self.parents.copy())

    def frames(self, start=None, end=None):
        """The ``anim.rotations[start:end]`` / ``anim.positions[start:end]`` slicing of towr_utils.py:784-785 (a copy)."""
        return Motion(self.rotations[start:end].copy(), self.positions[start:end].copy(), self.orients.copy(), self.offsets.copy(), self.parents.copy())


def global_transforms(motion):
    """Global rotation matrices (F x J x 3 x 3) and positions (F x J x 3) of every joint: Animation.transforms_global /
    positions_global (Animation.py:294-323, 379-414) -- child = parent o local, local = (rotation, translation)."""
    Rl = quat_to_matrix(motion.rotations)
    R = np.empty_like(Rl)
    p = np.empty_like(motion.positions, dtype=np.float64)
    for j in range(motion.n_joints):
        a = int(motion.parents[j])
        if a < 0:
            R[:, j] = Rl[:, j]; p[:, j] = motion.positions[:, j]
        else:
            R[:, j] = R[:, a] @ Rl[:, j]
            p[:, j] = p[:, a] + np.einsum('fab,fb->fa', R[:, a], motion.positions[:, j])
    return R, p


def positions_global(motion):
    return global_transforms(motion)[1]


def quat_from_euler_xyz_world(es):
    """`quat_from_euler(es, 'xyz', world=True)` -- qz (qy qx) -- in closed form: the product of the three axis quaternions written out (no general quaternion
    products with their cross products and temporaries: 6 x faster on the 10^5 .. 10^6 joint rotations of a batch of clips).  The axis quaternions carry
    `quat_from_angle_axis`'s 1 / (1 + 1e-10) on their sine, as the composed form does; the two agree to rounding."""
    es = np.asarray(es, dtype=np.float64)
    h = 0.5 * es
    c = np.cos(h); sn = np.sin(h) * (1.0 / (1.0 + 1e-10))
    cx, cy, cz = c[..., 0], c[..., 1], c[..., 2]
    sx, sy, sz = sn[..., 0], sn[..., 1], sn[..., 2]
    pw, px, py, pz = cy * cx, cy * sx, sy * cx, -(sy * sx)          # qy qx
    q = np.empty(es.shape[:-1] + (4,))
    q[..., 0] = cz * pw - sz * pz
    q[..., 1] = cz * px - sz * py
    q[..., 2] = cz * py + sz * px
    q[..., 3] = cz * pz + sz * pw
    return q


def positions_global_fast(rotations, positions, parents):
    """Global joint positions (N x J x 3) of N frames with one hierarchy: `positions_global` with every matrix entry held as its own contiguous length-N
    array (the chain products are then plain elementwise multiply-adds over the frames, without 3 x 3 temporaries or strided slices: ~15 x faster for the
    10^4 .. 10^5 frames of a batch of clips) and without the rotations of joints that have no children.  Same sums of products per entry as
    `global_transforms` (the order of the three terms of a dot product is the same), so the results agree to rounding."""
    rotations = np.asarray(rotations, dtype=np.float64); positions = np.asarray(positions, dtype=np.float64)
    N, J = rotations.shape[:2]
    parents = [int(a) for a in parents]
    has_child = [False] * J
    for a in parents:
        if a >= 0:
            has_child[a] = True
    q = np.ascontiguousarray(np.moveaxis(rotations, (1, 2), (1, 0)))          # 4 x J x N
    w, x, y, z = q[0], q[1], q[2], q[3]
    x2, y2, z2 = x + x, y + y, z + z
    # local rotation matrices, entry (r, c) as a J x N array (quat_to_matrix's formulas)
    L = [[1.0 - (y * y2 + z * z2), x * y2 - w * z2, x * z2 + w * y2],
         [x * y2 + w * z2, 1.0 - (x * x2 + z * z2), y * z2 - w * x2],
         [x * z2 - w * y2, y * z2 + w * x2, 1.0 - (x * x2 + y * y2)]]
    pl = np.ascontiguousarray(np.moveaxis(positions, (1, 2), (1, 0)))          # 3 x J x N
    R = [None] * J
    p = np.empty((3, J, N))
    for j in range(J):
        a = parents[j]
        if a < 0:
            R[j] = [[L[r][c][j] for c in range(3)] for r in range(3)]
            for r in range(3):
                p[r, j] = pl[r, j]
            continue
        Ra = R[a]
        if has_child[j]:
            R[j] = [[Ra[r][0] * L[0][c][j] + Ra[r][1] * L[1][c][j] + Ra[r][2] * L[2][c][j] for c in range(3)] for r in range(3)]
        for r in range(3):
            p[r, j] = p[r, a] + (Ra[r][0] * pl[0, j] + Ra[r][1] * pl[1, j] + Ra[r][2] * pl[2, j])
    return np.ascontiguousarray(np.moveaxis(p, 0, 2).swapaxes(0, 1))


# ----------------------------------------------------------------------------------------------------------------------
# BVH
# ----------------------------------------------------------------------------------------------------------------------
def load_bvh(path):
    """Reads a BVH file -> (Motion, joint names, frame time).  Semantics of BVH.load (BVH.py:25-168) for the files this
    pipeline handles (End Sites are not joints; rotation order from the first CHANNELS line; Euler angles in degrees,
    composed in local order; all joints are taken to have as many channels as the LAST joint declares: 3 = only the
    root carries a translation, 6 = every joint does; translations default to the joint offsets)."""
    with open(path) as f:
        text = f.read()
    cut = text.find('MOTION')
    if cut < 0:
        raise ValueError(path + ': no MOTION section')
    tok = text[:cut].split()
    names, offsets, parents = [], [], []
    order, channels = None, None
    stack, active, i, in_end_site = [], -1, 0, False
    pending = None                                           # joint declared, its '{' not yet seen
    while i < len(tok):
        t = tok[i]
        if t in ('ROOT', 'JOINT'):
            names.append(tok[i + 1]); offsets.append([0.0, 0.0, 0.0]); parents.append(active)
            pending = len(names) - 1
            i += 2
        elif t == 'End':                                     # "End Site"
            pending = 'end'
            i += 2
        elif t == '{':
            stack.append(active)
            if pending == 'end':
                in_end_site = True
            else:
                active = pending
            pending = None
            i += 1
        elif t == '}':
            prev = stack.pop()
            if in_end_site:
                in_end_site = False
            else:
                active = prev
            i += 1
        elif t == 'OFFSET':
            if not in_end_site:
                offsets[active] = [float(tok[i + 1]), float(tok[i + 2]), float(tok[i + 3])]
            i += 4
        elif t == 'CHANNELS':
            n = int(tok[i + 1])
            parts = tok[i + 2:i + 2 + n]
            channels = n
            if order is None:
                rot = parts[0:3] if n == 3 else parts[3:6]
                if all(p in _ROT_CHANNEL for p in rot):
                    order = ''.join(_ROT_CHANNEL[p] for p in rot)
            i += 2 + n
        else:
            i += 1                                           # HIERARCHY
    if not names or order is None:
        raise ValueError(path + ': no joints / rotation channels found')
    J = len(names)
    offsets = np.array(offsets, dtype=np.float64)
    parents = np.array(parents, dtype=int)
    mt = text[cut:].split()
    nf = int(mt[mt.index('Frames:') + 1])
    k = mt.index('Time:')
    frametime = float(mt[k + 1])
    data = np.array(mt[k + 2:], dtype=np.float64)
    per = {3: 3 + 3 * J, 6: 6 * J, 9: 3 + 9 * (J - 1)}.get(channels)
    if per is None:
        raise ValueError('%s: %d channels per joint are not supported' % (path, channels))
    if data.size < nf * per:
        raise ValueError('%s: expected %d motion values (%d frames x %d), found %d' % (path, nf * per, nf, per, data.size))
    data = data[:nf * per].reshape(nf, per)
    positions = np.repeat(offsets[None], nf, axis=0)
    eul = np.zeros((nf, J, 3))
    if channels == 3:
        positions[:, 0] = data[:, 0:3]
        eul[:] = data[:, 3:].reshape(nf, J, 3)
    elif channels == 6:
        d = data.reshape(nf, J, 6)
        positions[:] = d[..., 0:3]; eul[:] = d[..., 3:6]
    else:                                                    # 9: translation, rotation, scale per non-root joint (BVH.py:156-160)
        positions[:, 0] = data[:, 0:3]
        d = data[:, 3:].reshape(nf, J - 1, 9)
        eul[:, 1:] = d[..., 3:6]
        positions[:, 1:] += d[..., 0:3] * d[..., 6:9]
    rotations = quat_from_euler(np.radians(eul), order=order, world=False)
    orients = np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (J, 1))
    return Motion(rotations, positions, orients, offsets, parents), names, frametime


def save_bvh(path, motion, names=None, frametime=1.0 / 24.0, order='zyx'):
    """Writes the file BVH.save(filename, anim, names) produces (BVH.py:172-253 with its defaults -- the call of
    towr_utils.py:975 -- i.e. frame time 1/24 unless given, 'zyx' channels, translation channels on the root only, an
    End Site under every leaf, '%f' numbers, the trailing blanks of its 6-channel and motion lines)."""
    if order != 'zyx':
        raise NotImplementedError("only the reference's default channel order 'zyx' (Euler extraction 'xyz') is supported")
    J = motion.n_joints
    if names is None:
        names = ['joint_%d' % i for i in range(J)]
    children = [[] for _ in range(J)]
    for j in range(1, J):
        children[int(motion.parents[j])].append(j)
    rot_channels = ' '.join(_CHANNEL_OF[c] for c in order)
    out = ['HIERARCHY']

    def emit(j, depth):
        t = '\t' * depth
        out.append('%s%s %s' % (t, 'ROOT' if j == 0 else 'JOINT', names[j]))
        out.append(t + '{')
        t1 = t + '\t'
        out.append('%sOFFSET %f %f %f' % ((t1,) + tuple(motion.offsets[j])))
        out.append(t1 + ('CHANNELS 6 Xposition Yposition Zposition %s ' % rot_channels if j == 0 else 'CHANNELS 3 %s' % rot_channels))
        for c in children[j]:
            emit(c, depth + 1)
        if not children[j] and j != 0:
            out.extend([t1 + 'End Site', t1 + '{', '%s\tOFFSET %f %f %f' % (t1, 0.0, 0.0, 0.0), t1 + '}'])
        out.append(t + '}')

    emit(0, 0)
    out.append('MOTION')
    out.append('Frames: %i' % motion.n_frames)
    out.append('Frame Time: %f' % frametime)
    deg = np.degrees(quat_to_euler_xyz(motion.rotations))[..., [_AXIS[c] for c in order]]
    rows = np.concatenate([np.asarray(motion.positions)[:, 0], deg.reshape(motion.n_frames, -1)], axis=1)
    fmt = '%f ' * rows.shape[1]                   # one formatting call per frame (the same '%f' per number: 6 x faster than a call per number)
    out.extend(fmt % tuple(r) for r in rows.tolist())
    with open(path, 'w') as fh:
        fh.write('\n'.join(out) + '\n')
