"""Producer of the physics stage's four input files: SURVEY 8(f) rank 2, the reference's ``prepare_input``
(``src/utils/towr_utils.py:451-777``), without its motion library (which no longer imports under NumPy 2).

    prepare_input(anim_bvh, floor_file, contacts_file, out_dir, character, start_idx, end_idx, dt)

reads a BVH, the fitted floor (`floor_out.txt`: normal, point) and the per-frame foot contacts (`foot_contacts.npy`,
columns left heel, left toe, right heel, right toe) and writes ``skel_info.txt``, ``motion_info.txt``,
``terrain_info.txt`` and ``contact_info.txt`` -- or, with `prepare_sequence`, hands the same data to the solver in
memory (`io_formats.SeqInput`), skipping the files.  Small per-video array work: NumPy on the host, as in the reference.

Coordinates: the animation is y-up in centimetres; the solver is z-up in metres with x, y negated
(``p_solver = -0.01 * p[[x, z, y]]``, towr_utils.py:519-521, 568-571).
"""
import os

import numpy as np

from . import io_formats as iof
from . import skeleton_io as sk
from .apply_results import Character, add_heels, centre_of_mass

_SWAP = [0, 2, 1]


def _to_solver(p):
    return -0.01 * np.asarray(p, dtype=np.float64)[..., _SWAP]


def unwrap_like_reference(root_rot):
    """The reference's smoothing of the root Euler angles (towr_utils.py:620-629): when consecutive values differ by
    more than pi, 2 pi is added (previous value >= 0) or subtracted (previous value < 0) until they do not.  A jump in the
    other direction never terminates in the reference; here it raises."""
    out = np.array(root_rot, dtype=np.float64)
    for dim in range(3):
        cur = out[0, dim]
        for f in range(1, out.shape[0]):
            step = 2 * np.pi if cur >= 0.0 else -2 * np.pi
            nxt = out[f, dim]
            n = 0
            while abs(nxt - cur) > np.pi:
                nxt += step
                n += 1
                if n > 4:
                    raise ValueError('root orientation jumps away from its previous value at frame %d (the reference loops forever here)' % f)
            out[f, dim] = nxt
            cur = nxt
    return out


def contact_schedule(foot_contacts, start_idx, end_idx, dt, combined_contacts=False):
    """towr_utils.py:695-725: start flags and phase durations in file order (left toe, left heel, right toe, right
    heel).  As in the reference, the toes' START flag is taken from the foot's combined (heel or toe) signal unless
    `combined_contacts`, in which case it is the toe's own -- the opposite of which signal the toes' durations use."""
    fc = np.asarray(foot_contacts)
    left = np.amax(fc[:, [0, 1]], axis=1)[start_idx:end_idx]
    right = np.amax(fc[:, [2, 3]], axis=1)[start_idx:end_idx]
    each = fc[start_idx:end_idx][:, [1, 0, 3, 2]]
    start = [each[0, 0] if combined_contacts else left[0], each[0, 1], each[0, 2] if combined_contacts else right[0], each[0, 3]]
    signals = [left if combined_contacts else each[:, 0], each[:, 1], right if combined_contacts else each[:, 2], each[:, 3]]
    return [int(v) for v in start], [iof.contact_durations(sig, dt) for sig in signals]


def read_floor(floor_file):
    """floor_out.txt -> (normal, point) in the solver's frame (towr_utils.py:675-689): the point is in centimetres."""
    with open(floor_file) as f:
        normal = np.array([float(x) for x in f.readline().split()])
        point = np.array([float(x) for x in f.readline().split()]) * 0.01
    return -normal[_SWAP], -point[_SWAP]


def prepare_sequence(motion: sk.Motion, floor, foot_contacts, character: Character, start_idx=None, end_idx=None, dt=1.0 / 30.0,
                     combined_contacts=False) -> iof.SeqInput:
    """Everything `prepare_input` computes, as the in-memory input of the solver.  `floor` = (normal, point) in the
    solver's frame (see `read_floor`)."""
    for need in ('left_leg_chain', 'hip_inds', 'mass'):
        if getattr(character, need) is None:
            raise ValueError('Character.%s is required by prepare_input' % need)
    start_idx = 0 if start_idx is None else start_idx
    end_idx = motion.n_frames if end_idx is None else end_idx
    chain = list(character.left_leg_chain)
    # --- pass 1 (towr_utils.py:483-535): root rotation and translation zeroed -> hip offsets from the COM and the inertia about it
    body = motion.copy()
    body.rotations[:, 0] = sk.quat_from_euler(np.zeros((body.n_frames, 3)), order='xyz', world=True)
    body.positions[:, 0] = 0.0
    gp = sk.positions_global(body)
    com = centre_of_mass(gp, character)
    hip_offsets = _to_solver(gp[:, list(character.hip_inds), :] - com[:, None, :])
    leg_len = float(np.sum(np.linalg.norm(motion.offsets[chain[1:]], axis=1)) * 0.01)
    body.positions[:, 0] -= com
    rel = _to_solver(sk.positions_global(body))
    inertia = np.zeros((body.n_frames, 3, 3))
    for key, joints in character.seg_to_joints.items():
        r = np.mean(rel[:, list(joints), :], axis=1)
        m = character.seg_to_mass_perc[key] * 0.01 * character.mass
        inertia += m * (np.einsum('f,ab->fab', np.sum(r * r, axis=1), np.eye(3)) - np.einsum('fa,fb->fab', r, r))
    # --- pass 2 (towr_utils.py:542-655): the animation as it is, heels appended -> COM, root orientation, toe / heel trajectories
    anim = motion.copy() if character.heel_inds is not None else add_heels(motion, character.toe_inds, character.ankle_inds)
    pos = _to_solver(sk.positions_global(anim))
    lh, rh = (anim.n_joints - 2, anim.n_joints - 1) if character.heel_inds is None else character.heel_inds
    ltoe, rtoe = pos[:, character.toe_inds[0]], pos[:, character.toe_inds[1]]
    lheel, rheel = pos[:, lh], pos[:, rh]
    heel_dist = float(np.mean(np.linalg.norm(ltoe - lheel, axis=1)))
    heel_len = float((np.sum(np.linalg.norm(anim.offsets[chain[1:-1]], axis=1)) + np.linalg.norm(anim.offsets[lh])) * 0.01)
    angle, axis = sk.quat_angle_axis(anim.rotations[:, 0])
    root_rot = unwrap_like_reference(sk.quat_to_euler_xyz(sk.quat_from_angle_axis(angle, -axis[:, _SWAP])))
    com_traj = centre_of_mass(pos, character)
    start, durations = contact_schedule(foot_contacts, start_idx, end_idx, dt, combined_contacts)
    sl = slice(start_idx, end_idx)
    I = inertia[sl]
    return iof.SeqInput(F=end_idx - start_idx, dt=dt, hip_l=hip_offsets[sl, 0], hip_r=hip_offsets[sl, 1], leg_len=leg_len, heel_len=heel_len,
                        heel_dist=heel_dist, mass=float(character.mass),
                        inertia=np.stack([I[:, 0, 0], I[:, 1, 1], I[:, 2, 2], I[:, 0, 1], I[:, 0, 2], I[:, 1, 2]], axis=1),
                        com=com_traj[sl], euler=root_rot[sl], ltoe=ltoe[sl], lheel=lheel[sl], rtoe=rtoe[sl], rheel=rheel[sl],
                        normal=np.asarray(floor[0], dtype=np.float64), point=np.asarray(floor[1], dtype=np.float64),
                        start_contact=start, durations=durations)


def prepare_input(anim_bvh, floor_file, contacts_file, out_dir, character: Character, start_idx=None, end_idx=None, dt=1.0 / 30.0,
                  combined_contacts=False) -> iof.SeqInput:
    """Same arguments and files as the reference's `prepare_input`; missing inputs raise instead of printing."""
    for p in (anim_bvh, floor_file, contacts_file):
        if not os.path.exists(p):
            raise FileNotFoundError(p)
    motion, _, _ = sk.load_bvh(anim_bvh)
    seq = prepare_sequence(motion, read_floor(floor_file), np.load(contacts_file), character, start_idx, end_idx, dt, combined_contacts)
    iof.write_inputs(seq, out_dir)
    return seq
