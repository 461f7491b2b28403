"""Producer of the physics stage's four input files: SURVEY 8(f) rank 2, the reference's ``prepare_input``
(``src/utils/towr_utils.py:451-777``), without its motion library (which no longer imports under NumPy 2).

    prepare_input(anim_bvh, floor_file, contacts_file, out_dir, character, start_idx, end_idx, dt)

reads a BVH, the fitted floor (`floor_out.txt`: normal, point) and the per-frame foot contacts (`foot_contacts.npy`,
columns left heel, left toe, right heel, right toe) and writes ``skel_info.txt``, ``motion_info.txt``,
``terrain_info.txt`` and ``contact_info.txt`` -- or, with `prepare_sequence`, hands the same data to the solver in
memory (`io_formats.SeqInput`), skipping the files.  Per video this is small array work (NumPy on the host, as in the reference);
for a run of thousands of videos (BASELINE configs[2]) `prepare_sequences_device` does the per-frame numerics -- two forward-kinematics
passes, centre of mass, inertia, hip offsets, toe / heel trajectories -- for the frames of ALL clips in one launch of a hand-written HIP kernel
(libchd_prepare.so, include/chd_prepare.h; the float64 tensor-operation form of rounds 2-4 stays as a cross-check), reads the BVH files with the
library's native parser (`prepare_capi.load_bvh_batch`) and leaves the sequential pieces (root-angle unwrapping, contact run lengths) on the host.

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


def unwrap_batch(root_rots):
    """`unwrap_like_reference` for a list of (F_i, 3) arrays at once: the walk along the frames stays sequential, the clips and the three
    angles go through it together (same additions in the same order per entry)."""
    B = len(root_rots)
    Fm = max(r.shape[0] for r in root_rots)
    out = np.empty((B, Fm, 3))
    for b, r in enumerate(root_rots):
        out[b, :r.shape[0]] = r; out[b, r.shape[0]:] = r[-1]
    cur = out[:, 0].copy()
    for f in range(1, Fm):
        step = np.where(cur >= 0.0, 2 * np.pi, -2 * np.pi)
        nxt = out[:, f].copy()
        for _ in range(4):
            far = np.abs(nxt - cur) > np.pi
            if not far.any():
                break
            nxt = np.where(far, nxt + step, nxt)
        if (np.abs(nxt - cur) > np.pi).any():
            b = int(np.argwhere(np.abs(nxt - cur) > np.pi)[0][0])
            raise ValueError('clip %d: root orientation jumps away from its previous value at frame %d (the reference loops forever here)' % (b, f))
        out[:, f] = nxt
        cur = nxt
    return [out[b, :r.shape[0]] for b, r in enumerate(root_rots)]


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


def _fk_device(rot, pos, parents):
    """Animation.transforms_global on tensors: rot (B, F, J, 4), pos (B, F, J, 3) -> global positions (B, F, J, 3).  Rotation matrices from
    the quaternions as Quaternions.transforms does it (no normalisation), child = parent o local."""
    import torch
    w, x, y, z = rot.unbind(-1)
    x2, y2, z2 = x + x, y + y, z + z
    Rl = torch.stack([1.0 - (y * y2 + z * z2), x * y2 - w * z2, x * z2 + w * y2,
                      x * y2 + w * z2, 1.0 - (x * x2 + z * z2), y * z2 - w * x2,
                      x * z2 - w * y2, y * z2 + w * x2, 1.0 - (x * x2 + y * y2)], dim=-1).reshape(rot.shape[:-1] + (3, 3))
    R, P = [None] * len(parents), [None] * len(parents)
    for j, a in enumerate(parents):
        if a < 0:
            R[j] = Rl[:, :, j]; P[j] = pos[:, :, j]
        else:
            R[j] = R[a] @ Rl[:, :, j]
            P[j] = P[a] + (R[a] @ pos[:, :, j, :, None])[..., 0]
    return torch.stack(P, dim=2)


def prepare_sequences_device(motions, floors, foot_contacts, character: Character, starts=None, ends=None, dt=1.0 / 30.0, combined_contacts=False,
                             device='cuda', backend=None, frames_fn=None):
    """`prepare_sequence` for a list of clips of one character (same hierarchy) with the per-frame numerics batched on `device`.
    Equal to the NumPy path to rounding (tests/test_prepare_input.py; tests/test_config4_gpu.py on the MI355X).

    backend 'hip' (the default on a cuda device): ONE launch of the hand-written kernel of libchd_prepare.so over the frames of all clips (include/chd_prepare.h,
    csrc/chd_prepare_kernels.hpp); it raises when the library or a HIP device is missing -- no silent fallback.  backend 'torch': the same numerics as float64
    tensor operations (rounds 2-4; kept as an independent cross-check and for the CPU device of torch in tests).  `frames_fn` (tests): a stand-in for
    `prepare_capi.prep_frames` -- the host emulation of the kernel source."""
    if backend is None:
        backend = 'hip' if (frames_fn is not None or str(device).startswith('cuda')) else 'torch'
    if backend == 'hip':
        return _prepare_sequences_hip(motions, floors, foot_contacts, character, starts, ends, dt, combined_contacts, device, frames_fn)
    import torch
    for need in ('left_leg_chain', 'hip_inds', 'mass'):
        if getattr(character, need) is None:
            raise ValueError('Character.%s is required by prepare_input' % need)
    B = len(motions)
    starts = [0 if s is None else s for s in (starts or [None] * B)]
    ends = [m.n_frames if e is None else e for m, e in zip(motions, ends or [None] * B)]
    anims = [m.copy() if character.heel_inds is not None else add_heels(m, character.toe_inds, character.ankle_inds) for m in motions]
    J0, J1 = motions[0].n_joints, anims[0].n_joints
    parents0, parents1 = [int(a) for a in motions[0].parents], [int(a) for a in anims[0].parents]
    if any(m.n_joints != J0 or [int(a) for a in m.parents] != parents0 for m in motions):
        raise ValueError('prepare_sequences_device: the clips of a batch share one skeleton hierarchy')
    Fm = max(m.n_frames for m in motions)
    dev = torch.device(device)

    def pad(arrs, tail):                        # (F_i, ...) -> (B, Fm, ...) repeating the last frame
        out = np.empty((B, Fm) + tail)
        for b, a in enumerate(arrs):
            out[b, :a.shape[0]] = a; out[b, a.shape[0]:] = a[-1]
        return torch.from_numpy(out).to(dev)

    rot1 = pad([a.rotations for a in anims], (J1, 4)); pos1 = pad([a.positions for a in anims], (J1, 3))
    swap = torch.tensor(_SWAP, device=dev)
    seg = [(character.seg_to_mass_perc[k] * 0.01, torch.tensor(list(j), device=dev)) for k, j in character.seg_to_joints.items()]

    def com_of(gp):
        return sum(fr * gp[:, :, jj].mean(dim=2) for fr, jj in seg)

    # pass 1 (towr_utils.py:483-535): root rotation and translation zeroed -> hip offsets from the COM, inertia about it
    rot0 = rot1[:, :, :J0].clone(); pos0 = pos1[:, :, :J0].clone()
    rot0[:, :, 0] = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float64, device=dev); pos0[:, :, 0] = 0.0
    gp = _fk_device(rot0, pos0, parents0)
    com = com_of(gp)
    hip = -0.01 * (gp[:, :, list(character.hip_inds)] - com[:, :, None])[..., swap]
    pos0[:, :, 0] = pos0[:, :, 0] - com
    rel = -0.01 * _fk_device(rot0, pos0, parents0)[..., swap]
    eye = torch.eye(3, dtype=torch.float64, device=dev)
    inertia = torch.zeros((B, Fm, 3, 3), dtype=torch.float64, device=dev)
    for (fr, jj) in seg:
        r = rel[:, :, jj].mean(dim=2)
        inertia = inertia + fr * character.mass * ((r * r).sum(dim=-1)[..., None, None] * eye - r[..., :, None] * r[..., None, :])
    # pass 2 (towr_utils.py:542-655): the animation as it is (heels appended) -> COM, toe / heel trajectories
    pos = -0.01 * _fk_device(rot1, pos1, parents1)[..., swap]
    lh, rh = (J1 - 2, J1 - 1) if character.heel_inds is None else character.heel_inds
    com_traj = com_of(pos)
    feet = pos[:, :, [character.toe_inds[0], lh, character.toe_inds[1], rh]]
    hd = (feet[:, :, 0] - feet[:, :, 1]).norm(dim=-1)
    hip, inertia, com_traj, feet, hd = [t.cpu().numpy() for t in (hip, inertia, com_traj, feet, hd)]
    chain = list(character.left_leg_chain)
    out = []
    raw = []
    for a in anims:
        angle, axis = sk.quat_angle_axis(a.rotations[:, 0])
        raw.append(sk.quat_to_euler_xyz(sk.quat_from_angle_axis(angle, -axis[:, _SWAP])))
    root_rots = unwrap_batch(raw)
    for b, (m, a) in enumerate(zip(motions, anims)):
        F = m.n_frames
        sl = slice(starts[b], ends[b])
        root_rot = root_rots[b]
        start, durations = contact_schedule(foot_contacts[b], starts[b], ends[b], dt, combined_contacts)
        I = inertia[b, :F][sl]
        out.append(iof.SeqInput(F=ends[b] - starts[b], dt=dt, hip_l=hip[b, :F][sl, 0], hip_r=hip[b, :F][sl, 1],
                                leg_len=float(np.sum(np.linalg.norm(m.offsets[chain[1:]], axis=1)) * 0.01),
                                heel_len=float((np.sum(np.linalg.norm(a.offsets[chain[1:-1]], axis=1)) + np.linalg.norm(a.offsets[lh])) * 0.01),
                                heel_dist=float(np.mean(hd[b, :F])), mass=float(character.mass),
                                inertia=np.stack([I[:, 0, 0], I[:, 1, 1], I[:, 2, 2], I[:, 0, 1], I[:, 0, 2], I[:, 1, 2]], axis=1),
                                com=com_traj[b, :F][sl], euler=root_rot[sl], ltoe=feet[b, :F][sl, 0], lheel=feet[b, :F][sl, 1], rtoe=feet[b, :F][sl, 2],
                                rheel=feet[b, :F][sl, 3], normal=np.asarray(floors[b][0], dtype=np.float64), point=np.asarray(floors[b][1], dtype=np.float64),
                                start_contact=start, durations=durations))
    return out


def _prepare_sequences_hip(motions, floors, foot_contacts, character, starts, ends, dt, combined_contacts, device, frames_fn):
    from . import prepare_capi as pc
    for need in ('left_leg_chain', 'hip_inds', 'mass'):
        if getattr(character, need) is None:
            raise ValueError('Character.%s is required by prepare_input' % need)
    B = len(motions)
    starts = [0 if s is None else s for s in (starts or [None] * B)]
    ends = [m.n_frames if e is None else e for m, e in zip(motions, ends or [None] * B)]
    anims = [m.copy() if character.heel_inds is not None else add_heels(m, character.toe_inds, character.ankle_inds) for m in motions]
    J0 = motions[0].n_joints
    parents0, parents1 = [int(a) for a in motions[0].parents], [int(a) for a in anims[0].parents]
    if any(m.n_joints != J0 or [int(a) for a in m.parents] != parents0 for m in motions):
        raise ValueError('prepare_sequences_device: the clips of a batch share one skeleton hierarchy')
    skel = pc.skeleton_of(character, parents1, J0)
    rot = np.concatenate([a.rotations for a in anims], axis=0); pos = np.concatenate([a.positions for a in anims], axis=0)      # the frames of all clips, one after the other
    if frames_fn is not None:
        fr = frames_fn(skel, rot, pos)
    else:
        dev = str(device)
        fr = pc.prep_frames(skel, rot, pos, device=int(dev.split(':')[1]) if ':' in dev else 0)
    lh = skel.heel_inds[0]
    chain = list(character.left_leg_chain)
    raw = []
    for a in anims:
        angle, axis = sk.quat_angle_axis(a.rotations[:, 0])
        raw.append(sk.quat_to_euler_xyz(sk.quat_from_angle_axis(angle, -axis[:, _SWAP])))
    root_rots = unwrap_batch(raw)
    out, f0 = [], 0
    for b, (m, a) in enumerate(zip(motions, anims)):
        F = m.n_frames
        o = fr[f0:f0 + F]; f0 += F
        sl = slice(starts[b], ends[b])
        start, durations = contact_schedule(foot_contacts[b], starts[b], ends[b], dt, combined_contacts)
        out.append(iof.SeqInput(F=ends[b] - starts[b], dt=dt, hip_l=o[sl, 0:3], hip_r=o[sl, 3:6],
                                leg_len=float(np.sum(np.linalg.norm(m.offsets[chain[1:]], axis=1)) * 0.01),
                                heel_len=float((np.sum(np.linalg.norm(a.offsets[chain[1:-1]], axis=1)) + np.linalg.norm(a.offsets[lh])) * 0.01),
                                heel_dist=float(np.mean(o[:, 27])), mass=float(character.mass), inertia=o[sl, 6:12],
                                com=o[sl, 12:15], euler=root_rots[b][sl], ltoe=o[sl, 15:18], lheel=o[sl, 18:21], rtoe=o[sl, 21:24], rheel=o[sl, 24:27],
                                normal=np.asarray(floors[b][0], dtype=np.float64), point=np.asarray(floors[b][1], dtype=np.float64),
                                start_contact=start, durations=durations))
    return out


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
