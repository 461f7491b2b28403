"""Back-projection of the physics stage's output onto the full skeleton: the host side of SURVEY 8(f) rank 1.

Mirrors, for a whole batch of videos, what the reference does per video when it turns ``sol_out_*.txt`` into a BVH::

    res = load_results(data_path, flip_coords=True)                                    # towr_utils.py:51-121
    anim, names, anim_og, com_og = apply_results(res, anim_bvh, start, end, character)  # towr_utils.py:779-857
    save_anim = remove_heel_from_anim(anim)                                             # towr_utils.py:973-974
    BVH.save(out_bvh, save_anim, names)                                                 # towr_utils.py:975

Everything up to the solver call is small per-video array work and stays in NumPy, as in the reference; the solver call
itself (30 damped-least-squares iterations over every frame, the whole cost of this step) is ONE batched launch sequence
of the HIP library for all videos (`ik_backproject.IkBackProject`, include/chd_ik.h).  There is no CPU solver here:
`back_project` needs an `IkBackProject` (or, in tests, an object with the same `solve` method).

The reference looks the joint indices and segment tables of a character up by name
(src/utils/character_info_utils.py); here they are passed in as a `Character` value, so no table is baked in.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import io_formats as iof
from . import skeleton_io as sk


@dataclass
class Character:
    """The per-character look-ups apply_results makes (character_info_utils.py getters named in towr_utils.py:25-27)."""
    toe_inds: Sequence[int]                      # get_character_toe_inds: [left, right]
    ankle_inds: Sequence[int]                    # get_character_ankle_inds: [left, right]
    upper_body: Sequence[int]                    # get_character_upper_body (the root joint first)
    seg_to_joints: Dict[str, List[int]]          # get_character_seg_to_joint_map
    seg_to_mass_perc: Dict[str, float]           # get_character_seg_to_mass_perc_map (percent)
    heel_inds: Optional[Sequence[int]] = None    # get_character_heel_inds; None = not in `heeled_characters`: heels get appended
    # only prepare_input needs these (prepare_input.py)
    left_leg_chain: Optional[Sequence[int]] = None   # get_character_leg_chain(character, 'left'): hip ... ankle, toe
    hip_inds: Optional[Sequence[int]] = None         # get_character_hip_inds: [left, right]
    mass: Optional[float] = None                     # get_character_mass (kg)

    @staticmethod
    def from_json(path):
        """A JSON object with the field names above (how a deployment carries its character tables)."""
        import json
        with open(path) as f:
            d = json.load(f)
        unknown = set(d) - set(Character.__dataclass_fields__)
        if unknown:
            raise ValueError('%s: unknown character fields %s' % (path, sorted(unknown)))
        return Character(**d)


@dataclass
class TowrResults:
    """`TowrResults` of towr_utils.py:29-49, in the animation's frame (y up, after load_results' swaps)."""
    dt: float
    num_feet: int
    base_pos: np.ndarray        # F x 3 (metres)
    base_rot: np.ndarray        # F x 3 Euler angles 'xyz' in RADIANS (what Quaternions.euler returns, towr_utils.py:118)
    base_R: np.ndarray          # F x 3 x 3
    feet_pos: np.ndarray        # F x nFeet x 3
    feet_force: np.ndarray      # F x nFeet x 3
    feet_contact: np.ndarray    # F x nFeet


def to_animation_frame(sol: iof.Solution, flip_coords=True) -> TowrResults:
    """The second half of load_results (towr_utils.py:101-119): swap y and z (z is up in the solver, y in the
    animation), negate when `flip_coords`, and carry the base orientation over through its angle-axis form."""
    swap = [0, 2, 1]
    sgn = -1.0 if flip_coords else 1.0
    base_pos = sgn * np.asarray(sol.base_lin, dtype=np.float64)[:, swap]
    feet_pos = sgn * np.transpose(np.asarray(sol.ee_pos, dtype=np.float64), (1, 0, 2))[:, :, swap]
    feet_force = sgn * np.transpose(np.asarray(sol.ee_force, dtype=np.float64), (1, 0, 2))[:, :, swap]
    q = sk.quat_from_euler(np.radians(np.asarray(sol.base_ang_deg, dtype=np.float64)), order='xyz', world=True)
    angle, axis = sk.quat_angle_axis(q)
    q2 = sk.quat_from_angle_axis(angle, sgn * axis[:, swap])
    return TowrResults(dt=sol.dt, num_feet=int(np.asarray(sol.ee_pos).shape[0]), base_pos=base_pos, base_rot=sk.quat_to_euler_xyz(q2),
                       base_R=sk.quat_to_matrix(q2), feet_pos=feet_pos, feet_force=feet_force, feet_contact=np.asarray(sol.contact).T.copy())


def load_towr_results(path, flip_coords=True) -> TowrResults:
    """load_results(file_path, flip_coords) (towr_utils.py:51-121)."""
    return to_animation_frame(iof.load_results(path), flip_coords)


def add_heels(motion: sk.Motion, toe_inds, ankle_inds) -> sk.Motion:
    """add_heel_to_anim (towr_utils.py:401-423): two extra joints (left, right heel) as the last joints, children of the
    ankles, at the vertical offset of the toes, identity rotation."""
    m = motion.copy()
    F = m.n_frames
    heel_off = np.zeros((2, 3))
    heel_off[:, 1] = m.offsets[list(toe_inds), 1]
    ident = np.array([1.0, 0.0, 0.0, 0.0])
    m.offsets = np.concatenate([m.offsets, heel_off], axis=0)
    m.parents = np.concatenate([m.parents, np.asarray(ankle_inds, dtype=m.parents.dtype)])
    m.positions = np.concatenate([m.positions, np.repeat(heel_off[None], F, axis=0)], axis=1)
    m.orients = np.concatenate([m.orients, np.tile(ident, (2, 1))], axis=0)
    m.rotations = np.concatenate([m.rotations, np.tile(ident, (F, 2, 1))], axis=1)
    return m


def remove_heels(motion: sk.Motion) -> sk.Motion:
    """remove_heel_from_anim (towr_utils.py:425-433)."""
    n = motion.n_joints - 2
    return sk.Motion(motion.rotations[:, :n].copy(), motion.positions[:, :n].copy(), motion.orients[:n].copy(), motion.offsets[:n].copy(), motion.parents[:n].copy())


def centre_of_mass(gpos, character: Character):
    """COM per frame from the segment tables (towr_utils.py:803-810): mass fraction x mean of the segment's joints."""
    com = np.zeros((gpos.shape[0], 3))
    for key, joints in character.seg_to_joints.items():
        com += character.seg_to_mass_perc[key] * 0.01 * np.mean(gpos[:, list(joints), :], axis=1)
    return com


@dataclass
class BackProjectionTask:
    """One video between `prepare` and `finish`."""
    motion: sk.Motion                       # animation handed to the solver (root replaced by the optimised trajectory)
    names: list
    motion_og: sk.Motion                    # the sliced input animation (apply_results' anim_og)
    com_og: np.ndarray                      # its centre of mass per frame (com_og)
    targetmap: Dict[int, np.ndarray] = field(default_factory=dict)
    heels_added: bool = False


def prepare(res: TowrResults, motion: sk.Motion, names, start_idx, end_idx, character: Character) -> BackProjectionTask:
    """apply_results up to the solver call (towr_utils.py:782-840), from an already loaded animation."""
    anim = motion.frames(start_idx, end_idx)
    heels_added = False
    if character.heel_inds is None and res.feet_pos.shape[1] == 4:
        anim = add_heels(anim, character.toe_inds, character.ankle_inds)
        heels_added = True
    gpos = sk.positions_global(anim)
    com = centre_of_mass(gpos, character)
    upper = list(character.upper_body)
    upper_offsets = gpos[:, upper, :] - com[:, None, :]
    anim_og = anim.copy()
    F = anim.n_frames                                                    # = end_idx - start_idx for a long enough file
    desired = upper_offsets + res.base_pos[:F, None, :] * 100.0          # metres -> the animation's centimetres
    anim.rotations[:, 0, :] = sk.quat_from_euler(res.base_rot, order='xyz', world=True)[:F]
    anim.positions[:, 0, :] = desired[:, 0, :]
    targetmap = {}
    for i, j in enumerate(upper):
        targetmap[int(j)] = desired[:, i, :]
    targetmap[int(character.toe_inds[0])] = res.feet_pos[:F, 0, :] * 100.0
    targetmap[int(character.toe_inds[1])] = res.feet_pos[:F, 1, :] * 100.0
    if res.feet_pos.shape[1] == 4:
        lh, rh = (anim.n_joints - 2, anim.n_joints - 1) if character.heel_inds is None else character.heel_inds
        targetmap[int(lh)] = res.feet_pos[:F, 2, :] * 100.0
        targetmap[int(rh)] = res.feet_pos[:F, 3, :] * 100.0
    return BackProjectionTask(motion=anim, names=list(names), motion_og=anim_og, com_og=com, targetmap=targetmap, heels_added=heels_added)


def back_project(tasks: Sequence[BackProjectionTask], solver) -> None:
    """The solver call of apply_results (towr_utils.py:841-843) for all videos at once; the tasks' motions are updated
    in place like `ik()` updates `anim`."""
    seqs = []
    for t in tasks:
        joints = np.array(list(t.targetmap.keys()), dtype=np.int32)
        targets = np.stack([np.asarray(v, dtype=np.float64) for v in t.targetmap.values()], axis=0)
        seqs.append(dict(parents=np.asarray(t.motion.parents, dtype=np.int32), target_joints=joints, targets=targets,
                         rot=np.ascontiguousarray(t.motion.rotations), pos=np.ascontiguousarray(t.motion.positions)))
    for t, (rot, pos) in zip(tasks, solver.solve(seqs)):
        t.motion.rotations = rot
        t.motion.positions = pos


def finish(task: BackProjectionTask, out_bvh: str) -> None:
    """towr_utils.py:971-975: drop the appended heels and write the BVH (BVH.save defaults)."""
    sk.save_bvh(out_bvh, remove_heels(task.motion) if task.heels_added else task.motion, task.names)


def apply_results_batch(solution_files, anim_bvhs, out_bvhs, characters, solver, starts=None, ends=None, animations=None):
    """`--viz --out-bvh` of towr_utils.py (:951-975) for a list of videos: parse, prepare, ONE batched solver call, write.
    `animations`: a dict path -> (Motion, names, frame time) that is filled with the parsed input animations and consulted first -- the driver back-projects three
    solution kinds onto the SAME animations; the files that are missing from it are read in one call of the native reader (prepare_capi.load_bvh_batch: all files on
    the host's cores) instead of one Python parse per video and kind (0.3 of the 0.5 s this stage took for 32 videos)."""
    n = len(solution_files)
    starts = starts or [None] * n
    ends = ends or [None] * n
    chars = characters if isinstance(characters, (list, tuple)) else [characters] * n
    cache = animations if animations is not None else {}
    missing = sorted({p for p in anim_bvhs if p not in cache})
    if missing:
        from . import prepare_capi
        for p, parsed in zip(missing, prepare_capi.load_bvh_batch(missing)):
            cache[p] = parsed
    tasks = []
    for k in range(n):
        motion, names, _ = cache[anim_bvhs[k]]
        tasks.append(prepare(load_towr_results(solution_files[k], flip_coords=True), motion, names, starts[k], ends[k], chars[k]))
    back_project(tasks, solver)
    for k in range(n):
        finish(tasks[k], out_bvhs[k])
    return tasks
