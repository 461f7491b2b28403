"""Generates tests/golden/apply_golden.npz: the REFERENCE's BVH reader / writer and its `apply_results`
(src/utils/towr_utils.py:779-857, with `add_heel_to_anim` / `remove_heel_from_anim` :401-433) run on a synthetic
20-joint character -- SURVEY 8(f) rank 1, the stage that turns the physics output into the BVH a user gets.

* the BVH file is written by this repo's writer (skeleton_io.save_bvh) and read by the reference's `BVH.load`;
  the reference's `BVH.save` of what it read gives the bytes our writer must reproduce;
* the solution file is written by io_formats.write_solution and parsed by the reference's `load_results`;
* `apply_results` and `prepare_input` (towr_utils.py:451-777) are cut out of towr_utils.py with `ast` (the module as a whole needs matplotlib etc.) and run with
  the reference's own BVH / Animation / Quaternions / InverseKinematics modules; the character look-ups
  (character_info_utils getters) are replaced by the synthetic character's tables.

Run in the build container only (it reads /root/reference); tests use the committed fixture."""
import ast
import os
import sys
import types
from copy import deepcopy

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import chd_amd  # noqa: E402,F401
from chd_amd import io_formats as iof  # noqa: E402
from chd_amd import skeleton_io as sk  # noqa: E402

REF = '/root/reference/src'

# root, spine x3, head; left leg hip-knee-ankle-toe; right leg; left arm shoulder-elbow-wrist; right arm  (20 joints, depth-first)
NAMES = ['Hips', 'Spine', 'Spine1', 'Neck', 'Head', 'LeftArm', 'LeftForeArm', 'LeftHand', 'RightArm', 'RightForeArm', 'RightHand',
         'LeftUpLeg', 'LeftLeg', 'LeftFoot', 'LeftToe', 'mixamorig:LeftToeEnd', 'RightUpLeg', 'RightLeg', 'RightFoot', 'RightToe']
PARENTS = [-1, 0, 1, 2, 3, 2, 5, 6, 2, 8, 9, 0, 11, 12, 13, 14, 0, 16, 17, 18]
OFFSETS = [[0, 0, 0], [0, 10, 0], [0, 12, 0.5], [0, 14, 0], [0, 9, 1], [7, 10, 0], [22, 0, 0], [20, 0, 0], [-7, 10, 0], [-22, 0, 0], [-20, 0, 0],
           [9, -4, 0], [0, -40, 0], [0, -38, 0], [0, -7, 12], [0, 0, 6], [-9, -4, 0], [0, -40, 0], [0, -38, 0], [0, -7, 12]]
CHARACTER = dict(toe_inds=[14, 19], ankle_inds=[13, 18], upper_body=[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10],
                 seg_to_joints={'trunk': [0, 1, 2, 3], 'head': [3, 4], 'l_arm': [5, 6, 7], 'r_arm': [8, 9, 10],
                                'l_thigh': [11, 12], 'l_shank': [12, 13], 'l_foot': [13, 14], 'r_thigh': [16, 17], 'r_shank': [17, 18], 'r_foot': [18, 19]},
                 seg_to_mass_perc={'trunk': 43.0, 'head': 7.0, 'l_arm': 5.0, 'r_arm': 5.0, 'l_thigh': 10.5, 'l_shank': 6.5, 'l_foot': 3.0,
                                   'r_thigh': 10.5, 'r_shank': 6.5, 'r_foot': 3.0},
                 left_leg_chain=[11, 12, 13, 14], hip_inds=[11, 16], mass=70.0)


def synthetic_motion(F, seed):
    rng = np.random.default_rng(seed)
    J = len(PARENTS)
    offsets = np.array(OFFSETS, dtype=np.float64)
    eul = np.cumsum(rng.normal(size=(F, J, 3)) * 1.5, axis=0) + rng.normal(size=(1, J, 3)) * 12.0          # degrees, zyx channel order
    pos = np.repeat(offsets[None], F, axis=0)
    pos[:, 0] = np.array([0.0, 95.0, 0.0]) + np.cumsum(rng.normal(size=(F, 3)) * 0.8, axis=0)
    rot = sk.quat_from_euler(np.radians(eul), order='zyx', world=False)
    return sk.Motion(rot, pos, np.tile(np.array([1.0, 0, 0, 0]), (J, 1)), offsets, np.array(PARENTS))


def load_reference():
    shim = types.ModuleType('numpy.core.umath_tests')
    shim.matrix_multiply = np.matmul
    shim.inner1d = lambda a, b: np.einsum('...i,...i->...', a, b)
    sys.modules['numpy.core.umath_tests'] = shim
    sys.path.insert(0, os.path.join(REF, 'skeleton_fitting', 'ik'))
    import Animation
    import BVH
    from InverseKinematics import JacobianInverseKinematicsCK
    from Quaternions import Quaternions
    src = open(os.path.join(REF, 'utils', 'towr_utils.py')).read()
    tree = ast.parse(src)
    want = ('TowrResults', 'load_results', 'apply_results', 'add_heel_to_anim', 'remove_heel_from_anim', 'prepare_input', 'find_contact_durations')
    keep = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in want]
    assert len(keep) == len(want)
    C = CHARACTER
    ns = {'np': np, 'os': os, 'deepcopy': deepcopy, 'BVH': BVH, 'Animation': Animation, 'Quaternions': Quaternions,
          'JacobianInverseKinematicsCK': JacobianInverseKinematicsCK, 'heeled_characters': [],
          'get_character_toe_inds': lambda ch: list(C['toe_inds']), 'get_character_ankle_inds': lambda ch: list(C['ankle_inds']),
          'get_character_upper_body': lambda ch: list(C['upper_body']), 'get_character_seg_to_joint_map': lambda ch: C['seg_to_joints'],
          'get_character_seg_to_mass_perc_map': lambda ch: C['seg_to_mass_perc'], 'get_character_heel_inds': lambda ch: None,
          'get_character_leg_chain': lambda ch, side='left': list(C['left_leg_chain']) if side == 'left' else [16, 17, 18, 19],
          'get_character_hip_inds': lambda ch: list(C['hip_inds']), 'get_character_mass': lambda ch: C['mass']}
    exec(compile(ast.Module(body=keep, type_ignores=[]), 'towr_utils.py', 'exec'), ns)
    return ns, BVH, Animation


if __name__ == '__main__':
    ns, BVH, Animation = load_reference()
    F_file, start, end = 14, 2, 12
    F = end - start
    motion = synthetic_motion(F_file, seed=21)
    bvh_path = '/tmp/apply_golden_in.bvh'
    sk.save_bvh(bvh_path, motion, NAMES, frametime=1.0 / 30.0)
    out = {'bvh_text': np.frombuffer(open(bvh_path, 'rb').read(), dtype=np.uint8), 'names': np.array(NAMES), 'start_end': np.array([start, end])}
    # --- the reference reads our file ...
    anim, names, ft = BVH.load(bvh_path)
    assert list(names) == NAMES
    out['load_rot'] = anim.rotations.qs.copy(); out['load_pos'] = np.asarray(anim.positions).copy()
    out['load_offsets'] = np.asarray(anim.offsets).copy(); out['load_parents'] = np.asarray(anim.parents).copy(); out['load_frametime'] = np.array(ft)
    out['load_gpos'] = Animation.positions_global(anim)
    # --- ... and writes it back with its defaults (frame time 1/24)
    BVH.save('/tmp/apply_golden_ref.bvh', anim, names)
    out['ref_save_text'] = np.frombuffer(open('/tmp/apply_golden_ref.bvh', 'rb').read(), dtype=np.uint8)
    # --- a solution roughly consistent with the motion (solver frame: z up, negated; metres)
    rng = np.random.default_rng(5)
    sl = sk.Motion(motion.rotations[start:end], motion.positions[start:end], motion.orients, motion.offsets, motion.parents)
    gp = sk.positions_global(sl)
    com = np.zeros((F, 3))
    for key, joints in CHARACTER['seg_to_joints'].items():
        com += CHARACTER['seg_to_mass_perc'][key] * 0.01 * gp[:, joints].mean(axis=1)
    to_solver = lambda p: -(p[..., [0, 2, 1]]) * 0.01           # noqa: E731
    S = F + 1                                                   # the solver writes one more sample than there are frames at times
    pad = lambda a: np.concatenate([a, a[-1:]], axis=0)         # noqa: E731
    heel = gp[:, CHARACTER['ankle_inds']] + np.array([0.0, -7.0, 0.0])
    ee = np.stack([gp[:, 14], gp[:, 19], heel[:, 0], heel[:, 1]], axis=0)          # toe, toe, heel, heel
    sol = iof.Solution(dt=1 / 30, num_frames=S, base_lin=pad(to_solver(com + rng.normal(size=(F, 3)) * 1.5)),
                       base_ang_deg=pad(rng.uniform(-25, 25, size=(F, 3))),
                       ee_pos=np.stack([pad(to_solver(e + rng.normal(size=(F, 3)) * 1.0)) for e in ee], axis=0),
                       ee_force=rng.normal(size=(4, S, 3)) * 200, contact=rng.integers(0, 2, (4, S)))
    sol_path = '/tmp/apply_golden_sol.txt'
    iof.write_solution(sol, sol_path)
    out['sol_text'] = np.frombuffer(open(sol_path, 'rb').read(), dtype=np.uint8)
    res = ns['load_results'](sol_path, flip_coords=True)
    out['res_base_pos'] = res.base_pos; out['res_base_rot'] = res.base_rot; out['res_feet_pos'] = res.feet_pos
    # --- apply_results without and with the solver
    for tag, run_ik in (('noik', False), ('ik', True)):
        a, nm, a_og, com_og = ns['apply_results'](ns['load_results'](sol_path, flip_coords=True), bvh_path, start, end, 'synthetic', run_ik=run_ik)
        out[tag + '_rot'] = a.rotations.qs.copy(); out[tag + '_pos'] = np.asarray(a.positions).copy()
        out[tag + '_parents'] = np.asarray(a.parents).copy(); out[tag + '_offsets'] = np.asarray(a.offsets).copy()
        out[tag + '_gpos'] = Animation.positions_global(a)
        if run_ik:
            out['og_rot'] = a_og.rotations.qs.copy(); out['og_pos'] = np.asarray(a_og.positions).copy(); out['com_og'] = com_og
            sa = ns['remove_heel_from_anim'](a)
            BVH.save('/tmp/apply_golden_out.bvh', sa, nm)
            out['out_bvh_text'] = np.frombuffer(open('/tmp/apply_golden_out.bvh', 'rb').read(), dtype=np.uint8)
    # --- prepare_input (towr_utils.py:451-777) on the same file: floor and contacts as the kinematic stage leaves them
    floor_path, contacts_path, prep_dir = '/tmp/apply_golden_floor.txt', '/tmp/apply_golden_contacts.npy', '/tmp/apply_golden_prep'
    open(floor_path, 'w').write('0.02 0.999 -0.03\n3.0 1.5 -2.0\n')
    fc = (rng.random((F_file, 4)) < 0.5).astype(np.int64)
    for k in range(0, F_file, 4):
        fc[k:k + 4] = fc[k]
    np.save(contacts_path, fc)
    out['prep_floor_text'] = np.frombuffer(open(floor_path, 'rb').read(), dtype=np.uint8); out['prep_contacts'] = fc
    for tag, combined in (('prep', False), ('prepc', True)):
        ns['prepare_input'](bvh_path, floor_path, contacts_path, prep_dir, 'synthetic', start_idx=start, end_idx=end, dt=1.0 / 30.0, combined_contacts=combined)
        for name in ('skel_info.txt', 'motion_info.txt', 'terrain_info.txt', 'contact_info.txt'):
            out[tag + '_' + name.split('.')[0]] = np.frombuffer(open(os.path.join(prep_dir, name), 'rb').read(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'apply_golden.npz'), **out)
    print('wrote apply_golden.npz', {k: v.shape for k, v in out.items()})
