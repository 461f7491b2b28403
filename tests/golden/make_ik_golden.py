"""Generates tests/golden/ik_golden.npz: inputs and outputs of the REFERENCE's IK back-projection solver
(`JacobianInverseKinematicsCK`, src/skeleton_fitting/ik/InverseKinematics.py:326-561, called by
`towr_utils.apply_results`, towr_utils.py:779-857 with translate=True, iterations=30, smoothness=0.001, damping=7.0) on
small synthetic skeletons -- SURVEY 8(f) rank 1, the consumer of the physics stage's output.

The reference library (Holden's `Animation` / `Quaternions`) imports `numpy.core.umath_tests`, which NumPy 2 no longer
has; this script installs a two-function stand-in (`matrix_multiply` = `np.matmul`) before importing it.  Run in the
build container only (it reads /root/reference); tests use the committed fixture."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_IK = '/root/reference/src/skeleton_fitting/ik'


def load_reference():
    shim = types.ModuleType('numpy.core.umath_tests')
    shim.matrix_multiply = np.matmul
    shim.inner1d = lambda a, b: np.einsum('...i,...i->...', a, b)
    sys.modules['numpy.core.umath_tests'] = shim
    sys.path.insert(0, REF_IK)
    import Animation
    import InverseKinematics
    from Quaternions import Quaternions
    return Animation, InverseKinematics, Quaternions


def make_case(rng, parents, F, targets_joints, noise):
    nj = len(parents)
    offsets = rng.normal(size=(nj, 3)) * 10.0
    offsets[0] = 0.0
    eul = rng.normal(size=(F, nj, 3)) * 0.3
    pos = np.repeat(offsets[None], F, axis=0).copy()
    pos[:, 0] = rng.normal(size=(F, 3)) * 5.0
    return dict(parents=np.asarray(parents), offsets=offsets, euler0=eul, pos0=pos, target_joints=np.asarray(targets_joints), noise=noise)


if __name__ == '__main__':
    Animation, IK, Q = load_reference()
    rng = np.random.default_rng(3)
    cases = [make_case(rng, [-1, 0, 1, 2, 0, 4, 5, 0], 5, [3, 6, 7], 2.0),
             make_case(rng, [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 10, 13], 4, [4, 8, 0, 12, 14], 3.0),
             # the size apply_results works at: a 33-joint humanoid tree (two 5-joint legs + heels, spine, head, two 6-joint
             # arms) with 13 targets (upper body + toes + heels)
             make_case(rng, [-1, 0, 1, 2, 3, 4, 0, 6, 7, 8, 9, 0, 11, 12, 13, 14, 13, 16, 17, 18, 19, 20, 13, 22, 23, 24, 25, 26, 3, 8, 15, 21, 27],
                       6, [0, 11, 12, 13, 14, 15, 16, 22, 19, 25, 4, 9, 28], 2.0)]
    out = {'n_cases': np.array(len(cases))}
    for ci, cs in enumerate(cases):
        nj = len(cs['parents']); F = cs['euler0'].shape[0]
        rot = Q.from_euler(cs['euler0'], order='xyz', world=True)
        anim = Animation.Animation(rot, cs['pos0'].copy(), Q.id(nj), cs['offsets'], cs['parents'])
        gp0 = Animation.positions_global(anim)
        targets = {}
        tg = np.zeros((len(cs['target_joints']), F, 3))
        for k, j in enumerate(cs['target_joints']):
            tg[k] = gp0[:, j] + rng.normal(size=(F, 3)) * cs['noise']
            targets[int(j)] = tg[k]
        for iters in (1, 30):
            a2 = Animation.Animation(Q(rot.qs.copy()), cs['pos0'].copy(), Q.id(nj), cs['offsets'], cs['parents'])
            ik = IK.JacobianInverseKinematicsCK(a2, targets, translate=True, iterations=iters, smoothness=0.001, damping=7.0, silent=True)
            ik()
            key = 'c%d_it%d_' % (ci, iters)
            out[key + 'rot'] = a2.rotations.qs.copy(); out[key + 'pos'] = np.asarray(a2.positions).copy()
            out[key + 'gpos'] = Animation.positions_global(a2)
        key = 'c%d_' % ci
        out[key + 'parents'] = cs['parents']; out[key + 'rot0'] = rot.qs.copy(); out[key + 'pos0'] = cs['pos0']
        out[key + 'target_joints'] = cs['target_joints']; out[key + 'targets'] = tg; out[key + 'gpos0'] = gp0
        # building blocks, for the unit tests of the restatement
        out[key + 'euler_of_rot0'] = rot.euler()
        out[key + 'quat_of_global'] = Q.from_transforms(Animation.transforms_global(anim)).qs
    np.savez_compressed(os.path.join(HERE, 'ik_golden.npz'), **out)
    print('wrote ik_golden.npz', sorted(out.keys()))
