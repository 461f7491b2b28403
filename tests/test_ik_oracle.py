"""SURVEY 8(f) rank 1 groundwork: the numpy restatement of the reference's IK back-projection solver
(oracle/ik_oracle.py) against golden vectors produced by the reference solver itself
(tests/golden/make_ik_golden.py -> tests/golden/ik_golden.npz).  No product path exists for this row yet."""
import os

import numpy as np
import pytest

from oracle import ik_oracle as ik

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ik_golden.npz')


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


def _same_rotation(q, r, tol):
    """q and -q are the same rotation."""
    d = np.minimum(np.abs(q - r).max(axis=-1), np.abs(q + r).max(axis=-1))
    return d.max() < tol


def test_building_blocks(gold):
    for ci in range(int(gold['n_cases'])):
        k = 'c%d_' % ci
        rot0, pos0, parents = gold[k + 'rot0'], gold[k + 'pos0'], gold[k + 'parents']
        assert np.allclose(ik.quat_to_euler_xyz(rot0), gold[k + 'euler_of_rot0'], atol=1e-13)
        assert _same_rotation(ik.quat_from_euler_xyz_world(gold[k + 'euler_of_rot0']), rot0, 1e-9)      # (the 1e-10 of from_angle_axis)
        assert np.allclose(ik.positions_global(rot0, pos0, parents), gold[k + 'gpos0'], rtol=1e-12, atol=1e-12)
        gq = ik.quat_from_matrix(ik.transforms_global(rot0, pos0, parents))
        assert np.allclose(gq, gold[k + 'quat_of_global'], atol=1e-12)
        # quaternion -> matrix -> quaternion round trip, and rotation of a vector against the matrix form
        v = np.array([0.3, -1.2, 0.7])
        assert np.allclose(ik.quat_rotate(rot0, np.broadcast_to(v, rot0.shape[:-1] + (3,))), ik.quat_to_matrix(rot0) @ v, atol=1e-13)
    d = ik.descendants_mask(np.array([-1, 0, 1, 2, 0, 4, 5, 0]))
    assert d[0].sum() == 7 and list(np.flatnonzero(d[4])) == [5, 6] and d[3].sum() == 0


@pytest.mark.parametrize('iters', [1, 30])
def test_solver_matches_reference(gold, iters):
    """Same local rotations / positions and global joint positions as JacobianInverseKinematicsCK after 1 and after
    30 iterations (apply_results' setting), to 1e-8 relative (measured ~1e-12; LU vs numpy.linalg.solve)."""
    for ci in range(int(gold['n_cases'])):
        k = 'c%d_' % ci
        rot, pos = ik.ik_ck(gold[k + 'rot0'], gold[k + 'pos0'], gold[k + 'parents'], gold[k + 'target_joints'], gold[k + 'targets'],
                            iterations=iters, damping=7.0, smoothness=0.001, translate=True)
        kk = 'c%d_it%d_' % (ci, iters)
        assert _same_rotation(rot, gold[kk + 'rot'], 1e-9)
        assert np.allclose(pos, gold[kk + 'pos'], rtol=1e-8, atol=1e-8)
        gp = ik.positions_global(rot, pos, gold[k + 'parents'])
        assert np.allclose(gp, gold[kk + 'gpos'], rtol=1e-8, atol=1e-8)
        # the solver does what it is for: the targeted joints move towards their targets
        tj = gold[k + 'target_joints']; tg = np.swapaxes(gold[k + 'targets'], 0, 1)
        e0 = np.linalg.norm(gold[k + 'gpos0'][:, tj] - tg, axis=-1).mean()
        e1 = np.linalg.norm(gp[:, tj] - tg, axis=-1).mean()
        assert e1 < e0


def test_dual_form_matches_reference(gold):
    """The step solved as J^T (J J^T + lambda^2 I)^-1 e (3T x 3T instead of 6J x 6J; what the HIP path will do) gives the
    reference's result after 30 iterations to 1e-8."""
    for ci in range(int(gold['n_cases'])):
        k = 'c%d_' % ci
        rot, pos = ik.ik_ck(gold[k + 'rot0'], gold[k + 'pos0'], gold[k + 'parents'], gold[k + 'target_joints'], gold[k + 'targets'],
                            iterations=30, damping=7.0, smoothness=0.001, translate=True, dual=True)
        gp = ik.positions_global(rot, pos, gold[k + 'parents'])
        assert np.allclose(gp, gold['c%d_it30_gpos' % ci], rtol=1e-8, atol=1e-8)
        assert np.allclose(pos, gold['c%d_it30_pos' % ci], rtol=1e-8, atol=1e-8)
