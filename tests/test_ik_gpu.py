"""GPU parity of the IK back-projection step through its C ABI: libchd_ik.so on an MI355X against
the vectors produced by the reference solver.  NOT YET RUN ON A GPU (round 1's GPU budget went to the physics path):
marked `gpu_next`, i.e. outside `-m gpu` and outside `-m "not gpu"`'s expectations -- it skips without a GPU."""
import os

import numpy as np
import pytest

import chd_amd  # noqa: F401
from oracle import ik_oracle as ik

pytestmark = pytest.mark.gpu_next
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ik_golden.npz')


def test_hip_path_matches_reference_vectors():
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    from chd_amd.ik_backproject import IkBackProject
    from chd_amd.ik_capi import ChdIkConfig
    g = np.load(GOLD)
    cases = [dict(parents=g['c%d_parents' % c], target_joints=g['c%d_target_joints' % c], targets=g['c%d_targets' % c],
                  rot=g['c%d_rot0' % c], pos=g['c%d_pos0' % c]) for c in range(int(g['n_cases']))]
    for iters in (1, 30):
        outs = IkBackProject(device=0, config=ChdIkConfig.default(iterations=iters)).solve(cases)
        for c, (rot, pos) in enumerate(outs):
            gp = ik.positions_global(rot, pos, cases[c]['parents'])
            assert np.allclose(gp, g['c%d_it%d_gpos' % (c, iters)], rtol=1e-8, atol=1e-8)
            assert np.allclose(pos, g['c%d_it%d_pos' % (c, iters)], rtol=1e-8, atol=1e-8)
