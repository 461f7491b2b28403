"""The IK back-projection kernel source (chd_ik_kernels.hpp, the code hipcc compiles for gfx950), compiled for the host,
against the vectors produced by the reference solver (tests/golden/ik_golden.npz) and against the oracle."""
import os
import sys

import numpy as np
import pytest

import chd_amd  # noqa: F401
from chd_amd.ik_capi import ChdIkConfig
from oracle import ik_oracle as ik

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'host_emu'))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ik_golden.npz')


@pytest.fixture(scope='module')
def emu():
    import ik_emu
    ik_emu.build()
    return ik_emu


def _cases(g):
    return [dict(parents=g['c%d_parents' % c], target_joints=g['c%d_target_joints' % c], targets=g['c%d_targets' % c],
                 rot=g['c%d_rot0' % c], pos=g['c%d_pos0' % c]) for c in range(int(g['n_cases']))]


@pytest.mark.parametrize('iters', [1, 30])
def test_kernel_source_matches_reference_vectors(emu, iters):
    """Both golden cases in one batch (different skeletons and frame counts): global joint positions and local
    translations of the reference after 1 and 30 iterations to 1e-8 (dual-form solve, matrix-form axes)."""
    g = np.load(GOLD)
    cases = _cases(g)
    outs = emu.solve(cases, ChdIkConfig.default(iterations=iters))
    for c, (rot, pos) in enumerate(outs):
        gp = ik.positions_global(rot, pos, cases[c]['parents'])
        assert np.allclose(gp, g['c%d_it%d_gpos' % (c, iters)], rtol=1e-8, atol=1e-8)
        assert np.allclose(pos, g['c%d_it%d_pos' % (c, iters)], rtol=1e-8, atol=1e-8)
        ref = g['c%d_it%d_rot' % (c, iters)]
        assert np.minimum(np.abs(rot - ref).max(-1), np.abs(rot + ref).max(-1)).max() < 1e-8


def test_without_translation_and_input_checks(emu):
    """translate = 0 leaves the joint translations untouched and still agrees with the oracle; malformed input is refused."""
    g = np.load(GOLD)
    cs = _cases(g)[0]
    (rot, pos), = emu.solve([cs], ChdIkConfig.default(iterations=5, translate=0))
    assert np.array_equal(pos, cs['pos'])
    rot_o, pos_o = ik.ik_ck(cs['rot'], cs['pos'], cs['parents'], cs['target_joints'], cs['targets'], iterations=5, translate=False)
    assert np.allclose(ik.positions_global(rot, pos, cs['parents']), ik.positions_global(rot_o, pos_o, cs['parents']), rtol=1e-8, atol=1e-8)
    bad = dict(cs); bad['parents'] = np.array([-1, 2, 1, 2, 0, 4, 5, 0])          # parent after child
    with pytest.raises(RuntimeError):
        emu.solve([bad])
