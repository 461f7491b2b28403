"""Host side of the IK back-projection row: BVH reader / writer, `load_results` post-processing and
`apply_results` against vectors produced by the REFERENCE's own functions (tests/golden/make_apply_golden.py).

The product's solver is the HIP library (no CPU path); here its place is taken by the host emulation of the kernel
source and, as the checker, by the oracle."""
import os
import sys

import numpy as np
import pytest

import chd_amd  # noqa: F401
from chd_amd import apply_results as ar
from chd_amd import io_formats as iof
from chd_amd import skeleton_io as sk
from oracle import ik_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
sys.path.insert(0, os.path.join(HERE, 'host_emu'))


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(HERE, 'golden', 'apply_golden.npz'))


@pytest.fixture(scope='module')
def character():
    from make_apply_golden import CHARACTER
    return ar.Character(**CHARACTER)


def _write(tmp_path, name, arr):
    p = str(tmp_path / name)
    open(p, 'wb').write(arr.tobytes())
    return p


def _same_rotation(a, b, tol):
    return np.minimum(np.abs(a - b).max(-1), np.abs(a + b).max(-1)).max() < tol


class OracleSolver:
    def solve(self, seqs):
        return [ik_oracle.ik_ck(s['rot'], s['pos'], s['parents'], s['target_joints'], s['targets']) for s in seqs]


class EmuSolver:
    def solve(self, seqs):
        import ik_emu
        return ik_emu.solve(seqs)


def test_bvh_reader_matches_reference(gold, tmp_path):
    motion, names, ft = sk.load_bvh(_write(tmp_path, 'in.bvh', gold['bvh_text']))
    assert names == list(gold['names']) and 'mixamorig:LeftToeEnd' in names          # mixamo-style names with a colon
    assert np.array_equal(motion.parents, gold['load_parents']) and np.array_equal(motion.offsets, gold['load_offsets'])
    assert abs(ft - float(gold['load_frametime'])) < 1e-12
    assert np.array_equal(motion.positions, gold['load_pos'])
    assert np.abs(motion.rotations - gold['load_rot']).max() < 1e-15
    assert np.abs(sk.positions_global(motion) - gold['load_gpos']).max() < 1e-11


def test_bvh_writer_is_byte_identical_to_reference(gold, tmp_path):
    motion, names, _ = sk.load_bvh(_write(tmp_path, 'in.bvh', gold['bvh_text']))
    p = str(tmp_path / 'out.bvh')
    sk.save_bvh(p, motion, names)                               # BVH.save(filename, anim, names): frame time 1/24, 'zyx'
    assert open(p, 'rb').read() == gold['ref_save_text'].tobytes()
    again, names2, ft = sk.load_bvh(p)                          # and it reads back
    assert names2 == names and abs(ft - 0.041667) < 1e-12
    assert _same_rotation(again.rotations, motion.rotations, 1e-6)
    sk.save_bvh(p, motion)                                      # default joint names
    assert 'ROOT joint_0' in open(p).read()


def test_bvh_reader_refuses_malformed_files(gold, tmp_path):
    text = gold['bvh_text'].tobytes().decode()
    p = str(tmp_path / 'bad.bvh')
    open(p, 'w').write(text[:text.index('MOTION')])
    with pytest.raises(ValueError, match='no MOTION'):
        sk.load_bvh(p)
    open(p, 'w').write(text[:-200])                             # last frame cut short
    with pytest.raises(ValueError, match='motion values'):
        sk.load_bvh(p)
    six = text.replace('CHANNELS 3 Zrotation Yrotation Xrotation', 'CHANNELS 6 Xposition Yposition Zposition Zrotation Yrotation Xrotation')
    open(p, 'w').write(six)                                     # now declares 6 channels per joint but holds 3
    with pytest.raises(ValueError, match='motion values'):
        sk.load_bvh(p)


def test_six_channel_files(tmp_path):
    """Every joint with its own translation channels (BVH.py:151-154)."""
    rng = np.random.default_rng(0)
    F, J = 3, 3
    rows = rng.normal(size=(F, J, 6)) * 20
    body = ('HIERARCHY\nROOT a0\n{\n OFFSET 0 0 0\n CHANNELS 6 Xposition Yposition Zposition Yrotation Xrotation Zrotation\n JOINT a1\n {\n  OFFSET 1 2 3\n'
            '  CHANNELS 6 Xposition Yposition Zposition Yrotation Xrotation Zrotation\n  JOINT a2\n  {\n   OFFSET 0 5 0\n'
            '   CHANNELS 6 Xposition Yposition Zposition Yrotation Xrotation Zrotation\n   End Site\n   {\n    OFFSET 0 1 0\n   }\n  }\n }\n}\n'
            'MOTION\nFrames: %d\nFrame Time: 0.0333333\n' % F)
    body += '\n'.join(' '.join('%.9g' % v for v in rows[f].reshape(-1)) for f in range(F)) + '\n'
    p = str(tmp_path / 'six.bvh')
    open(p, 'w').write(body)
    m, names, ft = sk.load_bvh(p)
    assert names == ['a0', 'a1', 'a2'] and list(m.parents) == [-1, 0, 1] and np.array_equal(m.offsets[2], [0, 5, 0])
    assert np.allclose(m.positions, rows[..., :3], rtol=1e-8)
    want = sk.quat_from_euler(np.radians(rows[..., 3:]), order='yxz', world=False)
    assert np.allclose(m.rotations, want, atol=1e-9)


def test_load_results_postprocessing_matches_reference(gold, tmp_path):
    res = ar.load_towr_results(_write(tmp_path, 'sol.txt', gold['sol_text']), flip_coords=True)
    assert np.array_equal(res.base_pos, gold['res_base_pos']) and np.array_equal(res.feet_pos, gold['res_feet_pos'])
    assert np.abs(res.base_rot - gold['res_base_rot']).max() < 1e-14
    g = np.load(os.path.join(HERE, 'golden', 'io_golden.npz'))                      # second fixture: both flip settings, base_R
    p = _write(tmp_path, 'sol2.txt', g['file_text'])
    for tag, flip in (('flip', True), ('noflip', False)):
        r = ar.load_towr_results(p, flip_coords=flip)
        assert np.abs(r.base_rot - g[tag + '_base_rot']).max() < 1e-14 and np.abs(r.base_R - g[tag + '_base_R']).max() < 1e-14
        assert np.array_equal(r.feet_force, g[tag + '_feet_force']) and np.array_equal(r.feet_contact, g[tag + '_feet_contact'])


def _task(gold, tmp_path, character):
    motion, names, _ = sk.load_bvh(_write(tmp_path, 'in.bvh', gold['bvh_text']))
    res = ar.load_towr_results(_write(tmp_path, 'sol.txt', gold['sol_text']))
    s, e = [int(v) for v in gold['start_end']]
    return ar.prepare(res, motion, names, s, e, character)


def test_prepare_matches_reference_apply_results_without_ik(gold, tmp_path, character):
    t = _task(gold, tmp_path, character)
    assert t.heels_added and np.array_equal(t.motion.parents, gold['noik_parents']) and np.array_equal(t.motion.offsets, gold['noik_offsets'])
    assert np.abs(t.motion.rotations - gold['noik_rot']).max() < 1e-14
    assert np.abs(t.motion.positions - gold['noik_pos']).max() < 1e-11
    assert np.abs(t.motion_og.rotations - gold['og_rot']).max() < 1e-15 and np.array_equal(t.motion_og.positions, gold['og_pos'])
    assert np.abs(t.com_og - gold['com_og']).max() < 1e-11
    assert list(t.targetmap.keys()) == list(character.upper_body) + [14, 19, 20, 21]      # the order the solver's rows follow
    assert np.abs(sk.positions_global(t.motion) - gold['noik_gpos']).max() < 1e-10


@pytest.mark.parametrize('solver', [OracleSolver, EmuSolver])
def test_back_projection_matches_reference_apply_results(gold, tmp_path, character, solver):
    t = _task(gold, tmp_path, character)
    ar.back_project([t], solver())
    assert np.abs(sk.positions_global(t.motion) - gold['ik_gpos']).max() < 1e-7
    assert np.abs(t.motion.positions - gold['ik_pos']).max() < 1e-7
    assert _same_rotation(t.motion.rotations, gold['ik_rot'], 1e-8)
    out = str(tmp_path / 'out.bvh')
    ar.finish(t, out)
    ref = gold['out_bvh_text'].tobytes().decode()
    got = open(out).read()
    cut = ref.index('MOTION')
    assert got[:got.index('MOTION')] == ref[:cut]                                       # heels removed again, same hierarchy text
    a = np.array(got[got.index('Time:') + 5:].split(), dtype=np.float64)
    b = np.array(ref[ref.index('Time:') + 5:].split(), dtype=np.float64)
    assert a.shape == b.shape and np.abs(a - b).max() <= 2e-6                           # '%f' digits


def test_batch_driver_writes_one_bvh_per_video(gold, tmp_path, character):
    bvh = _write(tmp_path, 'in.bvh', gold['bvh_text']); sol = _write(tmp_path, 'sol.txt', gold['sol_text'])
    s, e = [int(v) for v in gold['start_end']]
    outs = [str(tmp_path / 'a.bvh'), str(tmp_path / 'b.bvh')]
    tasks = ar.apply_results_batch([sol, sol], [bvh, bvh], outs, character, EmuSolver(), starts=[s, s], ends=[e, e])
    assert len(tasks) == 2 and open(outs[0]).read() == open(outs[1]).read()
    m, names, _ = sk.load_bvh(outs[0])
    assert m.n_joints == 20 and m.n_frames == e - s


def test_bvh_round_trip_on_random_trees(tmp_path):
    """Writer -> reader on random joint trees (any branching, leaves anywhere in depth-first order): hierarchy exact, offsets
    and root translations to the '%f' digits, rotations to 1e-6."""
    rng = np.random.default_rng(42)
    for case in range(8):
        J = int(rng.integers(1, 24))
        parents = [-1]
        for j in range(1, J):                                   # depth-first order: the parent is the previous joint or one of its ancestors
            cand = [j - 1]
            while parents[cand[-1]] >= 0:
                cand.append(parents[cand[-1]])
            parents.append(int(rng.choice(cand)))
        F = int(rng.integers(1, 6))
        offsets = np.round(rng.normal(size=(J, 3)) * 20, 4); offsets[0] = 0
        eul = rng.uniform(-80, 80, size=(F, J, 3))              # away from the +-90 degree singularity of the middle angle
        pos = np.repeat(offsets[None], F, axis=0); pos[:, 0] = np.round(rng.normal(size=(F, 3)) * 50, 4)
        m = sk.Motion(sk.quat_from_euler(np.radians(eul), order='zyx', world=False), pos, np.tile([1.0, 0, 0, 0], (J, 1)), offsets, np.array(parents))
        names = ['j%d' % k for k in range(J)]
        p = str(tmp_path / ('t%d.bvh' % case))
        sk.save_bvh(p, m, names, frametime=0.04)
        back, names2, ft = sk.load_bvh(p)
        assert names2 == names and list(back.parents) == parents and abs(ft - 0.04) < 1e-9
        assert np.abs(back.offsets - offsets).max() < 1e-6 and np.abs(back.positions[:, 0] - pos[:, 0]).max() < 1e-6
        assert np.array_equal(back.positions[:, 1:], np.repeat(back.offsets[None, 1:], F, axis=0))
        assert _same_rotation(back.rotations, m.rotations, 1e-6)
        assert np.abs(sk.positions_global(back) - sk.positions_global(m)).max() < 1e-3
