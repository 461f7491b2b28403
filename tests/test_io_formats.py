"""File formats at the drop-in boundary: input writer -> reader round trip and the output writer
against the line-number-indexed parser of towr_utils.load_results."""
import os

import numpy as np

import chd_amd
from chd_amd import io_formats as iof
from chd_amd.synth import make_walk


def test_input_round_trip(tmp_path):
    seq = make_walk(seed=7, F=30, randomize=True)
    d = str(tmp_path / 'phys_optim_in_ybot')
    iof.write_inputs(seq, d)
    assert sorted(os.listdir(d)) == ['contact_info.txt', 'motion_info.txt', 'skel_info.txt', 'terrain_info.txt']
    back = iof.read_inputs(d, seq.F)
    for name in ('hip_l', 'hip_r', 'inertia', 'com', 'euler', 'ltoe', 'lheel', 'rtoe', 'rheel', 'normal', 'point'):
        assert np.array_equal(getattr(back, name), np.asarray(getattr(seq, name), dtype=np.float64)), name      # str(float) round-trips exactly
    assert back.start_contact == list(seq.start_contact)
    assert all(np.array_equal(a, b) for a, b in zip(back.durations, seq.durations))
    assert abs(sum(seq.durations[0]) - (seq.F - 1) * seq.dt) < 1e-9        # towr_utils.py:440


def test_solution_file_layout(tmp_path):
    S = 12
    rng = np.random.default_rng(0)
    sol = iof.Solution(dt=1 / 30, num_frames=S, base_lin=rng.normal(size=(S, 3)), base_ang_deg=rng.normal(size=(S, 3)) * 50,
                       ee_pos=rng.normal(size=(4, S, 3)), ee_force=rng.normal(size=(4, S, 3)) * 300, contact=rng.integers(0, 2, (4, S)))
    p = str(tmp_path / 'sol_out_dynamics.txt')
    iof.write_solution(sol, p)
    lines = open(p).read().split('\n')
    assert lines[0] == 'dt' and lines[2] == 'num_frames' and lines[4] == 'num_feet' and lines[6] == 'base_lin' and lines[8] == 'base_ang'
    assert [lines[10 + 2 * i] for i in range(4)] == ['foot%d_pos' % i for i in range(4)]
    assert not lines[7].endswith(' ') and len(lines[7].split(' ')) == 3 * S
    back = iof.load_results(p)
    assert back.num_frames == S
    assert np.allclose(back.base_lin, sol.base_lin, rtol=1e-9) and np.allclose(back.ee_force, sol.ee_force, rtol=1e-9)   # 10 significant digits
    assert np.array_equal(back.contact, sol.contact)


def test_solution_file_against_reference_parser(tmp_path):
    """tests/golden/io_golden.npz (tests/golden/make_io_golden.py): a file written by io_formats.write_solution, parsed
    by the REFERENCE's own `towr_utils.load_results` (towr_utils.py:51-121).  The writer still produces that exact file,
    and the arrays the reference obtained from it are the written ones in the reference's frame (y/z swapped, negated
    with flip_coords; contact flags transposed), to the 10 significant digits of the file."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'io_golden.npz'))
    sol = iof.Solution(dt=1 / 30, num_frames=int(g['in_base_lin'].shape[0]), base_lin=g['in_base_lin'], base_ang_deg=g['in_base_ang_deg'],
                       ee_pos=g['in_ee_pos'], ee_force=g['in_ee_force'], contact=g['in_contact'])
    p = str(tmp_path / 'sol.txt')
    iof.write_solution(sol, p)
    assert open(p, 'rb').read() == g['file_text'].tobytes()
    swap = [0, 2, 1]
    for tag, sgn in (('flip', -1.0), ('noflip', 1.0)):
        assert np.allclose(g[tag + '_base_pos'], sgn * sol.base_lin[:, swap], rtol=1e-9, atol=1e-12)
        for i in range(4):
            assert np.allclose(g[tag + '_feet_pos'][:, i, :], sgn * sol.ee_pos[i][:, swap], rtol=1e-9, atol=1e-12)
            assert np.allclose(g[tag + '_feet_force'][:, i, :], sgn * sol.ee_force[i][:, swap], rtol=1e-9, atol=1e-9)
        assert np.array_equal(g[tag + '_feet_contact'], np.asarray(sol.contact).T)
        R = g[tag + '_base_R']
        assert np.allclose(R @ np.transpose(R, (0, 2, 1)), np.eye(3)[None], atol=1e-6)       # the reference got valid rotations
    back = iof.load_results(p)
    assert np.allclose(back.base_lin, sol.base_lin, rtol=1e-9) and np.array_equal(back.contact, sol.contact)


def test_contact_durations_against_reference():
    """io_golden.npz `dur*`: phase durations computed by the reference's `find_contact_durations`
    (towr_utils.py:435-449) from random contact flags; ours are bit-identical (same `+= dt` accumulation)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'io_golden.npz'))
    for case in range(3):
        c = g['dur%d_contacts' % case]; dt = float(g['dur%d_dt' % case])
        ours = np.array(iof.contact_durations(c, dt))
        assert ours.shape == g['dur%d_ref' % case].shape and np.array_equal(ours, g['dur%d_ref' % case])
        assert abs(ours.sum() - (len(c) - 1) * dt) < 1e-9
