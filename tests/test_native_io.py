"""The native text-file reader / writer of the drop-in boundary (csrc/chd_io.hpp, used by `chd_phys_solve_dirs`)
compiled for the host: against the Python mirror (io_formats, itself pinned to the reference's parser by
tests/golden/io_golden.npz), byte for byte, and on the malformed inputs the reference's `operator>>` readers
(phys_optim.cpp:155-267) would silently mis-read -- ours must refuse them with a message."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import chd_amd
from chd_amd import io_formats as iof
from chd_amd.synth import make_walk

_HERE = os.path.dirname(os.path.abspath(__file__))
PD = C.POINTER(C.c_double)


@pytest.fixture(scope='module')
def io():
    src = os.path.join(_HERE, 'host_emu', 'io_emu.cpp')
    so = os.path.join(_HERE, 'host_emu', 'libio_emu.so')
    hdr = os.path.join(os.path.dirname(_HERE), 'contact-human-dynamics_amd', 'csrc', 'chd_io.hpp')
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-o', so, src])
    L = C.CDLL(so)
    L.io_emu_read.argtypes = [C.c_char_p, C.c_int, PD, C.POINTER(C.c_int), C.c_char_p, C.c_int]
    L.io_emu_write.argtypes = [C.c_char_p, C.c_double, C.c_int, C.c_int, C.c_int, PD, PD, PD, PD, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_char_p, C.c_int]
    return L


def native_read(io, d, F):
    reals = np.zeros(30 * F + 11 + 256)
    ints = (C.c_int * 8)()
    err = C.create_string_buffer(512)
    rc = io.io_emu_read(d.encode(), F, reals.ctypes.data_as(PD), ints, err, 512)
    return rc, err.value.decode(), reals, list(ints)


def test_reader_matches_python_mirror(io, tmp_path):
    for seed, F, tilt in ((7, 30, 0.0), (3, 90, 4.0)):
        seq = make_walk(seed=seed, F=F, randomize=True, tilt_deg=tilt)
        d = str(tmp_path / ('in_%d' % seed))
        iof.write_inputs(seq, d)
        rc, msg, reals, ints = native_read(io, d, F)
        assert rc == 0, msg
        py = iof.read_inputs(d, F)
        want = np.concatenate([np.ravel(py.hip_l), np.ravel(py.hip_r), np.ravel(py.inertia)] +
                              [np.ravel(getattr(py, k)) for k in ('com', 'euler', 'ltoe', 'lheel', 'rtoe', 'rheel')] +
                              [[py.dt, py.leg_len, py.heel_len, py.heel_dist, py.mass], py.normal, py.point] + [np.asarray(x) for x in py.durations])
        assert np.array_equal(reals[:want.size], want)          # both parsers round correctly: bit-identical doubles
        assert ints[:4] == list(py.start_contact) and ints[4:] == [len(x) for x in py.durations]


def test_reader_token_semantics_and_refusals(io, tmp_path):
    seq = make_walk(seed=1, F=20, randomize=True)
    F = seq.F
    d = str(tmp_path / 'in')
    iof.write_inputs(seq, d)
    ref = native_read(io, d, F)[2]

    def rewrite(name, fn):
        p = os.path.join(d, name)
        old = open(p).read()
        open(p, 'w').write(fn(old))
        return p, old

    # layout of white space is irrelevant (operator>> token stream); exponent notation is a number; trailing tokens are ignored
    p, old = rewrite('motion_info.txt', lambda s: '\n\n  ' + '\t\n'.join(s.split()) + '  \n 1e3 extra')
    rc, msg, reals, _ = native_read(io, d, F)
    assert rc == 0 and np.array_equal(reals, ref), msg
    open(p, 'w').write(old)
    p, old = rewrite('terrain_info.txt', lambda s: '0 0 1.0E+00\n0.0 0 -0e0\n')
    rc, msg, reals, _ = native_read(io, d, F)
    assert rc == 0 and list(reals[30 * F + 5:30 * F + 11]) == [0, 0, 1, 0, 0, 0]
    open(p, 'w').write(old)

    cases = [
        ('motion_info.txt', lambda s: ' '.join(s.split()[:-1]), 'expected %d tokens, found %d' % (1 + 18 * F, 18 * F)),      # ragged: one number short
        ('motion_info.txt', lambda s: '', 'expected 1 tokens, found 0'),                                                     # empty file
        ('skel_info.txt', lambda s: s.replace(s.split()[5], 'abc', 1), "bad number 'abc'"),
        ('contact_info.txt', lambda s: 'true' + s[1:], 'start flag must be 0 or 1'),                                           # operator>> into bool only takes 0 / 1
        ('contact_info.txt', lambda s: s.split('\n')[0].split()[0] + ' 0\n', 'phase count < 1'),
        ('contact_info.txt', lambda s: '\n'.join(s.split('\n')[:3]) + '\n', 'truncated'),                                      # fourth end effector missing
        ('contact_info.txt', lambda s: '1 5 0.1 0.2\n', 'expected 7 tokens, found 4'),                                         # fewer durations than announced
    ]
    for name, fn, expect in cases:
        p, old = rewrite(name, fn)
        rc, msg, _, _ = native_read(io, d, F)
        assert rc == 1 and expect in msg and name in msg, (name, msg)
        open(p, 'w').write(old)
    assert native_read(io, d, F)[0] == 0
    os.remove(os.path.join(d, 'terrain_info.txt'))
    rc, msg, _, _ = native_read(io, d, F)
    assert rc == 1 and 'cannot open' in msg and 'terrain_info.txt' in msg
    # more frames announced on the command line than the files hold
    iof.write_inputs(seq, d)
    rc, msg, _, _ = native_read(io, d, F + 1)
    assert rc == 1 and 'skel_info.txt: expected' in msg


def test_writer_is_byte_identical_to_python_mirror(io, tmp_path):
    rng = np.random.default_rng(5)
    S, cap = 61, 64                                  # fewer samples than capacity: end-effector blocks are capacity-strided
    base_lin = rng.normal(size=(cap, 3)); base_ang = rng.normal(size=(cap, 3)) * 90
    ee_pos = rng.normal(size=(4, cap, 3)); ee_force = rng.normal(size=(4, cap, 3)) * 400
    contact = rng.integers(0, 2, (4, cap)).astype(np.uint8)
    # values that stress "%.10g": zero, negative zero, tiny, huge, exact integers, a 10-digit rounding boundary
    base_lin[0] = [0.0, -0.0, 1e-5]; base_lin[1] = [123456789012.0, -1e-300, 2.0]; base_lin[2] = [0.99999999995, 1 / 3, -1e15]
    ee_force[0, 0] = [1000.0, 1e-11, -999.99999999]
    d = str(tmp_path / 'out'); os.makedirs(d)
    err = C.create_string_buffer(512)
    for dyn, dur in ((1, 0), (0, 0), (1, 1)):
        rc = io.io_emu_write(d.encode(), 1 / 30, cap, S, S + 0, base_lin.ctypes.data_as(PD), base_ang.ctypes.data_as(PD), ee_pos.ctypes.data_as(PD),
                             ee_force.ctypes.data_as(PD), contact.ctypes.data_as(C.POINTER(C.c_ubyte)), dyn, dur, err, 512)
        assert rc == 0, err.value
        assert sorted(os.listdir(d)) == ['sol_out_durations.txt', 'sol_out_dynamics.txt', 'sol_out_no_dynamics.txt', 'success_log.txt']
        sol = iof.Solution(dt=1 / 30, num_frames=S, base_lin=base_lin[:S], base_ang_deg=base_ang[:S], ee_pos=ee_pos[:, :S], ee_force=ee_force[:, :S], contact=contact[:, :S])
        p = str(tmp_path / 'py.txt')
        iof.write_solution(sol, p)
        want = open(p, 'rb').read()
        for name in ('sol_out_no_dynamics.txt', 'sol_out_dynamics.txt', 'sol_out_durations.txt'):
            assert open(os.path.join(d, name), 'rb').read() == want, name
        iof.write_success_log(p, dyn, dur)
        assert open(os.path.join(d, 'success_log.txt'), 'rb').read() == open(p, 'rb').read()
    back = iof.load_results(os.path.join(d, 'sol_out_dynamics.txt'))
    assert back.num_frames == S and np.array_equal(back.contact, contact[:, :S])
    # an unwritable directory is reported, not ignored
    rc = io.io_emu_write(str(tmp_path / 'missing' / 'dir').encode(), 1 / 30, cap, S, S, base_lin.ctypes.data_as(PD), base_ang.ctypes.data_as(PD),
                         ee_pos.ctypes.data_as(PD), ee_force.ctypes.data_as(PD), contact.ctypes.data_as(C.POINTER(C.c_ubyte)), 1, 1, err, 512)
    assert rc == 1 and b'cannot write' in err.value
