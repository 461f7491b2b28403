"""libchd_prepare.so (include/chd_prepare.h): the native BVH reader against the Python reader, and the HIP kernel's source -- through its
host emulation here, on the MI355X in tests/test_config4_gpu.py -- against the NumPy mirror of `prepare_input`, which tests/test_prepare_input.py pins to the
files the REFERENCE's own prepare_input wrote."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest
import torch

import chd_amd  # noqa: F401
from chd_amd import apply_results as ar
from chd_amd import prepare_capi as pc
from chd_amd import prepare_input as pi
from chd_amd import skeleton_io as sk

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, 'golden')); sys.path.insert(0, os.path.join(HERE, 'host_emu'))


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(HERE, 'golden', 'apply_golden.npz'))


@pytest.fixture(scope='module')
def character():
    from make_apply_golden import CHARACTER
    return ar.Character(**CHARACTER)


def test_library_loads_and_exports_the_header(gold):
    pc.build_library()
    lib = C.CDLL(pc.LIB_PATH)
    src = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'chd_prepare.h')).read(), flags=re.S)
    names = sorted(set(re.findall(r'\b(chd_[a-z_]+)\s*\(', src)))
    assert set(names) == set(pc.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n
    assert pc.load_library().chd_prep_version() == pc.ABI_VERSION
    # the ctypes mirror has the C layout (sizes through gcc)
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, 's.c'), 'w').write('#include <stdio.h>\n#include "%s"\nint main(){printf("%%zu %%zu\\n", sizeof(chd_prep_skeleton), sizeof(chd_bvh_clip));return 0;}' % os.path.join(ROOT, 'include', 'chd_prepare.h'))
        subprocess.check_call(['gcc', os.path.join(td, 's.c'), '-o', os.path.join(td, 's')])
        a, b = [int(v) for v in subprocess.check_output([os.path.join(td, 's')]).split()]
    assert a == C.sizeof(pc.ChdPrepSkeleton) and b == C.sizeof(pc.ChdBvhClip)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_kernel_entry_fails_loudly_without_a_gpu(gold, tmp_path, character):
    bvh = str(tmp_path / 'in.bvh'); open(bvh, 'wb').write(gold['bvh_text'].tobytes())
    m, _, _ = sk.load_bvh(bvh)
    with pytest.raises(RuntimeError, match='no HIP device'):
        pi.prepare_sequences_device([m], [(np.array([0.0, 0.0, 1.0]), np.zeros(3))], [gold['prep_contacts']], character, device='cuda:0')


def _bvh_variants(gold, tmp_path):
    """the fixture's file (3 channels per joint), the same motion with 6 channels on every joint, a file with End Sites in odd places and Windows line ends"""
    base = gold['bvh_text'].tobytes().decode()
    p0 = str(tmp_path / 'a.bvh'); open(p0, 'w').write(base)
    m, names, ft = sk.load_bvh(p0)
    p1 = str(tmp_path / 'b.bvh'); sk.save_bvh(p1, m, names, frametime=ft)
    p2 = str(tmp_path / 'c.bvh'); open(p2, 'w', newline='').write(open(p1).read().replace('\n', '\r\n'))
    # six channels on every joint: rewrite the CHANNELS lines and the motion rows by hand
    lines = open(p1).read().split('\n')
    cut = lines.index('MOTION')
    hdr = [re.sub(r'CHANNELS 3 (\w+) (\w+) (\w+)', r'CHANNELS 6 Xposition Yposition Zposition \1 \2 \3', ln) for ln in lines[:cut]]
    J = m.n_joints
    rows = []
    for ln in lines[cut + 3:]:
        v = ln.split()
        if not v:
            continue
        out = v[:6]
        for j in range(1, J):
            out += ['%f' % x for x in m.offsets[j]] + v[3 + 3 * j:6 + 3 * j]
        rows.append(' '.join(out))
    p3 = str(tmp_path / 'd.bvh'); open(p3, 'w').write('\n'.join(hdr + lines[cut:cut + 3] + rows) + '\n')
    return [p0, p1, p2, p3]


def test_native_bvh_reader_equals_the_python_reader(gold, tmp_path):
    paths = _bvh_variants(gold, tmp_path)
    got = pc.load_bvh_batch(paths, n_threads=3)
    for p, (m, names, ft) in zip(paths, got):
        rm, rnames, rft = sk.load_bvh(p)
        assert names == rnames and ft == rft and np.array_equal(m.parents, rm.parents)
        assert np.array_equal(m.offsets, rm.offsets) and np.array_equal(m.positions, rm.positions)            # numbers parsed by strtod / float(): identical
        assert m.rotations.shape == rm.rotations.shape and np.abs(m.rotations - rm.rotations).max() <= 4e-16    # libm's sin / cos against NumPy's vector ones
        assert np.array_equal(m.orients, rm.orients)


def test_native_bvh_reader_reports_bad_files(tmp_path, gold):
    good = str(tmp_path / 'ok.bvh'); open(good, 'wb').write(gold['bvh_text'].tobytes())
    cases = {'missing.bvh': None, 'nomotion.bvh': 'HIERARCHY\nROOT Hips\n{\nOFFSET 0 0 0\nCHANNELS 3 Zrotation Yrotation Xrotation\n}\n',
             'short.bvh': open(good).read().rsplit('\n', 3)[0] + '\n', 'empty.bvh': ''}
    for name, text in cases.items():
        p = str(tmp_path / name)
        if text is not None:
            open(p, 'w').write(text)
        with pytest.raises(ValueError, match=name.split('.')[0]):
            pc.load_bvh_batch([good, p])
        with pytest.raises((ValueError, FileNotFoundError, IndexError)):
            sk.load_bvh(p)                                                                                    # the Python reader refuses the same files


def test_kernel_source_equals_the_numpy_mirror(gold, tmp_path, character):
    """the HIP kernel's source through its host emulation on the three-clip batch of tests/test_prepare_input.py (different lengths, a scaled rotation,
    a shifted root), frame ranges included"""
    import prep_emu
    from test_prepare_input import ARRAYS, SCALARS, _batch_inputs
    clips, floors, fcs, starts, ends = _batch_inputs(gold, tmp_path)
    got = pi.prepare_sequences_device(clips, floors, fcs, character, starts, ends, dt=1.0 / 30.0, frames_fn=prep_emu.frames)
    for b, seq in enumerate(got):
        ref = pi.prepare_sequence(clips[b], floors[b], fcs[b], character, starts[b], ends[b], dt=1.0 / 30.0)
        assert seq.F == ref.F and seq.start_contact == ref.start_contact and seq.durations == ref.durations
        for k in ARRAYS:
            assert np.allclose(getattr(seq, k), getattr(ref, k), rtol=1e-11, atol=1e-13), (b, k)
        for k in SCALARS:
            assert abs(getattr(seq, k) - getattr(ref, k)) <= 1e-12 * abs(getattr(ref, k)), (b, k)
    # ... and the tensor-operation form of rounds 2-4 agrees with both (an independent third implementation)
    tor = pi.prepare_sequences_device(clips, floors, fcs, character, starts, ends, dt=1.0 / 30.0, device='cpu', backend='torch')
    for a, b in zip(got, tor):
        for k in ARRAYS:
            assert np.allclose(getattr(a, k), getattr(b, k), rtol=1e-11, atol=1e-13), k


def test_skeleton_tables_are_validated(gold, tmp_path, character):
    import prep_emu
    bvh = str(tmp_path / 'in.bvh'); open(bvh, 'wb').write(gold['bvh_text'].tobytes())
    m, _, _ = sk.load_bvh(bvh)
    anim = ar.add_heels(m, character.toe_inds, character.ankle_inds) if character.heel_inds is None else m
    s = pc.skeleton_of(character, [int(a) for a in anim.parents], m.n_joints)
    assert s.n_joints == anim.n_joints and s.n_joints_body == m.n_joints and s.n_segments == len(character.seg_to_joints)
    assert abs(sum(s.seg_mass_fraction[i] for i in range(s.n_segments)) - sum(character.seg_to_mass_perc.values()) * 0.01) < 1e-15
    L = pc.load_library()
    out = np.zeros((1, pc.OUT_STRIDE))
    bad = pc.ChdPrepSkeleton.from_buffer_copy(s); bad.parents[3] = 7                       # a joint in front of its parent
    rc = L.chd_prep_frames(C.byref(bad), 0, 1, anim.rotations[:1].ctypes.data_as(pc.PD), anim.positions[:1].ctypes.data_as(pc.PD), out.ctypes.data_as(pc.PD))
    assert rc != 0 and b'parents' in L.chd_prep_last_error()
    # round 6 (advisor): every index the kernel follows is checked before the launch -- a negative segment start, a parent below -1, a root with a parent
    for edit, word in ((lambda b: b.seg_first.__setitem__(0, -2), b'seg_first'), (lambda b: b.parents.__setitem__(5, -3), b'parents'), (lambda b: b.parents.__setitem__(0, 2), b'parents[0]')):
        bad = pc.ChdPrepSkeleton.from_buffer_copy(s); edit(bad)
        rc = L.chd_prep_frames(C.byref(bad), 0, 1, anim.rotations[:1].ctypes.data_as(pc.PD), anim.positions[:1].ctypes.data_as(pc.PD), out.ctypes.data_as(pc.PD))
        assert rc != 0 and word in L.chd_prep_last_error(), L.chd_prep_last_error()
