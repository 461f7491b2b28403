"""`prepare_input` (BVH + floor + contacts -> the physics stage's four input files) against the files
the REFERENCE's own `prepare_input` (towr_utils.py:451-777) wrote for the same synthetic character
(tests/golden/apply_golden.npz, generator tests/golden/make_apply_golden.py)."""
import os
import sys

import numpy as np
import pytest

import chd_amd  # noqa: F401
from chd_amd import apply_results as ar
from chd_amd import io_formats as iof
from chd_amd import prepare_input as pi

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
FILES = ('skel_info.txt', 'motion_info.txt', 'terrain_info.txt', 'contact_info.txt')


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(HERE, 'golden', 'apply_golden.npz'))


@pytest.fixture(scope='module')
def character():
    from make_apply_golden import CHARACTER
    return ar.Character(**CHARACTER)


def _inputs(gold, tmp_path):
    bvh = str(tmp_path / 'in.bvh'); floor = str(tmp_path / 'floor_out.txt'); contacts = str(tmp_path / 'foot_contacts.npy')
    open(bvh, 'wb').write(gold['bvh_text'].tobytes()); open(floor, 'wb').write(gold['prep_floor_text'].tobytes())
    np.save(contacts, gold['prep_contacts'])
    return bvh, floor, contacts


@pytest.mark.parametrize('tag,combined', [('prep', False), ('prepc', True)])
def test_files_match_reference(gold, tmp_path, character, tag, combined):
    bvh, floor, contacts = _inputs(gold, tmp_path)
    s, e = [int(v) for v in gold['start_end']]
    ours = str(tmp_path / 'ours'); ref = str(tmp_path / 'ref'); os.makedirs(ref)
    seq = pi.prepare_input(bvh, floor, contacts, ours, character, start_idx=s, end_idx=e, dt=1.0 / 30.0, combined_contacts=combined)
    for name in FILES:
        open(os.path.join(ref, name), 'wb').write(gold[tag + '_' + name.split('.')[0]].tobytes())
        a = open(os.path.join(ours, name)).read().split(); b = open(os.path.join(ref, name)).read().split()
        assert len(a) == len(b), name                                             # same token stream layout
    # the integer tokens (start flags, phase counts) and the accumulated durations are exactly the reference's
    assert open(os.path.join(ours, 'contact_info.txt')).read().split() == open(os.path.join(ref, 'contact_info.txt')).read().split()
    assert open(os.path.join(ours, 'terrain_info.txt')).read() == open(os.path.join(ref, 'terrain_info.txt')).read()
    A = iof.read_inputs(ours, e - s); B = iof.read_inputs(ref, e - s)
    for k in ('hip_l', 'hip_r', 'inertia', 'com', 'euler', 'ltoe', 'lheel', 'rtoe', 'rheel'):
        assert np.allclose(getattr(A, k), getattr(B, k), rtol=1e-12, atol=1e-14), k
    for k in ('dt', 'leg_len', 'heel_len', 'heel_dist', 'mass'):
        assert abs(getattr(A, k) - getattr(B, k)) <= 1e-14 * abs(getattr(B, k)), k
    assert A.start_contact == B.start_contact == seq.start_contact
    # what the call returns is what it wrote
    assert np.array_equal(np.asarray(seq.com, dtype=np.float64), A.com) and seq.F == e - s


def test_combined_flag_changes_the_toe_schedule_only(gold):
    fc = gold['prep_contacts']
    s0, d0 = pi.contact_schedule(fc, 2, 12, 1 / 30, combined_contacts=False)
    s1, d1 = pi.contact_schedule(fc, 2, 12, 1 / 30, combined_contacts=True)
    assert d0[1] == d1[1] and d0[3] == d1[3] and s0[1] == s1[1] and s0[3] == s1[3]
    for d in d0 + d1:
        assert abs(sum(d) - 9 / 30) < 1e-12                                       # (F - 1) dt, towr_utils.py:440


def test_root_angle_unwrapping():
    """towr_utils.py:620-629: a wrap from +pi to -pi (previous value >= 0) gets 2 pi added, from -pi to +pi subtracted;
    a jump in the direction the reference's loop never recovers from is reported."""
    up = np.linspace(3.0, 3.4, 6)                                                 # crosses +pi
    wrapped = np.where(up > np.pi, up - 2 * np.pi, up)
    r = np.stack([wrapped, -wrapped, np.zeros(6)], axis=1)
    out = pi.unwrap_like_reference(r)
    assert np.allclose(out[:, 0], up) and np.allclose(out[:, 1], -up) and np.all(out[:, 2] == 0)
    bad = np.zeros((3, 3)); bad[0, 0] = 0.1; bad[1, 0] = 0.1 + 3.5               # previous >= 0 and the next value is 3.5 higher
    with pytest.raises(ValueError, match='loops forever'):
        pi.unwrap_like_reference(bad)


def test_missing_inputs_and_character_fields(gold, tmp_path, character):
    bvh, floor, contacts = _inputs(gold, tmp_path)
    with pytest.raises(FileNotFoundError):
        pi.prepare_input(bvh, str(tmp_path / 'nope.txt'), contacts, str(tmp_path / 'o'), character, 0, 5)
    from make_apply_golden import CHARACTER
    partial = ar.Character(**{k: v for k, v in CHARACTER.items() if k != 'mass'})
    with pytest.raises(ValueError, match='mass'):
        pi.prepare_input(bvh, floor, contacts, str(tmp_path / 'o'), partial, 0, 5)
    seq = pi.prepare_input(bvh, floor, contacts, str(tmp_path / 'o'), character)        # whole file when no range is given
    assert seq.F == 14


def test_driver_prepare_stage_writes_the_input_directories(gold, tmp_path, character, monkeypatch):
    """`run_phys_mocap --prepare` (the towr_utils.py child process of run_phys_mocap.py:137-150): lays out
    phys_optim_in_<character>/ for every video directory; the solve itself needs a GPU, so the driver is stopped there."""
    import json
    from make_apply_golden import CHARACTER
    from chd_amd import run_phys_mocap as drv
    for v in ('vidA', 'vidB'):
        kin = tmp_path / 'data' / v / 'kinematic_results'
        os.makedirs(kin)
        open(kin / 'synth_out.bvh', 'wb').write(gold['bvh_text'].tobytes())
        open(kin / 'floor_out.txt', 'wb').write(gold['prep_floor_text'].tobytes())
        np.save(kin / 'foot_contacts.npy', gold['prep_contacts'])
    cj = str(tmp_path / 'character.json')
    json.dump(CHARACTER, open(cj, 'w'))

    class Stop(Exception):
        pass

    def no_gpu(*a, **k):
        raise Stop()
    monkeypatch.setattr(drv, 'PhysOptim', no_gpu)
    with pytest.raises(Stop):
        drv.main(['--data', str(tmp_path / 'data'), '--character', 'synth', '--prepare', '--character-json', cj, '--nframes', '12'])
    for v in ('vidA', 'vidB'):
        d = str(tmp_path / 'data' / v / 'phys_optim_in_synth')
        assert sorted(os.listdir(d)) == sorted(FILES)
        assert iof.read_inputs(d, 12).F == 12
    with pytest.raises(SystemExit):
        drv.main(['--data', str(tmp_path / 'data'), '--character', 'synth', '--prepare'])          # tables are required
    json.dump(dict(CHARACTER, colour='red'), open(cj, 'w'))
    with pytest.raises(ValueError, match='unknown character fields'):
        ar.Character.from_json(cj)


def test_driver_out_bvh_stage(gold, tmp_path, character, monkeypatch):
    """`run_phys_mocap --out-bvh` (the `towr_utils.py --viz --out-bvh` child process of run_phys_mocap.py:180-201): three
    BVH files per video, named as the reference names them.  The physics solve is replaced by a stub that drops the
    fixture's solution file and the HIP solver by the host emulation of its kernel source (no GPU here)."""
    import json
    sys.path.insert(0, os.path.join(HERE, 'host_emu'))
    import ik_emu
    from make_apply_golden import CHARACTER
    import importlib
    from chd_amd import run_phys_mocap as drv
    from chd_amd import skeleton_io as sk
    ik_backproject = importlib.import_module(drv.__package__ + '.ik_backproject')      # the module object the driver's relative import resolves to
    kin = tmp_path / 'data' / 'clip' / 'kinematic_results'
    os.makedirs(kin)
    open(kin / 'synth_out.bvh', 'wb').write(gold['bvh_text'].tobytes())
    open(kin / 'floor_out.txt', 'wb').write(gold['prep_floor_text'].tobytes())
    np.save(kin / 'foot_contacts.npy', gold['prep_contacts'])
    cj = str(tmp_path / 'character.json')
    json.dump(CHARACTER, open(cj, 'w'))

    class StubPhys:
        def __init__(self, **k):
            pass

        def solve_dirs(self, ins, outs, nframes):
            for o in outs:
                for kind in ('no_dynamics', 'dynamics'):                       # the durations stage "failed": no file
                    open(os.path.join(o, 'sol_out_%s.txt' % kind), 'wb').write(gold['sol_text'].tobytes())
            return [0] * len(outs)

        def close(self):
            pass

    class EmuIk:
        def __init__(self, **k):
            pass

        def solve(self, seqs):
            return ik_emu.solve(seqs)
    monkeypatch.setattr(drv, 'PhysOptim', StubPhys)
    monkeypatch.setattr(ik_backproject, 'IkBackProject', EmuIk)
    rc = drv.main(['--data', str(tmp_path / 'data'), '--character', 'synth', '--prepare', '--out-bvh', '--character-json', cj, '--nframes', '10'])
    assert rc == 0
    out = tmp_path / 'data' / 'clip' / 'phys_optim_out_synth'
    got = sorted(f for f in os.listdir(out) if f.endswith('.bvh'))
    assert got == ['clip_synth_dynamics.bvh', 'clip_synth_no_dynamics.bvh']
    m, names, _ = sk.load_bvh(str(out / got[0]))
    assert m.n_frames == 10 and m.n_joints == 20 and names[0] == 'Hips'


ARRAYS = ('hip_l', 'hip_r', 'inertia', 'com', 'euler', 'ltoe', 'lheel', 'rtoe', 'rheel', 'normal', 'point')
SCALARS = ('dt', 'leg_len', 'heel_len', 'heel_dist', 'mass')


def _batch_inputs(gold, tmp_path):
    from chd_amd import skeleton_io as sk
    bvh, floor, contacts = _inputs(gold, tmp_path)
    m, _, _ = sk.load_bvh(bvh)
    s, e = [int(v) for v in gold['start_end']]
    rng = np.random.default_rng(3)
    clips = [m, m.frames(1, m.n_frames), m.frames(0, m.n_frames - 2)]                  # different lengths: the batch is padded over the frames
    clips[1].rotations[:, 3] = clips[1].rotations[:, 3] * np.array([1.0, 0.98, 1.02, 1.0]); clips[2].positions[:, 0] += rng.normal(size=3)
    fcs = [gold['prep_contacts'], gold['prep_contacts'][1:], gold['prep_contacts'][:-2]]
    fl = pi.read_floor(floor)
    return clips, [fl] * 3, fcs, [s, 0, 1], [e, None, 9]


def check_device_batch(gold, tmp_path, character, device):
    clips, floors, fcs, starts, ends = _batch_inputs(gold, tmp_path)
    got = pi.prepare_sequences_device(clips, floors, fcs, character, starts, ends, dt=1.0 / 30.0, device=device)
    for b, seq in enumerate(got):
        ref = pi.prepare_sequence(clips[b], floors[b], fcs[b], character, starts[b], ends[b], dt=1.0 / 30.0)
        assert seq.F == ref.F and seq.start_contact == ref.start_contact and seq.durations == ref.durations
        for k in ARRAYS:
            assert np.allclose(getattr(seq, k), getattr(ref, k), rtol=1e-11, atol=1e-13), (b, k)
        for k in SCALARS:
            assert abs(getattr(seq, k) - getattr(ref, k)) <= 1e-12 * abs(getattr(ref, k)), (b, k)


def test_batched_unwrapping_equals_the_sequential_one():
    rng = np.random.default_rng(6)
    clips = []
    for F in (5, 40, 17):
        a = np.cumsum(rng.normal(size=(F, 3)) * 0.9, axis=0)
        clips.append((a + np.pi) % (2 * np.pi) - np.pi)                       # wrapped into (-pi, pi]: jumps of ~2 pi to undo
    clips[1][:, 0] = np.abs(clips[1][:, 0])                                  # a positive previous value: the only direction the reference can undo there
    ok = []
    for c in clips:
        try:
            ok.append(pi.unwrap_like_reference(c))
        except ValueError:
            ok.append(None)
    good = [c for c, o in zip(clips, ok) if o is not None]
    for got, want in zip(pi.unwrap_batch(good), [o for o in ok if o is not None]):
        assert np.array_equal(got, want)
    if any(o is None for o in ok):
        with pytest.raises(ValueError):
            pi.unwrap_batch(clips)


def test_batched_tensor_path_equals_the_numpy_path(gold, tmp_path, character):
    """prepare_sequences_device on the CPU device of torch: the same tensor code the GPU runs (its GPU twin: tests/test_config4_gpu.py)."""
    check_device_batch(gold, tmp_path, character, 'cpu')
