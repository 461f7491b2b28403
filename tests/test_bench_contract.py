"""Pieces of bench.py that can run without a GPU: the contact-net side metric (on the CPU device here) and the defaults
the driver relies on."""
import importlib
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module('bench')


def test_contact_net_rate_fields():
    b = _bench()
    r = b.contact_net_rate(torch.device('cpu'), n_videos=3, frames=40, reps=2)
    assert r['unit'] == 'frames/s' and r['fps'] > 0 and r['fps_end_to_end'] > 0
    assert r['windows'] == 3 * (40 - 8) and r['dtype'] == 'f32' and r['videos'] == 3 and r['frames'] == 40
    assert r['fps_end_to_end_device_ops'] > 0 and r['device_ops_labels_equal'] is True


def test_defaults():
    b = _bench()
    assert b.FRAMES == 90 and b.BATCH == 128 and b.CAPS == [7000, 7000, 7000, 2500, 2000, 7000]
    src = open(os.path.join(ROOT, 'bench.py')).read()
    # one handle, one stream, one persistent launch: no hardware-queue tuning, no child-process retry ladder
    assert 'GPU_MAX_HW_QUEUES' not in src and 'subprocess' not in src


def test_sequence_generation_is_seeded_and_order_preserving():
    b = _bench()
    import numpy as np
    a = b.make_sequences(5, 70, workers=3)
    c = b.make_sequences(5, 70, workers=1)
    assert len(a) == 70 and all(np.array_equal(x.com, y.com) and x.mass == y.mass for x, y in zip(a, c))


def test_parity_block_on_fixture_values():
    """The parity block of the bench line: fed with results equal to the committed oracle vectors it reports zero error on
    every sequence and lists the sequences on which a stage failed in the oracle separately."""
    import numpy as np
    import pytest
    b = _bench()
    path = os.path.join(ROOT, 'tests', 'golden', 'bench_parity_golden.npz')
    if not os.path.exists(path):
        pytest.skip('fixture not generated')
    g = np.load(path)

    class Snap:
        pass

    class Res:
        pass

    res = []
    for seed in range(128):
        key = 's%d_F90_t000' % seed
        r = Res(); r.snapshots = []
        st = list(g[key + '_status']); it = list(g[key + '_iters'])
        r.stage_status = st + [9] * (6 - len(st)); r.stage_iters = it + [0] * (6 - len(it)); r.stage_stalled = [0] * 6
        for k in range(3):
            sn = Snap()
            for name in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force', 'contact'):
                setattr(sn, name, g['%s_snap%d_%s' % (key, k, name)])
            r.snapshots.append(sn)
        res.append(r)
    out = b.parity_block(res, 0)
    assert out['sequences_compared'] == 128 and out['worst_rel_l2'] == 0.0 and out['sequences_above_1e-3'] == []
    assert out['stage_status_equal'] == 128 and out['stage_iterations_equal'] == 128 and out['contact_flags_equal'] == 128
    res[3].snapshots[2].ee_force = res[3].snapshots[2].ee_force * (1 + 2e-3)
    res[5].stage_stalled = [0, 0, 0, 0, 1, 0]
    out = b.parity_block(res, 0)
    assert out['sequences_above_1e-3'] == [3] and out['not_compared_ended_by_stall_guard'] == [5] and out['sequences_compared'] == 127
