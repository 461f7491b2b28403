"""Pieces of bench.py that can run without a GPU: the contact-net side metric (on the CPU device here) and the defaults
the driver relies on."""
import importlib.util
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_contact_net_rate_fields():
    b = _bench()
    r = b.contact_net_rate(torch.device('cpu'), n_videos=3, frames=40, reps=2)
    assert r['unit'] == 'frames/s' and r['fps'] > 0 and r['fps_end_to_end'] > 0
    assert r['windows'] == 3 * (40 - 8) and r['dtype'] == 'f32' and r['videos'] == 3 and r['frames'] == 40
    assert r['fps_end_to_end_device_ops'] > 0 and r['device_ops_labels_equal'] is True


def test_defaults():
    b = _bench()
    assert b.FRAMES == 90 and b.BATCH == 128 and b.DEFAULT_IN_FLIGHT >= 1
    assert os.environ.get('GPU_MAX_HW_QUEUES') is not None          # set before torch initialises HIP
