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


def _fake_bench(tmp_path, body):
    """A copy of bench.py whose worker part is replaced by `body` (the wrapper logic stays as it is)."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    marker = "    import torch\n    import torch.distributed as dist\n"
    assert src.count(marker) == 1
    p = str(tmp_path / 'bench_fake.py')
    open(p, 'w').write(src.replace(marker, body + "    return\n" + marker))
    return p


def test_wrapper_falls_back_to_fewer_launches_in_flight(tmp_path):
    """The HIP runtime aborts the process when it cannot create the queues / scratch for the requested depth; the
    single-GPU wrapper must then repeat the measurement with half as many launches in flight and matching queue count."""
    import subprocess
    import sys
    p = _fake_bench(tmp_path, "    if args.pipeline >= 12: os._exit(134)\n"
                              "    print('{\"depth\": %d, \"queues\": \"%s\"}' % (args.pipeline, os.environ.get('GPU_MAX_HW_QUEUES')), flush=True)\n")
    r = subprocess.run([sys.executable, p], capture_output=True, text=True, env={k: v for k, v in os.environ.items() if k != 'WORLD_SIZE'})
    assert r.returncode == 0 and r.stdout.strip() == '{"depth": 6, "queues": "6"}' and 'failed (exit 134)' in r.stderr


def test_wrapper_keeps_the_measurement_when_a_later_leg_dies(tmp_path):
    import subprocess
    import sys
    p = _fake_bench(tmp_path, "    print('{\"value\": 1}', flush=True)\n    print('not json', flush=True)\n    os.abort()\n")
    r = subprocess.run([sys.executable, p], capture_output=True, text=True, env={k: v for k, v in os.environ.items() if k != 'WORLD_SIZE'})
    assert r.returncode == 0 and r.stdout.strip() == '{"value": 1}' and 'after the measurement' in r.stderr
