"""The synchronisation protocol of the kinematic optimisation's workgroup clusters (csrc/chd_kinopt_kernels.hpp, kc_sync: tagged 8-byte granules in two-deep slots,
an all-to-all gather of at least one value per synchronisation, the neighbour's halo, a bounded wait) replayed on host threads with relaxed atomics and random delays:
tests/host_emu/cluster_protocol.cpp.  It holds the protocol's LOGIC -- what the GPU tests cannot single out; the device's memory system is the GPU tests' business
(tests/test_kinopt_gpu.py::test_clusters_under_concurrent_calls_and_uploads)."""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'host_emu', 'cluster_protocol.cpp')


def build(tmp_path, tsan):
    exe = str(tmp_path / ('cluster_protocol_tsan' if tsan else 'cluster_protocol'))
    cmd = ['g++', '-O1', '-g', '-std=c++17', '-pthread'] + (['-fsanitize=thread'] if tsan else []) + [SRC, '-o', exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        pytest.skip('cannot build the protocol replay here: ' + r.stdout[-300:])
    return exe


def run(exe, *args):
    r = subprocess.run([exe] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


def test_every_value_read_is_the_one_published_for_that_synchronisation(tmp_path):
    exe = build(tmp_path, tsan=False)
    for G in (2, 8, 16):
        out, _ = run(exe, '--G=%d' % G, '--rounds=20000')
        assert out == {'G': G, 'rounds': 20000, 'finished': G, 'wrong_values': 0, 'gave_up': 0}


def test_without_the_gather_from_every_rank_two_deep_slots_are_not_enough(tmp_path):
    """The test of the test: with the neighbour's halo alone a workgroup can run two synchronisations ahead of the neighbour on its other side and overwrite what that one
    has not read yet -- a value is lost (its reader waits until the bound) or wrong."""
    exe = build(tmp_path, tsan=False)
    out, _ = run(exe, '--neighbours-only', '--rounds=20000')
    assert out['gave_up'] == 1 or out['wrong_values'] > 0


def test_a_missing_member_ends_in_the_bounded_wait(tmp_path):
    exe = build(tmp_path, tsan=False)
    out, _ = run(exe, '--absent=3', '--rounds=100')
    assert out['gave_up'] == 1 and out['finished'] == 0 and out['wrong_values'] == 0


def test_clean_under_thread_sanitizer(tmp_path):
    exe = build(tmp_path, tsan=True)
    out, err = run(exe, '--rounds=3000')
    assert out['finished'] == 8 and out['wrong_values'] == 0 and out['gave_up'] == 0
    assert 'ThreadSanitizer' not in err
