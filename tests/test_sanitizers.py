"""The HOST side of libchd_phys.so under ThreadSanitizer and AddressSanitizer (SURVEY 5 "sanitizers").

tests/host_emu/pipeline_stress.cpp compiles contact-human-dynamics_amd/csrc/chd_phys.hip as C++ against a stand-in HIP runtime (tests/host_emu/hip_stub.hpp:
streams are threads, a launch starts one thread per resident workgroup running the host emulation of the kernel source), so the library's own pipelined call
-- table builder threads, four lanes with page-locked staging, finisher threads with their stage-4 fallback launches, workspace slots claimed by
compare-and-swap, workspace growth while launches are in flight -- runs with real concurrency under the sanitizers, for several chunk plans, plus a second
call on the warm handle.  The harness is validated by an injected race that ThreadSanitizer must report."""
import filecmp
import os
import subprocess
import sys

import pytest

import chd_amd  # noqa: F401
from chd_amd import io_formats as iof
from chd_amd.synth import make_walk

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, 'host_emu')
CSRC = os.path.join(os.path.dirname(HERE), 'contact-human-dynamics_amd', 'csrc')
SRCS = [os.path.join(EMU, 'pipeline_stress.cpp'), os.path.join(EMU, 'hip_stub.hpp')] + [os.path.join(CSRC, f) for f in ('chd_phys.hip', 'chd_kernels.hpp', 'chd_model.hpp', 'chd_device.hpp', 'chd_io.hpp')]
N_SEQ = 14


def build(tag, flags):
    exe = os.path.join(EMU, 'pipeline_stress_' + tag)
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in SRCS):
        subprocess.check_call(['g++', '-O1', '-g', '-std=c++17', '-x', 'c++', '-Wno-unused-variable', '-Wno-unused-value'] + flags + ['pipeline_stress.cpp', '-o', exe, '-pthread'], cwd=EMU)
    return exe


@pytest.fixture(scope='module')
def dirs(tmp_path_factory):
    root = tmp_path_factory.mktemp('stress')
    lines = []
    for i in range(N_SEQ):
        F = 24 + (i % 3) * 4 if i != N_SEQ - 3 else 44          # one longer sequence near the end: the workspaces must grow while earlier launches are in flight
        d_in = root / ('v%03d' % i) / 'in'; d_out = root / ('v%03d' % i) / 'out'
        iof.write_inputs(make_walk(seed=700 + i, F=F, randomize=True), str(d_in)); d_out.mkdir(parents=True)
        lines.append('%s %s %d' % (d_in, d_out, F))
    (root / 'list.txt').write_text('\n'.join(lines) + '\n')
    return root


def run(exe, root, max_iter, plans, env):
    p = subprocess.run([exe, str(root / 'list.txt'), str(max_iter)] + plans, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, CHD_STUB_CUS='6', **env), timeout=900)
    return p.returncode, p.stdout, p.stderr


def test_pipelined_call_is_clean_under_thread_sanitizer(dirs):
    """plans: chunks of 4 (four launches in flight on four lanes), chunks of 3 with TWO workspace slots (lanes reused behind stragglers, more resident workgroups
    than slots: the late ones wait), one chunk; results must not depend on the plan"""
    exe = build('tsan', ['-fsanitize=thread'])
    rc, out, err = run(exe, dirs, 5, ['4', '3:2', '-1'], dict(TSAN_OPTIONS='halt_on_error=0 report_signal_unsafe=0'))
    assert 'ThreadSanitizer' not in err, err[-4000:]
    assert rc == 0, (out, err[-2000:])
    assert out.count('plan ') == 3 and out.count('warm call: rc 0') == 3
    assert '4 chunks of 4' in out or '3 chunks of 4' in out
    for i in range(N_SEQ):
        base = dirs / ('v%03d' % i) / 'out'
        for name in ('sol_out_no_dynamics.txt', 'sol_out_dynamics.txt', 'sol_out_durations.txt', 'success_log.txt'):
            assert filecmp.cmp(base / 'plan0' / name, base / 'plan1' / name, shallow=False) and filecmp.cmp(base / 'plan0' / name, base / 'plan2' / name, shallow=False), (i, name)


def test_thread_sanitizer_sees_an_injected_race(dirs):
    """the harness is worth something only if it reports a race that is there: every finisher thread writes one field of the shared handle unsynchronised"""
    exe = build('tsan_inject', ['-fsanitize=thread', '-DCHD_STRESS_INJECT_RACE'])
    rc, out, err = run(exe, dirs, 2, ['4'], dict(TSAN_OPTIONS='halt_on_error=0 report_signal_unsafe=0'))
    assert 'ThreadSanitizer: data race' in err and 'chd_phys.hip' in err


def test_pipelined_call_is_clean_under_address_sanitizer(dirs):
    """heap overruns, use after free (round 4's finisher bug freed page-locked staging the handle still pointed to), double frees and leaks"""
    exe = build('asan', ['-fsanitize=address'])
    rc, out, err = run(exe, dirs, 5, ['4', '3:2'], dict(ASAN_OPTIONS='detect_leaks=1'))
    assert 'AddressSanitizer' not in err and 'LeakSanitizer' not in err, err[-4000:]
    assert rc == 0, (out, err[-2000:])
