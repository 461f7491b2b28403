"""The solver-kernel source, compiled for the host (one emulated thread), against the oracle.
This exercises exactly the code hipcc compiles for gfx950 (chd_kernels.hpp + chd_model.hpp), minus the
parallel execution, so algorithmic regressions are caught in the CPU-only container."""
import os
import sys

import numpy as np
import pytest

import chd_amd
from chd_amd.phys_capi import default_config
from chd_amd.synth import make_walk

from common import oracle_run, rel_max, snapshot_errors

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'host_emu'))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def emu():
    import emu as e
    e.build()
    return e


def test_eval_parity_all_stages(emu, oracle_lib):
    from oracle.oracle import OracleProblem
    seq = make_walk(seed=4, F=40, randomize=True, tilt_deg=3.0)
    o = OracleProblem(seq); e = emu.EmuProblem(seq)
    rng = np.random.default_rng(2)
    x0 = None
    for st in range(5):
        o.set_stage(st)
        sz = e.sizes(st)
        assert (sz['n'], sz['m']) == (o.n, o.m)
        if x0 is None:
            x0 = o.get_x()
        x = x0.copy() if st != 4 else np.concatenate([x0, o.get_x()[x0.size:]])
        pert = 0.01 * rng.normal(size=x.size)
        if st == 4:
            pert[x0.size:] *= 0.01
        x = x + pert
        fo, go, co, Jo, Ho = o.eval(x, jac=True, hess=True)
        r = e.eval(st, x)
        assert r['err'] == 0
        assert abs(r['f'] - fo) <= 1e-11 * abs(fo)
        for a, b in ((r['g'], go), (r['c'], co), (r['J'], Jo), (r['H'], Ho)):
            assert rel_max(a, b) < 1e-11
        o.set_x(x0 if st != 4 else np.concatenate([x0, o.get_x()[x0.size:]]))


def test_bordered_band_factorisation(emu):
    """The factorisation of the kernel source (right-looking bordered band L D L^T) solves K x = b."""
    seq = make_walk(seed=0, F=40, randomize=True)
    e = emu.EmuProblem(seq, default_config())
    for st in (1, 4):
        sz = e.sizes(st)
        b = np.random.default_rng(st).normal(size=sz['n'] + sz['m'])
        x, bad = e.linsolve(st, b, dw=1e-2, dval=1e-3, refine=2)
        assert bad == 0 and np.isfinite(x).all()


@pytest.mark.parametrize('seed,F,tilt', [(2, 40, 0.0), (6, 60, 5.0), (9, 90, 0.0), (103, 90, 0.0), (67, 90, 0.0)])
def test_staged_solve_parity(emu, oracle_lib, seed, F, tilt):
    """Kernel source (host emulation) vs oracle through all stages: same statuses, same iteration counts, snapshots to
    1e-8 -- flat and tilted floors, 40 / 60 / 90 frames; the last two are the sequences on which oracle and kernel parted ways in round 4 until the
    oracle took its KKT ordering from the structure like the kernel's table builder (the positional pivot test depends on the elimination order)."""
    seq = make_walk(seed=seed, F=F, randomize=True, tilt_deg=tilt)
    caps = [300] * 6
    e = emu.EmuProblem(seq, default_config(max_iter=caps))
    e.solve(0, 4)
    stats, snaps = e.results()
    if stats[4, 0] != 0:
        assert e.rebuild_fallback() == 1
        e.solve(5, 5)
        stats, snaps = e.results()
    ostats, osnaps = oracle_run(seq, caps)
    for st in range(len(ostats)):
        assert int(stats[st, 0]) == ostats[st][0] and int(stats[st, 1]) == ostats[st][1]
    from chd_amd.io_formats import Solution
    for k in range(3):
        s = snaps[k]
        sol = Solution(dt=seq.dt, num_frames=s['num_frames'], base_lin=s['base_lin'], base_ang_deg=s['base_ang_deg'],
                       ee_pos=s['ee_pos'], ee_force=s['ee_force'], contact=s['contact'])
        err = snapshot_errors(sol, osnaps[k])
        assert err['contact_mismatch'] == 0
        assert max(err['base_lin'], err['base_ang_deg'], err['ee_pos'], err['ee_force']) < 1e-8, err


@pytest.mark.parametrize('seed,F', [(6, 40), (3, 60)])
def test_damping_rule_1_keeps_kernel_source_and_oracle_in_lockstep(emu, oracle_lib, seed, F):
    """chd_config.damping_rule = 1 (round 6: the damping also grows after an ACCEPTED step with a poor actual / predicted merit reduction; off by default) is implemented in the
    kernel source and in the oracle (IpmOptions::ratio_low = 0.25): same statuses, same iteration counts, same snapshots -- and it is a different path from the default's (duration stage:
    21 -> 56 and 62 -> 31 iterations on these two; the MI355X test holds the bench straggler 1688 the same way)."""
    from oracle.oracle import lib
    seq = make_walk(seed=seed, F=F, randomize=True)
    caps = [7000, 7000, 7000, 2500, 2000, 7000]
    e = emu.EmuProblem(seq, default_config(max_iter=caps, damping_rule=1))
    e.solve(0, 4)
    stats, snaps = e.results()
    if stats[4, 0] != 0:
        assert e.rebuild_fallback() == 1
        e.solve(5, 5)
        stats, snaps = e.results()
    lib().orc_set_ratio_low(0.25)
    try:
        ostats, osnaps = oracle_run(seq, caps)
    finally:
        lib().orc_set_ratio_low(0.0)
    assert [(int(stats[st, 0]), int(stats[st, 1])) for st in range(len(ostats))] == [(s[0], s[1]) for s in ostats]
    from chd_amd.io_formats import Solution
    for k in range(3):
        s = snaps[k]
        sol = Solution(dt=seq.dt, num_frames=s['num_frames'], base_lin=s['base_lin'], base_ang_deg=s['base_ang_deg'], ee_pos=s['ee_pos'], ee_force=s['ee_force'], contact=s['contact'])
        err = snapshot_errors(sol, osnaps[k])
        assert err['contact_mismatch'] == 0 and max(err['base_lin'], err['base_ang_deg'], err['ee_pos'], err['ee_force']) < 1e-8, err
    d = emu.EmuProblem(seq, default_config(max_iter=caps)); d.solve(0, 4)
    assert int(d.results()[0][4, 1]) != int(stats[4, 1])          # (the rule is not a no-op)


def test_duration_block_of_lagrangian_hessian_vs_finite_differences(emu, oracle_lib):
    """Stage 3 uses the exact duration-duration block of the Lagrangian Hessian (second derivatives of the
    phase-based splines with respect to the phase durations).  Kernel source (class tables) and oracle (pairwise
    formulas) are written independently; both must match central differences of grad f + J^T lambda."""
    from oracle.oracle import OracleProblem
    seq = make_walk(seed=3, F=40, randomize=True, tilt_deg=3.0)
    e = emu.EmuProblem(seq)
    sz = e.sizes(4); n, m = sz['n'], sz['m']
    nd = n - sum(len(d) - 1 for d in seq.durations)
    rng = np.random.default_rng(0)
    x = e.eval(4)['x'].copy()
    x[:nd] += 0.01 * rng.normal(size=nd)
    x[nd:] *= 1 + 0.03 * rng.normal(size=n - nd)
    lam = 0.3 * rng.normal(size=m)
    A = e.eval(4, x, lam=lam)['H'][nd:, nd:]

    def grad_lagrangian(xx):
        r = e.eval(4, xx, hess=False)
        return r['g'] + r['J'].T @ lam
    h = 1e-6
    FD = np.zeros_like(A)
    for k in range(nd, n):
        xp = x.copy(); xm = x.copy(); xp[k] += h; xm[k] -= h
        FD[:, k - nd] = ((grad_lagrangian(xp) - grad_lagrangian(xm)) / (2 * h))[nd:]
    assert np.abs(A - FD).max() <= 1e-6 * np.abs(FD).max()
    o = OracleProblem(seq); o.set_stage(4)
    Ho = o.eval(x, jac=True, hess=True, lam=lam)[4]
    assert np.abs(Ho[nd:, nd:] - A).max() <= 1e-11 * np.abs(A).max()
    # the node x duration block (round 4): kernel source (records of the row tasks + sample cache, gathered per node variable) against the oracle's (per-family
    # scatter) and against the same finite differences; it also holds force / centre-of-mass / base-angle x duration entries
    Hk = e.eval(4, x, lam=lam)['H']
    FDx = np.zeros((nd, n - nd))
    for k in range(nd, n):
        xp = x.copy(); xm = x.copy(); xp[k] += h; xm[k] -= h
        FDx[:, k - nd] = ((grad_lagrangian(xp) - grad_lagrangian(xm)) / (2 * h))[:nd]
    assert np.abs(Hk[:nd, nd:] - FDx).max() <= 1e-6 * np.abs(FDx).max()
    assert np.abs(Hk[:nd, nd:] - Ho[:nd, nd:]).max() <= 1e-11 * np.abs(Ho[:nd, nd:]).max()
    assert np.abs(Hk[:nd, :nd] - Ho[:nd, :nd]).max() <= 1e-11 * np.abs(Ho[:nd, :nd]).max()


def test_kkt_structure_tables(emu):
    """Host structure tables the kernel relies on (chd_model.hpp): the first band position coupled to each border
    position is never later than the first numerically non-zero entry of that border row of the assembled KKT matrix
    (every stage, initial point), and a full staged solve raises no structure violation
    (the emulation build checks the band / border envelopes at every factorisation and prints on stderr)."""
    import ctypes as C
    seq = make_walk(seed=3, F=60, randomize=True)
    e = emu.EmuProblem(seq)
    for st in range(5):
        out = (C.c_double * 8)(); first = (C.c_int * 1024)()
        emu.lib().emu_border_first(C.c_void_p(e.h), st, out, first)
        bc, nb = int(out[1]), int(out[2])
        numeric = np.array(first[:bc]); structural = np.array(first[512:512 + bc])
        assert (numeric >= structural).all(), st
        assert (structural <= nb).all() and (structural < nb).any(), st
        st_ = (C.c_double * 8)()
        emu.lib().emu_nact_stats(C.c_void_p(e.h), st, st_)
        assert 0 < st_[0] <= e.sizes(st)['w'] + bc      # mean number of active window rows per panel
    e.solve(0, 4)
    stats = e.results()['stats'] if isinstance(e.results(), dict) else None
    assert stats is None or all(int(s[0]) in (0, -1, -2) for s in stats)


def test_blocked_border_factorisation_for_borders_beyond_lds(emu, oracle_lib, monkeypatch):
    """A border whose Schur complement does not fit LDS (600 frames: 700 rows) is factored by 16-column panels staged through LDS (dense_ldlt_blocked).  The
    emulation's LDS holds any border, so the path is forced (CHD_EMU_BORDER_IN_HBM): linear solves stay accurate and a whole staged solve stays in lockstep
    with the oracle."""
    monkeypatch.setenv('CHD_EMU_BORDER_IN_HBM', '1')
    seq = make_walk(seed=0, F=40, randomize=True)
    e = emu.EmuProblem(seq, default_config())
    for st in (1, 4):
        sz = e.sizes(st)
        b = np.random.default_rng(st).normal(size=sz['n'] + sz['m'])
        x, bad = e.linsolve(st, b, dw=1e-2, dval=1e-3, refine=2)
        assert bad == 0 and np.isfinite(x).all()
    test_staged_solve_parity(emu, oracle_lib, 6, 60, 5.0)


def test_occupancy_list_of_the_kkt_matrix(emu):
    """The readers of the unfactored KKT matrix walk a list of the entries ever written in the stage, kept by the writers through a signed-zero marker
    (chd_kernels.hpp "KKT storage").  After two evaluations per stage -- the second at moved durations in the duration stage, so that samples change
    polynomial and the pattern grows -- every stored non-zero is in the list, the list is exactly the set of mask bits, and the product over the list equals
    the dense product over the storage."""
    import ctypes as C
    L = emu.lib()
    L.emu_list_check.argtypes = [C.c_void_p, C.c_int, emu.PD, emu.PD, emu.PD, emu.PD, emu.PD]
    rng = np.random.default_rng(11)
    grown = 0
    for seed, F in ((3, 60), (7, 90)):
        e = emu.EmuProblem(make_walk(seed=seed, F=F, randomize=True))
        for st in range(5):
            sz = e.sizes(st)
            n, m = sz['n'], sz['m']
            x0 = e.eval(st, jac=False, hess=False)['x']
            x1 = x0 + 1e-3 * rng.normal(size=n)
            if st == 4:                                   # durations: shorten / lengthen phases by up to 4 % (much more overflows the band, which the solver reports as an error) so that samples cross polynomial boundaries
                nd = n - e.sizes(3)['n']
                x1[n - nd:] = x0[n - nd:] * (1 + 0.04 * rng.uniform(-1, 1, size=nd))
            lam0 = np.zeros(m); lam1 = rng.normal(size=m)
            out = np.zeros(8)
            err = L.emu_list_check(C.c_void_p(e.h), st, emu._p(lam0), emu._p(np.ascontiguousarray(x1)), emu._p(lam1), emu._p(rng.normal(size=n + m)), emu._p(out))
            assert err == 0, (seed, st)
            missing, listed, bits, perr, pmag, added = out[:6]
            assert missing == 0 and listed == bits and listed > 0, (seed, st, out)
            assert perr <= 1e-12 * max(pmag, 1.0), (seed, st, perr, pmag)
            if st == 4:
                grown += added
    assert grown > 0        # the second evaluation of the duration stage did move the pattern


def test_inertia_retry_switch_keeps_kernel_and_oracle_in_lockstep():
    """A sequence on which kernel source and oracle used to part ways in the duration stage (seed 31: at its second
    iteration the factorisation meets a pivot of unexpected sign, replaces it, and the oracle's line search accepted the
    resulting step while the kernel source's -- NaN -- was rejected).  With inertia retry (the default: CHD_INERTIA_RETRY=1,
    IpmOptions::inertia_retry) such a factorisation is a failed attempt and the two stay in lockstep to 1e-8; the variant
    without it (the code that ran on the GPU in round 1) is built as well to show the difference."""
    import subprocess
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests')); sys.path.insert(0, os.path.join(%r, 'tests', 'host_emu'))
import chd_amd
from chd_amd.phys_capi import default_config
from chd_amd.synth import make_walk
from chd_amd.io_formats import Solution
from common import oracle_run, snapshot_errors
from oracle import oracle as O
import emu
emu.build()
O.lib().orc_set_inertia_retry(int(os.environ.get('CHD_EMU_VARIANT', '') != 'noinertia'))
caps = [300] * 6
seq = make_walk(seed=31, F=90, randomize=True)
e = emu.EmuProblem(seq, default_config(max_iter=caps))
e.solve(0, 4)
stats, snaps = e.results()
ostats, osnaps = oracle_run(seq, caps)
same = all(int(stats[st, 0]) == ostats[st][0] and int(stats[st, 1]) == ostats[st][1] for st in range(5))
worst = 0.0
for k in range(3):
    s = snaps[k]
    sol = Solution(dt=seq.dt, num_frames=s['num_frames'], base_lin=s['base_lin'], base_ang_deg=s['base_ang_deg'], ee_pos=s['ee_pos'], ee_force=s['ee_force'], contact=s['contact'])
    err = snapshot_errors(sol, osnaps[k])
    worst = max(worst, err['base_lin'], err['base_ang_deg'], err['ee_pos'], err['ee_force'])
print('RESULT', int(same), '%%.3e' %% worst, [int(stats[st, 1]) for st in range(5)])
""" % (ROOT, ROOT, ROOT)
    out = {}
    for variant in ('', 'noinertia'):
        env = dict(os.environ, CHD_EMU_VARIANT=variant)
        r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')]
        assert line, r.stderr[-800:]
        out[variant] = line[0].split()
    assert out[''][1] == '1' and float(out[''][2]) < 1e-8, out                           # lockstep (default build)
    assert out['noinertia'][1] == '0' and float(out['noinertia'][2]) > 1e-3, out          # what the retry fixes


def test_model_switch_equals_a_second_evaluation_bit_for_bit(emu, monkeypatch, capfd):
    """Round 6: when the first model's factorisation fails, K0 becomes the second model by writing a side list of Gauss-Newton values over the entries the exact blocks changed
    (chd_kernels.hpp model_switch) instead of a second evaluation.  With CHD_EMU_SWITCH_CHECK the emulation evaluates the second model the old way as well after every switch and
    compares every listed entry and the objective bit for bit (a difference sets the error flag and prints a line); CHD_EMU_SWITCH_OFF takes the old path throughout: same iterates."""
    caps = [7000, 7000, 7000, 2500, 2000, 7000]
    out = {}
    for mode in ('check', 'off'):
        monkeypatch.delenv('CHD_EMU_SWITCH_CHECK', raising=False); monkeypatch.delenv('CHD_EMU_SWITCH_OFF', raising=False)
        monkeypatch.setenv('CHD_EMU_SWITCH_CHECK' if mode == 'check' else 'CHD_EMU_SWITCH_OFF', '2' if mode == 'check' else '1')
        rows = []
        for seed, F in ((3, 60), (8, 60), (1, 40)):
            e = emu.EmuProblem(make_walk(seed=seed, F=F, randomize=True), default_config(max_iter=caps))
            e.solve(0, 4)
            st, sn = e.results()
            rows.append(([(int(st[k][0]), int(st[k][1])) for k in range(5)], [np.asarray(sn[k]['ee_force']).copy() for k in range(3)]))
        out[mode] = rows
        err = capfd.readouterr().err
        assert 'MISMATCH' not in err
        if mode == 'check':
            assert err.count('model switch ok') >= 5          # (the path is exercised: stages with fixed durations and the duration stage)
    for a, b in zip(out['check'], out['off']):
        assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
