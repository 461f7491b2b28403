"""GPU parity: the HIP physics path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Tolerance for trajectories: 1e-3 relative L2 (BASELINE.json north_star);
contact flags bit-exact."""
import numpy as np
import pytest

import chd_amd
from chd_amd.synth import make_walk

from common import oracle_run, rel_max, snapshot_errors

pytestmark = pytest.mark.gpu
CAP = [300] * 6


@pytest.fixture(scope='module')
def solver():
    from chd_amd.phys_optim import PhysOptim, default_config
    s = PhysOptim(device=0, config=default_config(max_iter=CAP))
    yield s
    s.close()


def test_eval_matches_oracle(solver, oracle_lib):
    """f, gradient, constraint values, Jacobian and Gauss-Newton Hessian of every stage: 1e-10 relative."""
    from oracle.oracle import OracleProblem
    seq = make_walk(seed=0, F=60, randomize=True)
    o = OracleProblem(seq)
    b = solver.upload([seq])
    rng = np.random.default_rng(1)
    for st in range(5):
        o.set_stage(st)
        assert b.sizes(0, st)['n'] == o.n and b.sizes(0, st)['m'] == o.m
        x = o.get_x() if st == 0 else x0
        x0 = o.get_x() if st == 0 else x0
        x = x0[:o.n].copy() if st != 4 else np.concatenate([x0, o.get_x()[len(x0):]])
        x = x + 0.01 * rng.normal(size=x.size) * (1.0 if st != 4 else np.concatenate([np.ones(len(x0)), 0.01 * np.ones(x.size - len(x0))]))
        fo, go, co, Jo, Ho = o.eval(x, jac=True, hess=True)
        r = b.debug_eval(0, st, x)
        assert r['err'] == 0
        assert abs(r['f'] - fo) <= 1e-10 * abs(fo)
        assert rel_max(r['g'], go) < 1e-10 and rel_max(r['c'], co) < 1e-10
        assert rel_max(r['J'], Jo) < 1e-10 and rel_max(r['H'], Ho) < 1e-10
    b.free()


@pytest.mark.parametrize('seed,F', [(0, 60), (5, 90)])
def test_solve_matches_oracle(solver, oracle_lib, seed, F):
    seq = make_walk(seed=seed, F=F, randomize=True)
    ostats, osnaps = oracle_run(seq, CAP)
    res, st = solver.solve([seq])
    r = res[0]
    for k in range(3):
        e = snapshot_errors(r.snapshots[k], osnaps[k])
        assert e['n_samples'][0] == e['n_samples'][1]
        assert e['contact_mismatch'] == 0
        for key in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force'):
            assert e[key] < 1e-3, (k, key, e)
    for stg in range(len(ostats)):
        assert r.stage_status[stg] == ostats[stg][0]


def test_damping_rule_1_on_gpu_in_lockstep_with_the_emulation_of_the_kernel_source():
    """chd_config.damping_rule = 1 through the C ABI on the MI355X against the host emulation of the same source (itself held to the oracle's ratio_low = 0.25 by
    tests/test_host_emu.py): identical stage statuses and iteration counts, snapshots to 1e-9 -- on walks incl. the bench straggler 1688, whose duration stage the rule changes."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'host_emu'))
    import emu
    from chd_amd.phys_optim import PhysOptim, default_config
    emu.build()
    caps = [7000, 7000, 7000, 2500, 2000, 7000]
    seqs = [make_walk(seed=s, F=F, randomize=True) for s, F in ((9, 90), (1688, 90), (2, 40), (31, 60))]
    s = PhysOptim(device=0, config=default_config(max_iter=caps, damping_rule=1))
    res, _ = s.solve(seqs)
    s.close()
    d = PhysOptim(device=0, config=default_config(max_iter=caps))
    res0, _ = d.solve(seqs[1:2])
    d.close()
    assert list(res0[0].stage_iters) != list(res[1].stage_iters)          # (not a no-op on the straggler)
    for seq, r in zip(seqs, res):
        e = emu.EmuProblem(seq, default_config(max_iter=caps, damping_rule=1))
        e.solve(0, 4)
        st, sn = e.results()
        n_st = 5
        if int(st[4][0]) != 0:
            assert e.rebuild_fallback() == 1
            e.solve(5, 5); st, sn = e.results(); n_st = 6
        assert [(int(st[k][0]), int(st[k][1])) for k in range(n_st)] == [(int(r.stage_status[k]), int(r.stage_iters[k])) for k in range(n_st)]
        for k in range(3):
            err = snapshot_errors(r.snapshots[k], sn[k])
            assert err['contact_mismatch'] == 0 and max(err['base_lin'], err['base_ang_deg'], err['ee_pos'], err['ee_force']) < 1e-9, err


def test_solve_matches_golden_vectors(solver):
    """The committed oracle outputs (tests/golden/phys_golden.npz, made by tests/golden/make_phys_golden.py): the HIP
    path reproduces the three snapshots of every case to 1e-3 relative L2 (measured: ~1e-12), contact flags bit-exact,
    stage statuses identical, without running the oracle."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'phys_golden.npz'))
    cases = [(0, 60), (5, 90), (2, 40)]
    seqs = [make_walk(seed=s, F=F, randomize=True) for s, F in cases]
    res, _ = solver.solve(seqs)          # one ragged batch (40, 60 and 90 frames)
    for (seed, F), r in zip(cases, res):
        key = 's%d_F%d' % (seed, F)
        assert list(r.stage_status[:len(g[key + '_status'])]) == list(g[key + '_status'])
        for k in range(3):
            sn = r.snapshots[k]
            for name, val in (('base_lin', sn.base_lin), ('base_ang_deg', sn.base_ang_deg), ('ee_pos', sn.ee_pos), ('ee_force', sn.ee_force)):
                ref = g['%s_snap%d_%s' % (key, k, name)]
                err = np.linalg.norm(np.asarray(val) - ref) / max(np.linalg.norm(ref), 1e-300)
                assert err < 1e-3, (key, k, name, err)
            assert np.array_equal(np.asarray(sn.contact), g['%s_snap%d_contact' % (key, k)])


def test_batch_slot_independence_and_constraints(solver):
    """Full-size property checks (no oracle needed): a sequence gives bit-identical results wherever it sits in a
    batch and whatever its neighbours are (owner-computes accumulation, fixed-tree reductions), every stage reports
    a small constraint violation, and the contact flags of the first two snapshots follow the input schedule."""
    seqs = [make_walk(seed=s, F=90, randomize=True) for s in range(12)]
    res_a, _ = solver.solve(seqs)
    order = [7, 3, 11, 0, 5]
    res_b, _ = solver.solve([seqs[i] for i in order])
    for j, i in enumerate(order):
        for k in range(3):
            a, b = res_a[i].snapshots[k], res_b[j].snapshots[k]
            assert np.array_equal(a.base_lin, b.base_lin) and np.array_equal(a.ee_force, b.ee_force) and np.array_equal(a.ee_pos, b.ee_pos)
        assert res_a[i].stage_iters == res_b[j].stage_iters
    for i, r in enumerate(res_a):
        for stg in range(5):
            assert r.stage_status[stg] in (0, -1, -2)
            if r.stage_status[stg] == 0:
                assert r.stage_constr_viol[stg] < 1e-3
        for k in range(3):
            assert r.snapshots[k].base_lin.shape == (90, 3) and np.isfinite(r.snapshots[k].ee_force).all()
        toe_l = np.asarray(seqs[i].contacts[:, 1])
        assert np.abs(r.snapshots[0].contact[0][:-1] - toe_l[:-1]).sum() <= len(seqs[i].durations[0])
        # stance feet do not move: the foot position is constant wherever the solver says "contact" (snapshot before durations move)
        c0 = r.snapshots[1].contact[0].astype(bool)
        p0 = r.snapshots[1].ee_pos[0]
        runs = np.flatnonzero(c0[1:] & c0[:-1])
        assert np.abs(p0[runs + 1] - p0[runs]).max() < 1e-9


def test_file_interface_round_trip(solver, tmp_path):
    """chd_phys_solve_dirs: the four input files in, the three solution files + success_log out, parsed by the
    line-indexed reader of towr_utils.load_results; identical to the in-memory interface."""
    from chd_amd import io_formats as iof
    seq = make_walk(seed=2, F=40, randomize=True)
    din = str(tmp_path / 'phys_optim_in_ybot'); dout = str(tmp_path / 'phys_optim_out_ybot')
    iof.write_inputs(seq, din)
    import os
    os.makedirs(dout)
    st = solver.solve_dirs([din], [dout], [seq.F])
    assert st == [0]
    assert sorted(os.listdir(dout)) == ['sol_out_durations.txt', 'sol_out_dynamics.txt', 'sol_out_no_dynamics.txt', 'success_log.txt']
    mem, _ = solver.solve([iof.read_inputs(din, seq.F)])
    for k, name in enumerate(('sol_out_no_dynamics.txt', 'sol_out_dynamics.txt', 'sol_out_durations.txt')):
        sol = iof.load_results(os.path.join(dout, name))
        assert sol.num_frames == seq.F
        assert np.allclose(sol.base_lin, mem[0].snapshots[k].base_lin, rtol=1e-8, atol=1e-9)
        assert np.allclose(sol.ee_force, mem[0].snapshots[k].ee_force, rtol=1e-8, atol=1e-7)
        assert np.array_equal(sol.contact, mem[0].snapshots[k].contact)
    log = open(os.path.join(dout, 'success_log.txt')).read().split()
    assert log[0] == 'dynamics' and log[2] == 'durations' and log[1] in '01' and log[3] in '01'


REF_CAPS = [7000, 7000, 7000, 2500, 2000, 7000]


def test_bench_workload_matches_oracle():
    """The BASELINE workload itself (bench.py's first batch: seeds 0..127, 90 frames, reference iteration caps) and 32
    tilted-floor sequences, one persistent launch, against the committed oracle results
    (tests/golden/bench_parity_golden.npz, made by tests/golden/make_bench_parity_golden.py): every sequence whose
    stages all converge in the oracle -> identical stage statuses and iteration counts, contact flags bit-exact,
    trajectories and forces within 1e-3 relative L2 (measured: ~1e-12).  A sequence on which a stage fails in the oracle
    (its iterates stop on the noise floor of the merit function, where two implementations need not take the same
    path) must fail in the same stage and stay within 1e-2."""
    import os
    import sys
    from chd_amd.phys_optim import PhysOptim, default_config
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_bench_parity_golden as mk
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'bench_parity_golden.npz'))
    cases = [c for c in mk.FLAT + mk.TILTED + mk.HARD + mk.PIPE if mk.case_key(*c) + '_status' in g.files]
    assert len(cases) >= 160
    seqs = [mk.make_case(*c) for c in cases]
    s = PhysOptim(device=0, config=default_config(max_iter=REF_CAPS))
    res, st = s.solve(seqs)
    s.close()
    assert st['n_rejected'] == 0 and st['n_stalled'] == 0
    worst = 0.0; n_failed = 0
    for c, r in zip(cases, res):
        key = mk.case_key(*c)
        gs = list(g[key + '_status']); gi = list(g[key + '_iters'])
        ok = all(v == 0 for v in gs)
        err = 0.0
        for k in range(3):
            sn = r.snapshots[k]
            for name, val in (('base_lin', sn.base_lin), ('base_ang_deg', sn.base_ang_deg), ('ee_pos', sn.ee_pos), ('ee_force', sn.ee_force)):
                ref = g['%s_snap%d_%s' % (key, k, name)]
                assert ref.shape == np.asarray(val).shape, (key, k, name)
                nr = np.linalg.norm(ref)
                if nr > 0:
                    err = max(err, float(np.linalg.norm(np.asarray(val) - ref) / nr))
            if ok or k < 2:
                assert np.array_equal(np.asarray(sn.contact), g['%s_snap%d_contact' % (key, k)]), (key, k)
        if ok:
            assert list(r.stage_status[:len(gs)]) == gs and list(r.stage_iters[:len(gi)]) == gi, (key, r.stage_status, r.stage_iters, gs, gi)
            assert err < 1e-3, (key, err)
            worst = max(worst, err)
        else:
            n_failed += 1
            first_bad = next(i for i, v in enumerate(gs) if v != 0)
            assert r.stage_status[first_bad] == gs[first_bad] and list(r.stage_status[:first_bad]) == gs[:first_bad], (key, r.stage_status, gs)
            assert err < 1e-2, (key, err)
    print('bench workload parity: %d sequences, worst rel-L2 %.2e on the %d converging ones' % (len(cases), worst, len(cases) - n_failed))
    assert n_failed <= 3          # (round 4: two of the 200 have a failing stage in the oracle and take the fallback -- bench seed 88 and a hard seed; tilted and pipeline families: none)


def test_bad_sequence_loses_only_itself(tmp_path):
    """The reference runs one process per video, so a video with inconsistent inputs only loses itself
    (run_phys_mocap.py:159-174).  Same here: a batch with a too-short sequence and one whose contact schedules do not sum to
    the same total time (parameters.cpp:150) solves the good ones; the bad ones come back rejected (stage_status -4 in
    memory, status -3 and no output files through the directory interface), and the good results are bit-identical to a
    batch without them."""
    import copy
    import os
    from chd_amd import io_formats as iof
    from chd_amd.phys_optim import PhysOptim, default_config
    good = [make_walk(seed=s, F=40, randomize=True) for s in (2, 3)]
    short = make_walk(seed=4, F=6, randomize=True)
    skew = copy.deepcopy(good[0]); skew.durations = [list(d) for d in skew.durations]; skew.durations[1][0] += 0.05
    s = PhysOptim(device=0, config=default_config(max_iter=CAP))
    ref, _ = s.solve(good)
    res, st = s.solve([short, good[0], skew, good[1]])
    assert st['n_rejected'] == 2
    assert res[0].rejected and res[2].rejected and res[0].stage_status == [-4] * 6 and res[0].snapshots[0].base_lin.shape[0] == 0
    for a, b in ((res[1], ref[0]), (res[3], ref[1])):
        assert not a.rejected and a.stage_status == b.stage_status and a.stage_iters == b.stage_iters
        for k in range(3):
            assert np.array_equal(a.snapshots[k].base_lin, b.snapshots[k].base_lin) and np.array_equal(a.snapshots[k].ee_force, b.snapshots[k].ee_force)
    # directory interface
    dirs = []
    for i, q in enumerate([good[0], skew, good[1]]):
        din = str(tmp_path / ('in%d' % i)); dout = str(tmp_path / ('out%d' % i))
        iof.write_inputs(q, din); os.makedirs(dout)
        dirs.append((din, dout, q.F))
    missing = str(tmp_path / 'nowhere')
    stt = s.solve_dirs([d[0] for d in dirs] + [missing], [d[1] for d in dirs] + [str(tmp_path / 'out0')], [d[2] for d in dirs] + [40])
    assert stt == [0, -3, 0, -1]
    assert 'rejected' in s.last_error() or 'nowhere' in s.last_error()
    assert len(os.listdir(dirs[0][1])) == 4 and len(os.listdir(dirs[1][1])) == 0 and len(os.listdir(dirs[2][1])) == 4
    with pytest.raises(Exception):
        s.solve([short])                       # nothing solvable: the call itself fails
    s.close()


def test_long_sequence_matches_oracle():
    """BASELINE configs[4]: one 600-frame sequence on a floor tilted by 10 degrees (KKT dimension 13 592, border 700), reference
    caps, alone in a launch, against the oracle's committed result (an hour of CPU: make_bench_parity_golden.py --long)."""
    import os
    import sys
    from chd_amd.phys_optim import PhysOptim, default_config
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_bench_parity_golden as mk
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'bench_parity_golden.npz'))
    case = mk.LONG[0]
    key = mk.case_key(*case)
    if key + '_status' not in g.files:
        pytest.skip('fixture made without --long')
    s = PhysOptim(device=0, config=default_config(max_iter=REF_CAPS))
    res, st = s.solve([mk.make_case(*case)])
    s.close()
    r = res[0]
    gs = list(g[key + '_status']); gi = list(g[key + '_iters'])
    assert list(r.stage_status[:len(gs)]) == gs and list(r.stage_iters[:len(gi)]) == gi
    assert r.sizes['kkt_dim'] > 13000
    worst = 0.0
    for k in range(3):
        sn = r.snapshots[k]
        for name, val in (('base_lin', sn.base_lin), ('base_ang_deg', sn.base_ang_deg), ('ee_pos', sn.ee_pos), ('ee_force', sn.ee_force)):
            ref = g['%s_snap%d_%s' % (key, k, name)]
            assert ref.shape == np.asarray(val).shape
            if np.linalg.norm(ref) > 0:
                worst = max(worst, float(np.linalg.norm(np.asarray(val) - ref) / np.linalg.norm(ref)))
        assert np.array_equal(np.asarray(sn.contact), g['%s_snap%d_contact' % (key, k)])
    print('600-frame sequence: worst rel-L2 %.2e, kernel %.0f ms' % (worst, st['kernel_ms'][0]))
    assert worst < 1e-3


def test_narrow_panels_match_the_default_width():
    """chd_config.lds_kilobytes narrows the factorisation's panels (32 -> 16 -> 8 columns: the look-ahead wavefront, its hand-over buffer and the
    tile passes are written for any of them).  K x = b on the KKT matrices of all stages with 80 KB and 44 KB of LDS against the default: the same
    solution to 1e-9 (a narrower panel is a different elimination order), no replaced pivots -- and a whole staged solve that ends where the default's does."""
    from chd_amd.phys_optim import PhysOptim, default_config
    seqs = [make_walk(seed=5, F=60, randomize=True)]
    rng = np.random.default_rng(1)
    ref = None
    sols = {}
    for kb in (0, 80, 44):
        s = PhysOptim(device=0, config=default_config(lds_kilobytes=kb))
        b = s.upload(seqs)
        try:
            xs = []
            r2 = np.random.default_rng(1)
            for stage in range(5):
                N = b.sizes(0, stage)['kkt_dim']
                rhs = r2.normal(size=N)
                x, info = b.debug_linsolve(0, stage, rhs, dw=1e-2, dval=1e-3, which=1)
                assert info['ran'] == 1 and info['bad_pivots'] == 0 and np.isfinite(x).all(), (kb, stage, info)
                xs.append(x)
            b.solve()
            res = b.fetch()[0]
            sols[kb] = (xs, res)
        finally:
            b.free(); s.close()
    for kb in (80, 44):
        for stage in range(5):
            err = np.linalg.norm(sols[kb][0][stage] - sols[0][0][stage]) / np.linalg.norm(sols[0][0][stage])
            assert err < 1e-9, (kb, stage, err)
        a, d = sols[kb][1], sols[0][1]
        assert list(a.stage_status) == list(d.stage_status), (kb, a.stage_status, d.stage_status)
        for key in ('base_lin', 'ee_pos'):
            va, vd = getattr(a.snapshots[-1], key), getattr(d.snapshots[-1], key)
            assert np.allclose(va, vd, rtol=0, atol=1e-6), (kb, key, np.abs(va - vd).max())
