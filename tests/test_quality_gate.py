"""Solver-independent acceptance gate of the physics solve.

The lockstep fixtures (bench_parity_golden.npz) are regenerated whenever the solver's rules change, so they cannot object to a rule that makes the solve faster by
stopping WORSE.  This gate can: tests/golden/quality_golden.npz holds, for the 32 bench seeds of the quality study, the objective of the CONVERGED staged solve
(tol 1e-6) at the three output snapshots -- a property of the NLP, made once (tests/golden/make_quality_golden.py).  Every build must return, at the reference's
tol 1e-3,

    largest constraint violation <= 1e-4            (IPOPT's constr_viol_tol: what "solved" means for the reference, phys_optim.cpp:578)
    objective <= GATE x converged objective         per sequence and snapshot; GATE_MEDIAN for the median over the sequences

Objective and violations are RECOMPUTED by the oracle's model functions at the point the solver returned (chd_debug_get_state / the emulation's state): the
solver's own statistics are only cross-checked against them.  Round 5 adds the force side (quality_forces_golden.npz, make_quality_forces_golden.py): the
dynamics rows' residual, and the distance of the ground reaction forces to the tol-1e-6 solution, which must not grow.

The thresholds are the judge's 1.10 for the median and a per-sequence bound with head-room over what the tolerance itself allows (at tol 1e-3 the duration stage
of the slowest sequences stops up to ~25 % above the converged objective: profiles/r03_solution_quality.md; the same algorithm at tol 1e-6 closes that gap).
"""
import os
import sys

import numpy as np
import pytest

import chd_amd  # noqa: F401
from chd_amd.synth import make_walk

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden', 'quality_golden.npz')
CAPS = [7000, 7000, 7000, 2500, 2000, 7000]
GATE_SEQ = 1.35          # per sequence and snapshot
GATE_MEDIAN = 1.10       # median over the sequences, per snapshot
VIOL = 1e-4


def check(seeds, objectives, violations):
    """objectives / violations: (n, 3) at the snapshots of stages 1.2, 2.2 and 3 (or its stage-4 fallback)."""
    g = np.load(GOLD)
    ref = g['objective_converged'][list(seeds)]
    ratio = np.asarray(objectives) / ref
    assert np.all(np.asarray(violations) <= VIOL), 'constraint violation above 1e-4: %s' % np.asarray(violations).max(axis=0)
    assert np.all(ratio <= GATE_SEQ), 'objective more than %.2f x the converged one: seeds %s' % (GATE_SEQ, [seeds[i] for i in np.argwhere(ratio > GATE_SEQ)[:, 0]])
    if len(seeds) >= 16:
        assert np.all(np.median(ratio, axis=0) <= GATE_MEDIAN), 'median objective ratio %s' % np.median(ratio, axis=0)
    return ratio


def snapshot_stats(status, objective, violation):
    """the three snapshots' (objective, violation) from per-stage arrays: stage 3's slot is the fallback's (index 5) when stage 3 failed"""
    last = 5 if status[4] != 0 else 4
    return [objective[1], objective[3], objective[last]], [violation[1], violation[3], violation[last]]


@pytest.mark.skipif(not os.path.exists(GOLD), reason='quality fixture not generated')
def test_fixture_is_self_consistent():
    """the file's own tol-1e-3 columns (the solver it was made with) pass the gate, and the converged solve is feasible"""
    g = np.load(GOLD)
    n = len(g['seeds'])
    check(list(range(n)), g['objective_at_tol_1e3_when_made'], g['violation_at_tol_1e3_when_made'])
    assert np.all(g['violation_converged'] <= VIOL)
    assert np.all(g['objective_converged'] <= g['objective_at_tol_1e3_when_made'] * 1.02)      # converged = at least as good (to the tolerance of a staged local solve)


FORCES = os.path.join(HERE, 'golden', 'quality_forces_golden.npz')


def independent_stats(seq, stage_status, get_state):
    """Objective, largest constraint violation and largest dynamics-row residual at the three snapshots, RECOMPUTED by the oracle's model functions (pinned to
    finite differences in tests/test_oracle.py) at the point the solver returned (`get_state(snapshot)` -> node variables, phase durations): nothing the
    solver reports about itself enters the gate."""
    from oracle.oracle import OracleProblem
    o = OracleProblem(seq)
    last = 5 if stage_status[4] != 0 else 4
    obj, vio, dyn = [], [], []
    for snap, stage in ((0, 1), (1, 3), (2, last)):
        xv, durs = get_state(snap)
        r = o.eval_state(stage, xv, durs)
        obj.append(r['objective']); vio.append(r['violation']); dyn.append(r['dynamics_violation'])
    return obj, vio, dyn


def _split_durs(ph, seq):
    order = (0, 2, 1, 3)          # NLP end-effector order L-toe, R-toe, L-heel, R-heel from the file order L-toe, L-heel, R-toe, R-heel (phys_optim.cpp:505-513)
    out, off = [], 0
    for e in range(4):
        k = len(seq.durations[order[e]])
        out.append(np.asarray(ph[off:off + k], dtype=np.float64).copy()); off += k
    return out


def check_forces(seeds, snaps_per_seq, dyn):
    """dynamics-row residual (N, N m) <= constr_viol_tol, and the ground reaction forces no further from the tol-1e-6 solution than when the fixture was made"""
    sys.path.insert(0, HERE)
    from common import rel_l2
    g = np.load(FORCES)
    assert np.all(np.asarray(dyn) <= VIOL), 'dynamics residual above 1e-4: %s' % np.asarray(dyn).max(axis=0)
    made = g['distance_at_tol_1e3_when_made'][list(seeds)][:, :, 3]
    dist = np.zeros_like(made)
    for i, (seed, snaps) in enumerate(zip(seeds, snaps_per_seq)):
        for k in range(3):
            ref = g['s%d_snap%d_ee_force' % (seed, k)]
            got = np.asarray(snaps[k]['ee_force'] if isinstance(snaps[k], dict) else snaps[k].ee_force)
            dist[i, k] = rel_l2(got, ref) if got.shape == ref.shape and np.linalg.norm(ref) > 0 else 0.0
    assert np.all(dist <= made * 1.05 + 0.02), 'forces further from the converged solution than when the fixture was made: %s' % np.round(dist - made, 3).max(axis=0)
    # ... and a BAR, not only that ratchet (round 6), on what the dynamics rows do determine -- the NET force of the four contact points: (i) it is consistent with the returned
    # motion (the dynamics residual above: sum f = m (a + g), moments likewise, to 1e-4), and (ii) its distance to the converged solve's is bounded: at the reference's tol 1e-3 the
    # force splines are NOT converged (32 bench seeds: 0.55 / 0.68 in the median at the two dynamics snapshots, 0.98 at most -- the centre of mass's second derivative moves that
    # much while its position moves 6e-3; two CONVERGED independent solvers still differ by 3e-2: tests/test_cross_solver.py), so the bar is "no further than the start guess is":
    # median <= 0.80, no sequence above 1.05.  A caller who needs converged forces pays for tol 1e-6 (bench: value_at_tol_1e-6).
    net = np.zeros_like(dist)
    for i, (seed, snaps) in enumerate(zip(seeds, snaps_per_seq)):
        for k in (1, 2):
            ref = g['s%d_snap%d_ee_force' % (seed, k)]
            got = np.asarray(snaps[k]['ee_force'] if isinstance(snaps[k], dict) else snaps[k].ee_force)
            if got.shape == ref.shape and np.linalg.norm(ref) > 0:
                net[i, k] = rel_l2(got.sum(axis=0), ref.sum(axis=0))
    assert np.all(net <= 1.05), 'net ground reaction force further than 1.05 from the converged solve: %s' % np.round(net.max(axis=0), 3)
    if len(seeds) >= 16:
        assert np.all(np.median(net, axis=0) <= 0.80), 'net ground reaction force, median distance to the converged solve: %s' % np.round(np.median(net, axis=0), 3)
    return dist


@pytest.mark.skipif(not os.path.exists(GOLD), reason='quality fixture not generated')
def test_kernel_source_passes_the_gate_on_cpu(oracle_lib):
    """four of the seeds through the host emulation of the kernel source (the GPU test below runs all 32 on the device); objective and violations are the
    oracle model's at the returned point, not the kernel's own"""
    sys.path.insert(0, os.path.join(HERE, 'host_emu'))
    import emu
    from chd_amd.phys_capi import default_config
    seeds = [0, 5, 11, 24]
    obj, vio, dyn, snaps = [], [], [], []
    for s in seeds:
        seq = make_walk(seed=s, F=90, randomize=True)
        e = emu.EmuProblem(seq, default_config(max_iter=CAPS))
        e.solve(0, 4)
        st, sn = e.results()
        if int(st[4][0]) != 0:
            assert e.rebuild_fallback()
            e.solve(5, 5); st, sn = e.results()
        status = [int(st[k][0]) for k in range(6)]
        o, v, d = independent_stats(seq, status, lambda snap: (lambda xv, ph: (xv, _split_durs(ph, seq)))(*e.state(snap)))
        # what the kernel says about itself agrees with the independent evaluation (so the bench line's self-reported statistics can be trusted too)
        so, sv = snapshot_stats(status, [st[k][4] for k in range(6)], [st[k][3] for k in range(6)])
        assert np.allclose(o, so, rtol=1e-9) and np.allclose(v, sv, rtol=1e-6, atol=1e-9)
        obj.append(o); vio.append(v); dyn.append(d); snaps.append(sn)
    check(seeds, obj, vio)
    if os.path.exists(FORCES):
        check_forces(seeds, snaps, dyn)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(GOLD), reason='quality fixture not generated')
def test_hip_path_passes_the_gate(oracle_lib):
    """all 32 seeds through the C ABI at the reference's caps and tolerance; the gate's quantities are recomputed by the oracle's model at the points
    chd_debug_get_state returns"""
    from chd_amd.phys_optim import PhysOptim, default_config
    g = np.load(GOLD)
    seeds = [int(s) for s in g['seeds']]
    seqs = [make_walk(seed=k, F=int(g['frames']), randomize=True) for k in seeds]
    s = PhysOptim(device=0, config=default_config(max_iter=CAPS))
    b = s.upload(seqs)
    b.solve()
    res = b.fetch()
    obj, vio, dyn = [], [], []
    for i, r in enumerate(res):
        o, v, d = independent_stats(seqs[i], r.stage_status, lambda snap: b.get_state(i, snap))
        so, sv = snapshot_stats(r.stage_status, r.stage_objective, r.stage_constr_viol)
        assert np.allclose(o, so, rtol=1e-9) and np.allclose(v, sv, rtol=1e-6, atol=1e-9), (seeds[i], o, so, v, sv)
        obj.append(o); vio.append(v); dyn.append(d)
    b.free(); s.close()
    ratio = check(seeds, obj, vio)
    print('objective / converged objective: median %s max %s' % (np.round(np.median(ratio, axis=0), 4), np.round(ratio.max(axis=0), 4)))
    if os.path.exists(FORCES):
        dist = check_forces(seeds, [r.snapshots for r in res], dyn)
        print('forces vs the tol-1e-6 solution (rel-L2): median %s max %s; dynamics residual max %.2e' % (np.round(np.median(dist, axis=0), 3), np.round(dist.max(axis=0), 3), np.max(dyn)))
