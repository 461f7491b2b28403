"""CPU oracle self-checks (the reference has no tests and its binary cannot be built: SURVEY 4 / 8c).
The oracle's analytic Jacobian is verified against central finite differences for every stage, the
dynamics rows against a hand-built static equilibrium, and the whole staged solve against the committed
golden vectors (oracle-generated: they pin regressions and HIP parity, not parity with IPOPT)."""
import os

import numpy as np
import pytest

import chd_amd
from chd_amd.synth import make_walk

from common import oracle_run, rel_l2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'phys_golden.npz')


@pytest.mark.parametrize('stage', [0, 1, 2, 3, 4])
def test_jacobian_and_gradient_vs_finite_differences(oracle_lib, stage):
    from oracle.oracle import OracleProblem
    seq = make_walk(seed=3, F=40, randomize=True, tilt_deg=4.0)
    o = OracleProblem(seq)
    o.set_stage(stage)
    rng = np.random.default_rng(0)
    x = o.get_x()
    x = x + 1e-3 * rng.normal(size=x.size)
    f, g, c, J, _ = o.eval(x)
    cols = rng.choice(x.size, size=min(60, x.size), replace=False)
    if stage == 4:
        cols = np.concatenate([cols, np.arange(o.var_offsets()[10], x.size)])     # every duration variable
    h = 1e-6
    for j in cols:
        xp = x.copy(); xm = x.copy(); xp[j] += h; xm[j] -= h
        fp, _, cp, _, _ = o.eval(xp, jac=False); fm, _, cm, _, _ = o.eval(xm, jac=False)
        fd = (cp - cm) / (2 * h)
        scale = max(1.0, np.abs(J[:, j]).max())
        assert np.abs(fd - J[:, j]).max() <= 2e-5 * scale, (stage, j)
        assert abs((fp - fm) / (2 * h) - g[j]) <= 2e-5 * max(1.0, abs(g[j])), (stage, j)


def test_static_equilibrium_satisfies_dynamics(oracle_lib):
    """Standing still with m g / 4 on each of the four contact points (all below the COM, symmetric): the six
    centroidal-dynamics rows vanish (humanoid_rigid_body_dynamics.cpp:89-115)."""
    from oracle.oracle import OracleProblem
    from chd_amd.io_formats import SeqInput
    F, dt, m = 40, 1.0 / 30, 73.0
    one = np.ones((F, 1))
    def const(v): return one * np.asarray(v, dtype=float)[None, :]
    T = (F - 1) * dt
    seq = SeqInput(F=F, dt=dt, hip_l=const([0, .09, -.05]), hip_r=const([0, -.09, -.05]), leg_len=1.2, heel_len=1.2, heel_dist=0.2,
                   mass=m, inertia=const([9, 9, 1.2, 0, 0, 0]), com=const([0, 0, .9]), euler=const([0, 0, 0]),
                   ltoe=const([.1, .1, 0]), lheel=const([-.1, .1, 0]), rtoe=const([.1, -.1, 0]), rheel=const([-.1, -.1, 0]),
                   normal=np.array([0., 0, 1]), point=np.zeros(3), start_contact=[1, 1, 1, 1], durations=[[T]] * 4)
    o = OracleProblem(seq)
    o.set_stage(2)
    x = o.get_x()
    off = o.var_offsets()
    # base at rest at the data position, feet at their data positions (stance variables), forces keep the initial m g / 4
    xb = x.copy()
    xb[off[0]:off[1]] = 0.0
    f, g, c, J, _ = o.eval(xb, jac=False)     # builds nothing new; we only need row families
    fam = o.row_family()
    # set base-lin node positions to the COM and velocities to 0, base-ang all zero, ee positions to targets
    n_lin = off[1] - off[0]
    lin = np.zeros(n_lin)
    # NodesVariablesAll layout with the two fixed end velocities removed: node0 (p), nodes 1..N-2 (p, v), node N-1 (p)
    lin[0:3] = [0, 0, .9]
    k = 3
    while k + 6 <= n_lin - 3:
        lin[k:k + 3] = [0, 0, .9]; lin[k + 3:k + 6] = 0; k += 6
    lin[k:k + 3] = [0, 0, .9]
    xb[off[0]:off[1]] = lin
    xb[off[1]:off[2]] = 0.0
    for e, p in enumerate(([.1, .1, 0], [.1, -.1, 0], [-.1, .1, 0], [-.1, -.1, 0])):      # NLP ee order L-toe, R-toe, L-heel, R-heel
        xb[off[2 + e]:off[3 + e]] = p
    f, g, c, J, _ = o.eval(xb, jac=False)
    assert np.abs(c[fam == 16]).max() < 1e-9


def test_golden_vectors_reproduce(oracle_lib):
    g = np.load(GOLD)
    for seed, F in [(0, 60), (2, 40)]:
        seq = make_walk(seed=seed, F=F, randomize=True)
        stats, snaps = oracle_run(seq, [300] * 6)
        key = 's%d_F%d' % (seed, F)
        assert [s[0] for s in stats] == list(g[key + '_status'])
        assert [s[1] for s in stats] == list(g[key + '_iters'])
        for k in range(3):
            for name in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force'):
                assert rel_l2(snaps[k][name], g['%s_snap%d_%s' % (key, k, name)]) < 1e-9
            assert np.array_equal(snaps[k]['contact'], g['%s_snap%d_contact' % (key, k)])


def test_output_sampling_contract(oracle_lib):
    """SaveSolution: num_frames header int((T+1e-5)/dt)+1 and one sample per data frame (phys_optim.cpp:71-84);
    contact flags follow the input schedule before the durations are optimised."""
    seq = make_walk(seed=1, F=40, randomize=True)
    stats, snaps = oracle_run(seq, [50] * 6)
    for sn in snaps:
        assert sn['num_frames'] == seq.F and sn['n_samples'] == seq.F
    toe_l = np.asarray(seq.contacts[:, 1])          # foot_contacts.npy column 1 = l_toe; NLP ee 0 = L-toe
    # a frame exactly on a phase boundary belongs to the earlier phase (Spline::GetSegmentID): at most one flip per boundary
    assert np.abs(snaps[0]['contact'][0][:-1] - toe_l[:-1]).sum() <= len(seq.durations[0])


def test_heel_distance_curvature_vs_finite_differences(oracle_lib):
    """The exact node-node block lam * grad^2 c of the heel-distance rows (ee_dist_constraint.cpp:29-94; what the solver adds to its
    Gauss-Newton Hessian): H(lam) - H(0) against central differences of J^T lam, multipliers on the heel rows only."""
    from oracle.oracle import OracleProblem
    seq = make_walk(seed=3, F=40, randomize=True, tilt_deg=4.0)
    o = OracleProblem(seq)
    o.set_stage(3)
    rng = np.random.default_rng(2)
    x = o.get_x() + 1e-2 * rng.normal(size=o.n)
    fam = o.row_family()
    lam = np.where(fam == 8, rng.normal(size=o.m) * 30.0, 0.0)
    assert (fam == 8).sum() > 0
    _, _, _, _, H1 = o.eval(x, jac=True, hess=True, lam=lam)
    _, _, _, _, H0 = o.eval(x, jac=True, hess=True)
    D = H1 - H0
    assert np.abs(D - D.T).max() < 1e-12 and np.abs(D).max() > 1.0
    cols = np.flatnonzero(np.abs(D).sum(axis=0) > 0)
    cols = rng.choice(cols, size=min(40, cols.size), replace=False)
    h = 1e-6
    for j in cols:
        xp = x.copy(); xm = x.copy(); xp[j] += h; xm[j] -= h
        Jp = o.eval(xp)[3]; Jm = o.eval(xm)[3]
        fd = (Jp - Jm).T @ lam / (2 * h)
        assert np.abs(fd - D[:, j]).max() <= 1e-5 * max(1.0, np.abs(D[:, j]).max()), j


def test_node_duration_block_vs_finite_differences(oracle_lib):
    """The exact node x duration block of the Lagrangian Hessian in the duration stage (round 4; `Problem::dur_cross`, the per-family records of nlp_model.hpp):
    H[x, T] against central differences of grad_x (f + lam^T c), for the cost terms alone (lam = 0) and with multipliers on every row family; the
    duration x duration block with it."""
    from oracle.oracle import OracleProblem
    seq = make_walk(seed=3, F=40, randomize=True, tilt_deg=5.0)
    o = OracleProblem(seq)
    o.set_stage(4)
    n, m = o.n, o.m
    nn = int(o.var_offsets()[10])
    rng = np.random.default_rng(1)
    x = o.get_x(); x[:nn] += 0.01 * rng.normal(size=nn); x[nn:] *= 1 + 0.05 * rng.normal(size=n - nn)
    for lam in (np.zeros(m), rng.normal(size=m)):
        H = o.eval(x, hess=True, lam=lam)[4]
        h = 1e-6
        for j in range(nn, n):
            xp = x.copy(); xm = x.copy(); xp[j] += h; xm[j] -= h
            rp = o.eval(xp); rm = o.eval(xm)
            fd = ((rp[1] + rp[3].T @ lam) - (rm[1] + rm[3].T @ lam)) / (2 * h)
            scale = max(1.0, np.abs(fd).max())
            assert np.abs(fd[:nn] - H[:nn, j]).max() <= 2e-6 * scale, ('node x duration', j)
            assert np.abs(fd[nn:] - H[nn:, j]).max() <= 2e-6 * scale, ('duration x duration', j)


def test_exact_curvature_blocks_match_finite_differences(oracle_lib):
    """The study blocks of profiles/r04_curvature_study.md (NOT in the shipped model: they slow the solve down): exact node-node curvature of the dynamics
    rows (torque term by hand, angular term by second-order AD) and of the leg-length rows, H(lam) - H(0) against central differences of J^T lam."""
    from oracle.oracle import OracleProblem, lib
    lib().orc_set_study_mask(7, 0.0)
    try:
        seq = make_walk(seed=3, F=40, randomize=True, tilt_deg=5.0)
        o = OracleProblem(seq)
        o.set_stage(3)
        rng = np.random.default_rng(2)
        x = o.get_x() + 1e-2 * rng.normal(size=o.n)
        fam = o.row_family()
        for f_ in (16, 4):
            lam = np.where(fam == f_, rng.normal(size=o.m), 0.0)
            D = o.eval(x, hess=True, lam=lam)[4] - o.eval(x, hess=True, lam=np.zeros(o.m))[4]
            assert np.abs(D - D.T).max() <= 1e-9 * np.abs(D).max() and np.abs(D).max() > 1.0
            cols = np.flatnonzero(np.abs(D).sum(axis=0) > 0)
            cols = rng.choice(cols, size=min(40, cols.size), replace=False)
            h = 1e-6
            for j in cols:
                xp = x.copy(); xm = x.copy(); xp[j] += h; xm[j] -= h
                fd = (o.eval(xp)[3] - o.eval(xm)[3]).T @ lam / (2 * h)
                assert np.abs(fd - D[:, j]).max() <= 1e-5 * max(1.0, np.abs(D[:, j]).max()), (f_, j)
    finally:
        lib().orc_set_study_mask(0, 0.0)
