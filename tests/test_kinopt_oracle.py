"""The kinematic-optimisation oracle (oracle/kinopt_oracle.py) against vectors produced by the REFERENCE's
own functions (tests/golden/make_kinopt_golden.py: `optimize_trajectory`, SciPy `least_squares`, scikit-learn
`HuberRegressor`), and its restated third-party algorithms against the installed packages.

How exact can this be?  The residual, the Jacobian, the skeleton fit and the IK initialisation are deterministic formulas:
they must agree to rounding.  The two `least_squares` solves are not reproducible beyond ~1e-4 even by SciPy itself: LSMR runs
into its iteration limit min(m, n) on this ill-conditioned Jacobian, and the solve ends on `xtol` after rejected steps (the
reference's Jacobian is inexact), so handing SciPy the SAME Jacobian as a dense array instead of a sparse matrix moves the
solution by 5e-5..1e-4 (test_scipy_reproducibility_sets_the_tolerance).  The solver-level tolerance is therefore 1e-3."""
import os

import numpy as np
import pytest

from oracle import kinopt_oracle as ko

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'kinopt_golden.npz')
W1 = (1000.0, 0.1, 0.5, 0.3, 10.0, 0.0)
W2 = (1000.0, 0.1, 0.5, 0.3, 10.0, 10.0)


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


def problem(g, ci, li):
    k = 'c%d_' % ci; q = '%slsq%d_' % (k, li)
    return ko.Problem(g[k + 'fit_offsets'], g[k + 'skel_parents'], g[k + 'poses3D'], g[k + 'root_pos'], g[q + 'pose2d_n'], g[q + 'proj_w'], g[q + 'data_w'],
                      g[q + 'vel'], g[q + 'floor_n'], g[q + 'floor_p'], W1 if li == 0 else W2), q


def test_tables_and_preparation(gold):
    g = gold
    assert np.array_equal(g['forward_mapping'], ko.FORWARD) and np.array_equal(g['backward_mapping'], ko.BACKWARD)
    for ci in range(int(g['n_cases'])):
        k = 'c%d_' % ci
        offs = ko.update_skeleton(g[k + 'skel_offsets'], g[k + 'skel_parents'], g[k + 'poses3D'][:, ko.FORWARD] + g[k + 'root_pos'][:, None])
        assert rel(offs, g[k + 'fit_offsets']) < 1e-14
        p2n, pw, dw = ko.prepare_weights(g[k + 'poses2D'], g[k + 'conf'], g[k + 'pp'], g[k + 'focal'])
        assert rel(pw, g[k + 'lsq0_proj_w']) < 1e-15 and rel(dw, g[k + 'lsq0_data_w']) < 1e-15 and rel(p2n, g[k + 'lsq0_pose2d_n']) < 1e-15


@pytest.mark.parametrize('ci,li', [(0, 0), (0, 1), (1, 1), (2, 1)])
def test_residual_and_jacobian_match_the_reference_functions(gold, ci, li):
    """fun_anim_for_projection / jac_anim_for_projection_sparse (optimize_trajectory.py:237-483) at the start point of each solve --
    including the Jacobian's misplaced root column in the projection rows."""
    P, q = problem(gold, ci, li)
    x0 = gold[q + 'x0']
    assert rel(P.fun(x0), gold[q + 'f0']) < 1e-13
    J = P.jac(x0)
    assert rel(J @ gold[q + 'v'], gold[q + 'Jv']) < 1e-13
    assert rel(J.T @ gold[q + 'u'], gold[q + 'JTu']) < 1e-13


def test_lsmr_matches_scipy():
    from scipy.sparse.linalg import lsmr
    rng = np.random.default_rng(0)
    for m, n, damp in [(60, 25, 0.0), (40, 40, 0.3), (200, 30, 1e-3)]:
        A = rng.normal(size=(m, n)) * np.logspace(0, -3, n)[None]
        b = rng.normal(size=m)
        ref = lsmr(A, b, damp=damp)
        x, istop, itn = ko.lsmr(A.dot, A.T.dot, b, n, damp=damp)
        assert (istop, itn) == (ref[1], ref[2])
        assert rel(x, ref[0]) < 1e-12


def test_trf_matches_scipy_on_a_small_problem():
    from scipy.optimize import least_squares
    rng = np.random.default_rng(1)
    A = rng.normal(size=(30, 6)); t = rng.normal(size=30)

    def fun(x):
        return np.tanh(A @ x) - t + 0.1 * np.concatenate([x, np.zeros(24)]) ** 2

    def jac(x):
        J = (1 - np.tanh(A @ x) ** 2)[:, None] * A
        J[:6] += 0.2 * np.diag(x)
        return J

    x0 = rng.normal(size=6) * 0.3
    ref = least_squares(fun, x0, jac=jac, max_nfev=50, gtol=1e-12, bounds=[-np.inf, np.inf], tr_solver='lsmr')
    x, cost, nfev, njev, status = ko.trf_lsmr(fun, jac, x0)
    assert (nfev, njev, status) == (ref.nfev, ref.njev, ref.status)
    assert rel(x, ref.x) < 1e-9 and abs(cost - ref.cost) < 1e-10 * max(1.0, ref.cost)


def test_huber_matches_sklearn():
    from sklearn.linear_model import HuberRegressor
    rng = np.random.default_rng(2)
    X = rng.normal(size=(40, 2)) * 50
    y = X @ np.array([0.05, -0.1]) + 130 + rng.normal(size=40) * 0.8
    y[[3, 17]] += np.array([12.0, -9.0])
    for eps in (1.5, 2.2):
        h = HuberRegressor(epsilon=eps).fit(X, y)
        coef, c0, sigma, outl = ko.huber_fit(X, y, eps)
        assert rel(coef, h.coef_) < 1e-9 and abs(c0 - h.intercept_) < 1e-9 * abs(h.intercept_) and abs(sigma - h.scale_) < 1e-9 * h.scale_
        assert np.array_equal(outl, h.outliers_) and outl.sum() >= 2


def test_scipy_reproducibility_sets_the_tolerance(gold):
    """SciPy on the same residual / Jacobian functions, Jacobian handed over once as a sparse matrix (what the reference does)
    and once as a dense array: the two results differ by more than rounding -- the bar no restatement can beat."""
    from scipy import sparse
    from scipy.optimize import least_squares
    P, q = problem(gold, 0, 0)
    x0 = gold[q + 'x0']
    kw = dict(max_nfev=50, gtol=1e-12, bounds=[-np.inf, np.inf], tr_solver='lsmr')
    a = least_squares(P.fun, x0, jac=lambda x: sparse.lil_matrix(P.jac(x)), **kw)
    b = least_squares(P.fun, x0, jac=P.jac, **kw)
    assert 1e-7 < rel(a.x, b.x) < 1e-3
    assert rel(a.x, gold[q + 'x']) < 1e-3 and rel(b.x, gold[q + 'x']) < 1e-3


@pytest.mark.parametrize('ci', [0, 1, 2])
def test_whole_optimisation_matches_the_reference(gold, ci):
    g = gold; k = 'c%d_' % ci
    fl = (g[k + 'floor_in_n'], g[k + 'floor_in_p']) if int(g[k + 'given_floor']) else (None, None)
    r = ko.optimize_trajectory(g[k + 'poses2D'], g[k + 'conf'], g[k + 'poses3D'], g[k + 'root_pos'], g[k + 'joint_angles'], g[k + 'skel_offsets'],
                               g[k + 'skel_parents'], g[k + 'pp'], g[k + 'focal'], g[k + 'vel'], fl[0], fl[1])
    assert rel(r['ik_rot'], g[k + 'ik_rot']) < 1e-12                         # IK initialisation: deterministic, to rounding
    for li in range(2):
        q = '%slsq%d_' % (k, li)
        assert rel(r['stages'][li]['x'], g[q + 'x']) < 1e-3                  # (see the module docstring)
        assert abs(r['stages'][li]['cost'] - float(g[q + 'cost'])) < 2e-3 * float(g[q + 'cost'])
    assert np.array_equal(r['vel'], g[k + 'out_vel'])                        # relabelled contacts: exact
    assert np.abs(r['floor_n'] - g[k + 'out_floor_n']).max() < 2e-3
    assert np.abs(r['floor_p'] - g[k + 'out_floor_p']).max() < 0.2           # centimetres (the plane's height under the origin)
    assert rel(r['pose3d'], g[k + 'out_pose3d']) < 1e-3 and rel(r['proj2d'], g[k + 'out_proj2d']) < 1e-3
