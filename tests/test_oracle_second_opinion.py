"""A second opinion on the oracle's SOLVER, independent of its own multipliers and of its linear algebra (the
oracle's model functions f, grad, c, J are pinned to finite differences in tests/test_oracle.py, and this file only goes
through them: OracleProblem.eval / bounds).

1. KKT check with SciPy: at the point the oracle's interior-point method returns, multipliers are re-estimated from
   scratch by bounded least squares (scipy.optimize.lsq_linear: min |g + J_A^T lam| over the rows within 1e-3 of a
   bound, sign-constrained for inequality rows) -- stationarity, complementarity and feasibility must be small and must
   shrink with the solver's tolerance.
2. Independent solve with SciPy: SLSQP (a dense active-set SQP with BFGS: a different algorithm family) from the same
   starting point on the same functions must not find a better point and must approach the oracle's objective.

This pins "the oracle finds KKT points of the NLP it states"; it does NOT pin the oracle against IPOPT's iterates (the
reference binary cannot be built: SURVEY 8c).  tests/tools/tolerance_sensitivity.py quantifies how far two correct
solvers stopping at tol = 1e-3 can be apart (profiles/r02_tolerance_sensitivity.md)."""
import numpy as np
import pytest
import scipy.linalg as sl
from scipy.optimize import lsq_linear, minimize

import chd_amd  # noqa: F401
from chd_amd.synth import make_walk


def _stage_start(seq, stage):
    from oracle.oracle import OracleProblem
    o = OracleProblem(seq)
    for st in range(stage):
        o.solve_stage(st)
    o.set_stage(stage)
    return o.get_x()


def _solve(seq, stage, x0, tol):
    from oracle.oracle import OracleProblem
    o = OracleProblem(seq, tol=tol)
    o.set_stage(stage); o.set_x(x0)
    status, info = o.solve_stage(stage)
    return o, status, info, o.get_x()


def _kkt_residuals(o, x):
    cl, cu = o.bounds()
    f, g, c, J, _ = o.eval(x)
    lo = np.where(cl < -1e18, -np.inf, cl); hi = np.where(cu > 1e18, np.inf, cu)
    eq = hi - lo <= 0
    dl = np.where(np.isfinite(lo), c - lo, np.inf); du = np.where(np.isfinite(hi), hi - c, np.inf)
    scale = np.maximum(1.0, np.abs(J).max(axis=1))
    A = np.flatnonzero(eq | (np.minimum(dl, du) <= 1e-3 * scale))
    near_l = dl[A] <= du[A]
    # L = f + lam^T c: a row at its lower bound needs lam <= 0, at its upper bound lam >= 0, an equality row is free
    lb = np.where(eq[A], -np.inf, np.where(near_l, -np.inf, 0.0)); ub = np.where(eq[A], np.inf, np.where(near_l, 0.0, np.inf))
    lam = lsq_linear(J[A].T, -g, bounds=(lb, ub), tol=1e-13, max_iter=500).x
    slack = np.where(eq[A], 0.0, np.minimum(dl[A], du[A]))
    viol = np.maximum(np.maximum(lo - c, 0), np.maximum(c - hi, 0))
    # the solver's own row scaling (IPOPT's gradient-based scaling, nlp_scaling_max_gradient = 100), recomputed here from J: its stopping test bounds the
    # SCALED violation by tol and the unscaled one by constr_viol_tol = 1e-4 -- both are checked (the unscaled bound alone let the check loosen)
    rowscale = np.where(np.abs(J).max(axis=1) > 100.0, 100.0 / np.maximum(np.abs(J).max(axis=1), 1e-300), 1.0)
    return dict(f=f, stationarity=float(np.abs(J[A].T @ lam + g).max()), complementarity=float(np.abs(lam * slack).max()),
                feasibility=float(viol.max()), feasibility_scaled=float((viol * rowscale).max()), gmax=float(np.abs(g).max()))


@pytest.mark.parametrize('stage', [1, 3])       # 1.2 (kinematic rows) and 2.2 (dynamics, forces, height): phys_optim.cpp:591-599, :648-656
def test_oracle_point_satisfies_kkt_by_scipy(oracle_lib, stage):
    seq = make_walk(seed=1, F=40, randomize=True)
    x0 = _stage_start(seq, stage)
    res = {}
    for tol in (1e-3, 1e-5):
        o, status, info, x = _solve(seq, stage, x0, tol)
        assert status == 0
        res[tol] = _kkt_residuals(o, x)
    loose, tight = res[1e-3], res[1e-5]
    # the reference's tolerance (phys_optim.cpp:578): KKT to ~1e-2 of the gradient scale, feasible to constr_viol_tol
    assert loose['stationarity'] <= 0.3 * loose['gmax'] and loose['feasibility'] <= 1e-4
    # 100 x tighter: the independently estimated KKT residual follows
    assert tight['stationarity'] <= 2e-2 * tight['gmax'] and tight['stationarity'] < 0.2 * loose['stationarity']
    # (the unscaled violation is bounded by constr_viol_tol = 1e-4 at every tol, like IPOPT's: the scaled one follows tol, and rows are scaled by up to 100)
    assert tight['complementarity'] <= 1e-5 and tight['feasibility'] <= 1e-4
    # the scaled violation follows the tolerance (rows are scaled at the stage's starting point; here at the solution: a factor of head-room)
    assert loose['feasibility_scaled'] <= 2e-3 and tight['feasibility_scaled'] <= 2e-5, (loose, tight)
    assert tight['f'] <= loose['f'] + 1e-9            # (the barrier pushes the loose solution inside)


def test_scipy_slsqp_does_not_beat_the_oracle(oracle_lib):
    """Stage 1.2, F = 40: SLSQP from the same start on the same callbacks (redundant equality rows -- duplicate stance
    samples -- removed by a rank-revealing QR, SLSQP's LSQ sub-problem needs full row rank)."""
    from oracle.oracle import OracleProblem
    stage = 1
    seq = make_walk(seed=1, F=40, randomize=True)
    x0 = _stage_start(seq, stage)
    _, status, info, xo = _solve(seq, stage, x0, 1e-6)
    assert status == 0
    op = OracleProblem(seq); op.set_stage(stage)
    cl, cu = op.bounds()
    lo = np.where(cl < -1e18, -np.inf, cl); hi = np.where(cu > 1e18, np.inf, cu)
    eq = np.flatnonzero(hi - lo <= 0)
    rng = np.random.default_rng(0)
    Js = np.concatenate([op.eval(x0 + 1e-2 * rng.normal(size=x0.size))[3][eq] for _ in range(2)], axis=1)
    _, R, piv = sl.qr(Js.T, pivoting=True, mode='economic')
    d = np.abs(np.diag(R)); keep = np.sort(eq[piv[:int((d > 1e-9 * d[0]).sum())]])
    il = np.flatnonzero((hi - lo > 0) & np.isfinite(lo)); iu = np.flatnonzero((hi - lo > 0) & np.isfinite(hi))
    cache = {}

    def ev(x):
        k = x.tobytes()
        if k not in cache:
            cache.clear(); cache[k] = op.eval(x, jac=True)
        return cache[k]
    cons = [dict(type='eq', fun=lambda x: ev(x)[2][keep] - lo[keep], jac=lambda x: ev(x)[3][keep]),
            dict(type='ineq', fun=lambda x: np.concatenate([ev(x)[2][il] - lo[il], hi[iu] - ev(x)[2][iu]]),
                 jac=lambda x: np.concatenate([ev(x)[3][il], -ev(x)[3][iu]]))]
    r = minimize(lambda x: ev(x)[0], x0, jac=lambda x: ev(x)[1], constraints=cons, method='SLSQP', options=dict(maxiter=150, ftol=1e-12))
    cs = op.eval(r.x, jac=False)[2]
    viol = max(np.maximum(lo - cs, 0).max(), np.maximum(cs - hi, 0).max())
    fo = info['objective']
    assert viol < 1e-4
    assert r.fun >= fo - 1e-6 * abs(fo)                   # SciPy finds no better feasible point ...
    assert abs(r.fun - fo) <= 3e-3 * abs(fo)              # ... and has come within 0.3 % of the oracle's objective (still descending: BFGS)
    assert np.linalg.norm(r.x - xo) <= 1e-2 * np.linalg.norm(xo)
