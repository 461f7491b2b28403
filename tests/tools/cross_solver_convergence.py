"""Is the solution of the stage NLPs solver-independent AT CONVERGENCE?  (the well-posed form of "parity with another solver".)

Two correct solvers that both stop at IPOPT's tol = 1e-3 agree only if they take the same iterates (profiles/r02_tolerance_sensitivity.md).  What CAN be asked of any
pair of correct solvers is that they reach the same point when both are run to convergence from the same start.  For every sequence this script
  1. runs the stages in front of the stage under test with the shipped algorithm (oracle, tight tolerance) -- the common start point;
  2. solves the stage with the SHIPPED algorithm (oracle/ipm_solver.hpp: primal-dual interior point, Gauss-Newton + exact blocks, l1 merit) at tol 1e-7;
  3. solves it again, from the same start, with SciPy's `trust-constr` (Byrd-Hribar-Nocedal trust-region interior point / Lalee-Nocedal-Plantenga SQP: no line search, no
     Levenberg damping, its own barrier and trust-region logic) on the oracle's model functions (values, exact sparse Jacobian, Gauss-Newton Hessian of the objective), gtol 1e-9, xtol 1e-10;
  4. optionally with the oracle's IPOPT-like mode (L-BFGS(6) + filter line search), which has no restoration phase and may stop early;
and reports the relative L2 distance of the sampled solution between the solvers: centre of mass, base angles, feet, NET force (sum over the four contact points), NET
moment about the centre of mass, and the per-contact forces.  The first five are what the dynamics rows determine; four contact points sharing one wrench are not.

    python tests/tools/cross_solver_convergence.py [--seeds 16] [--frames 40] [--stages 1 3] [--ipopt-like] [--out tests/golden/cross_solver_golden.json]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path.insert(0, ROOT); sys.path.insert(0, TESTS)
CAPS = [7000, 7000, 7000, 7000, 7000, 7000]
QUANT = ('com', 'base_angles', 'feet', 'net_force', 'net_moment_about_com', 'contact_forces')


def rel(a, b):
    nb = float(np.linalg.norm(b))
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / nb) if nb > 0 else float(np.linalg.norm(a))


def wrench(s):
    """net force (n x 3) and net moment about the centre of mass (n x 3) of a sampled solution"""
    f = np.asarray(s['ee_force']); p = np.asarray(s['ee_pos']); com = np.asarray(s['base_lin'])
    return f.sum(axis=0), np.cross(p - com[None], f).sum(axis=0)


def distances(a, b):
    fa, ma = wrench(a); fb, mb = wrench(b)
    return {'com': rel(a['base_lin'], b['base_lin']), 'base_angles': rel(a['base_ang_deg'], b['base_ang_deg']), 'feet': rel(a['ee_pos'], b['ee_pos']),
            'net_force': rel(fa, fb), 'net_moment_about_com': rel(ma, mb), 'contact_forces': rel(a['ee_force'], b['ee_force'])}


def start_of(seq, stage, tol):
    from oracle.oracle import OracleProblem
    o = OracleProblem(seq, max_iter=CAPS, tol=tol)
    for st in range(stage):
        o.solve_stage(st)
    o.set_stage(stage)
    return o


def work(job):
    seed, F, stage, tol, with_ipopt_like = job
    import chd_amd  # noqa: F401
    import scipy.linalg as sl
    import scipy.optimize as so
    import scipy.sparse as sp
    from chd_amd.synth import make_walk
    from oracle import oracle
    seq = make_walk(seed=seed, F=F, randomize=True)
    out = {'seed': seed, 'frames': F, 'stage': stage}
    # ---- shipped algorithm
    t0 = time.time()
    o = start_of(seq, stage, tol)
    x0 = o.get_x()
    status, info = o.solve_stage(stage)
    xa = o.get_x(); sa = o.sample_solution()
    fa, _, ca, _, _ = o.eval(xa, jac=False)
    cl, cu = o.bounds_at(xa)
    out['shipped'] = {'status': status, 'iterations': info['iters'], 'objective': fa, 'violation': float(np.maximum(np.maximum(cl - ca, ca - cu), 0).max()), 'seconds': time.time() - t0}
    # ---- trust-constr on the same model functions, from the same start
    t0 = time.time()
    o2 = start_of(seq, stage, tol)
    assert np.array_equal(o2.get_x(), x0)
    cl0, cu0 = o2.bounds_at(x0)
    INF = 1e19
    lo = np.where(cl0 <= -INF, -np.inf, cl0); hi = np.where(cu0 >= INF, np.inf, cu0)
    # redundant equality rows (duplicate stance samples) removed by a rank-revealing QR of the Jacobian at two perturbed points: trust-constr's projections need full row rank
    # (with them it falls back to a dense SVD per iteration and does not converge); the feasible set is unchanged
    eq = np.flatnonzero(hi - lo <= 0); iq = np.flatnonzero(hi - lo > 0)
    rng = np.random.default_rng(0)
    Js = np.concatenate([o2.eval(x0 + 1e-2 * rng.normal(size=x0.size))[3][eq] for _ in range(2)], axis=1)
    _, R, piv = sl.qr(Js.T, pivoting=True, mode='economic')
    d = np.abs(np.diag(R)); keep = np.sort(eq[piv[:int((d > 1e-9 * d[0]).sum())]])
    rows = np.concatenate([keep, iq])
    cache = {}

    def ev(x):
        k = x.tobytes()
        if cache.get('k') != k:
            f, g, c, J, H = o2.eval(x, jac=True, hess=True)          # H: the oracle's Gauss-Newton Hessian of the least-squares objective (no multipliers passed: no constraint curvature)
            cache.update(k=k, f=f, g=g, c=c[rows], J=sp.csr_matrix(J[rows]), H=H)
        return cache
    zero_h = sp.csr_matrix((x0.size, x0.size))
    con = so.NonlinearConstraint(lambda x: ev(x)['c'], lo[rows], hi[rows], jac=lambda x: ev(x)['J'], hess=lambda x, v: zero_h)
    r = so.minimize(lambda x: ev(x)['f'], x0, jac=lambda x: ev(x)['g'], hess=lambda x: ev(x)['H'], constraints=[con], method='trust-constr',
                    options={'gtol': 1e-9, 'xtol': 1e-10, 'maxiter': 5000, 'sparse_jacobian': True, 'initial_barrier_parameter': 1e-3})
    xb = np.asarray(r.x)
    fb, _, cb, _, _ = o2.eval(xb, jac=False)
    clb, cub = o2.bounds_at(xb)
    o2.set_x(xb); sb = o2.sample_solution()
    out['trust_constr'] = {'status': int(r.status), 'iterations': int(r.nit), 'objective': fb, 'violation': float(np.maximum(np.maximum(clb - cb, cb - cub), 0).max()),
                           'optimality': float(r.optimality), 'seconds': time.time() - t0}
    out['shipped_vs_trust_constr'] = distances(sa, sb)
    out['x_rel'] = rel(xa, xb)
    if with_ipopt_like:
        t0 = time.time()
        oracle.lib().orc_set_ipopt_like(2)
        o3 = start_of(seq, stage, tol) if stage == 0 else None
        if o3 is None:                       # the stages in front run with the shipped algorithm (same start), only the stage under test with the IPOPT-like mode
            oracle.lib().orc_set_ipopt_like(0)
            o3 = start_of(seq, stage, tol)
            oracle.lib().orc_set_ipopt_like(2)
        st3, info3 = o3.solve_stage(stage, 20000)
        oracle.lib().orc_set_ipopt_like(0)
        xc = o3.get_x(); sc = o3.sample_solution()
        fc, _, cc, _, _ = o3.eval(xc, jac=False)
        clc, cuc = o3.bounds_at(xc)
        out['ipopt_like'] = {'status': st3, 'iterations': info3['iters'], 'objective': fc, 'violation': float(np.maximum(np.maximum(clc - cc, cc - cuc), 0).max()), 'seconds': time.time() - t0}
        out['shipped_vs_ipopt_like'] = distances(sa, sc)
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=16)
    ap.add_argument('--frames', type=int, default=40)
    ap.add_argument('--stages', type=int, nargs='*', default=[1, 3])
    ap.add_argument('--tol', type=float, default=1e-7)
    ap.add_argument('--ipopt-like', action='store_true')
    ap.add_argument('--workers', type=int, default=8)
    ap.add_argument('--out', default='')
    a = ap.parse_args()
    from oracle import oracle
    oracle.build()
    jobs = [(s, a.frames, st, a.tol, a.ipopt_like) for st in a.stages for s in range(a.seeds)]
    with mp.get_context('spawn').Pool(a.workers) as pool:
        res = []
        for r in pool.imap_unordered(work, jobs):
            res.append(r)
            d = r['shipped_vs_trust_constr']
            print('stage %d seed %2d  shipped: %4d it f %.8e viol %.1e | trust-constr: %5d it status %d f %.8e viol %.1e (%.0f s) | rel-L2 ' % (
                r['stage'], r['seed'], r['shipped']['iterations'], r['shipped']['objective'], r['shipped']['violation'], r['trust_constr']['iterations'], r['trust_constr']['status'],
                r['trust_constr']['objective'], r['trust_constr']['violation'], r['trust_constr']['seconds']) + ' '.join('%s %.1e' % (q, d[q]) for q in QUANT)
                + ('' if 'ipopt_like' not in r else ' | ipopt-like: %d it status %d f %.8e: ' % (r['ipopt_like']['iterations'], r['ipopt_like']['status'], r['ipopt_like']['objective'])
                   + ' '.join('%s %.1e' % (q, r['shipped_vs_ipopt_like'][q]) for q in QUANT)), flush=True)
    res.sort(key=lambda r: (r['stage'], r['seed']))
    summary = {}
    for st in a.stages:
        rows = [r for r in res if r['stage'] == st]
        summary['stage_%d' % st] = {'sequences': len(rows),
                                    'shipped_vs_trust_constr_median': {q: float(np.median([r['shipped_vs_trust_constr'][q] for r in rows])) for q in QUANT},
                                    'shipped_vs_trust_constr_max': {q: float(np.max([r['shipped_vs_trust_constr'][q] for r in rows])) for q in QUANT},
                                    'objective_rel_difference_max': float(np.max([abs(r['shipped']['objective'] - r['trust_constr']['objective']) / abs(r['shipped']['objective']) for r in rows]))}
        if a.ipopt_like:
            summary['stage_%d' % st]['shipped_vs_ipopt_like_median'] = {q: float(np.median([r['shipped_vs_ipopt_like'][q] for r in rows])) for q in QUANT}
    print(json.dumps(summary, indent=1))
    if a.out:
        json.dump({'what': __doc__.split('\n\n')[0], 'generator': 'tests/tools/cross_solver_convergence.py --seeds %d --frames %d --stages %s --tol %g%s' % (a.seeds, a.frames, ' '.join(map(str, a.stages)), a.tol, ' --ipopt-like' if a.ipopt_like else ''),
                   'quantities': QUANT, 'summary': summary, 'sequences': res}, open(a.out, 'w'), indent=1)
