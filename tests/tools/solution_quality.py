"""Solver-independent quality of the shipped solve: for bench seeds, at the three output snapshots
(stages 1.2, 2.2, 3), of

    A  the shipped algorithm at the reference's tol = 1e-3 (phys_optim.cpp:578)             -- what libchd_phys.so computes
    B  the IPOPT-like variant of the oracle (L-BFGS(6), mu_init 0.1; IpmOptions::lbfgs) at 1e-3  -- first `n_ipopt_like` seeds only
    C  the shipped algorithm at tol = 1e-6                                                   -- the "converged" reference point

the stage's objective, the largest violation of any constraint row (unscaled), the largest residual of the dynamics rows
(humanoid_dynamic_constraint.cpp:63-143: N resp. N m), and the relative L2 distance of COM / feet / forces to C's solution.
All on the CPU oracle (test infrastructure); the HIP path equals A to 1e-12 (tests/test_gpu_parity.py).

    python tests/tools/solution_quality.py [n_seeds] [n_ipopt_like] [frames] [workers]      -> markdown on stdout

(The IPOPT-like variant needs thousands of iterations per stage with a dense limited-memory update: hours for 90 frames.  The committed
study runs it as a second invocation on 40-frame sequences: `solution_quality.py 6 6 40 6`.)
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
CAPS = [7000, 7000, 7000, 2500, 2000, 7000]
FAM_DYNAMIC = 16          # oracle/nlp_model.hpp: row family flag of the dynamics rows


def staged(seq, tol, ipopt_like=False):
    from oracle.oracle import OracleProblem, lib
    lib().orc_set_ipopt_like(1 if ipopt_like else 0)
    o = OracleProblem(seq, max_iter=CAPS, tol=tol)
    out = []; stats = []
    t0 = time.time()
    for st in range(5):
        status, info = o.solve_stage(st)
        stats.append((status, info['iters']))
        if st in (1, 3, 4):
            x = o.get_x()
            f, _, c, _, _ = o.eval(x, jac=False)
            cl, cu = o.bounds()
            fam = o.row_family()
            viol = np.maximum(np.maximum(cl - c, c - cu), 0.0)
            dyn = viol[fam == FAM_DYNAMIC]
            out.append(dict(obj=f, viol=float(viol.max()) if viol.size else 0.0, dyn=float(dyn.max()) if dyn.size else 0.0, snap=o.sample_solution()))
    lib().orc_set_ipopt_like(0)
    return stats, out, time.time() - t0


def work(args):
    seed, F, with_b = args
    import chd_amd  # noqa: F401
    from chd_amd.synth import make_walk
    from common import rel_l2
    seq = make_walk(seed=seed, F=F, randomize=True)
    res = {'A': staged(seq, 1e-3), 'C': staged(seq, 1e-6)}
    if with_b:
        res['B'] = staged(seq, 1e-3, ipopt_like=True)
    ref = res['C'][1]
    rows = {}
    for name, (stats, out, dt) in res.items():
        rows[name] = dict(stats=stats, seconds=dt, snaps=[])
        for k in range(len(out)):
            d = {q: (rel_l2(out[k]['snap'][q], ref[k]['snap'][q]) if np.asarray(out[k]['snap'][q]).shape == np.asarray(ref[k]['snap'][q]).shape and np.linalg.norm(ref[k]['snap'][q]) > 0 else 0.0)
                 for q in ('base_lin', 'ee_pos', 'ee_force')}
            rows[name]['snaps'].append(dict(obj=out[k]['obj'], viol=out[k]['viol'], dyn=out[k]['dyn'], **d))
    return seed, rows


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    F = int(sys.argv[3]) if len(sys.argv) > 3 else 90
    workers = int(sys.argv[4]) if len(sys.argv) > 4 else min(n, os.cpu_count() or 1)
    from oracle import oracle
    oracle.build()
    with mp.get_context('fork').Pool(workers) as pool:
        res = pool.map(work, [(s, F, s < nb) for s in range(n)], chunksize=1)
    names = ('sol_out_no_dynamics (stage 1.2)', 'sol_out_dynamics (stage 2.2)', 'sol_out_durations (stage 3)')
    lab = {'A': 'A shipped @ 1e-3', 'B': 'B IPOPT-like @ 1e-3', 'C': 'C shipped @ 1e-6'}
    print('%d-frame bench sequences, seeds 0..%d (IPOPT-like variant: seeds 0..%d).  Medians (and maxima) over the seeds.\n' % (F, n - 1, nb - 1))
    print('| snapshot | solver | objective / objective of C | max constraint violation | max dynamics residual | COM vs C | feet vs C | forces vs C | IPM iterations | CPU s |')
    print('|---|---|---|---|---|---|---|---|---|---|')
    for k in range(3):
        for name in ('A', 'B', 'C'):
            rs = [(r[name], r['C']) for _, r in res if name in r and len(r[name]['snaps']) > k]
            if not rs:
                continue
            med = lambda v: float(np.median(v))      # noqa: E731
            ratio = [a['snaps'][k]['obj'] / c['snaps'][k]['obj'] for a, c in rs]
            col = lambda q: [a['snaps'][k][q] for a, _ in rs]      # noqa: E731
            its = [sum(s[1] for s in a['stats']) for a, _ in rs]
            print('| %s | %s | %.4f (max %.4f) | %.1e (%.1e) | %.1e (%.1e) | %.1e (%.1e) | %.1e (%.1e) | %.1e (%.1e) | %.0f | %.0f |'
                  % (names[k], lab[name], med(ratio), max(ratio), med(col('viol')), max(col('viol')), med(col('dyn')), max(col('dyn')), med(col('base_lin')), max(col('base_lin')),
                     med(col('ee_pos')), max(col('ee_pos')), med(col('ee_force')), max(col('ee_force')), med(its), med([a['seconds'] for a, _ in rs])))
    fails = {name: sum(1 for _, r in res if name in r and any(s[0] != 0 for s in r[name]['stats'])) for name in ('A', 'B', 'C')}
    print('\nSequences with a failed stage: A %d, B %d (of %d), C %d.' % (fails['A'], fails['B'], nb, fails['C']))
