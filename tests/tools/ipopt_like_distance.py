"""Distance between the shipped algorithm (Gauss-Newton + exact duration block, mu_init 1e-3 on warm stages) and an IPOPT-like
variant of the same interior-point method (limited-memory BFGS(6) Hessian as phys_optim.cpp:572 selects, IPOPT's mu_init 0.1
on every stage; oracle only: IpmOptions::lbfgs) at the reference's tol = 1e-3 -- the only in-container estimate of what
"within 1e-3 relative L2 of the IPOPT reference" could mean (the reference binary cannot be built: SURVEY 8c).  It is NOT
IPOPT: no filter line search, no restoration phase, MA57 replaced by a banded LDL^T.

    python tests/tools/ipopt_like_distance.py [n_sequences] [frames] [workers]      -> markdown on stdout
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
CAPS = [7000, 7000, 7000, 2500, 2000, 7000]


def work(args):
    seed, F = args
    import chd_amd  # noqa: F401
    from chd_amd.synth import make_walk
    from common import oracle_run, rel_l2
    from oracle.oracle import lib
    seq = make_walk(seed=seed, F=F, randomize=True)
    t0 = time.time()
    sa, a = oracle_run(seq, CAPS)
    t1 = time.time()
    lib().orc_set_ipopt_like(1)
    sb, b = oracle_run(seq, CAPS)
    lib().orc_set_ipopt_like(0)
    t2 = time.time()
    out = []
    for k in range(3):
        row = {q: rel_l2(a[k][q], b[k][q]) for q in ('base_lin', 'base_ang_deg', 'ee_pos')}
        fa, fb = np.asarray(a[k]['ee_force']), np.asarray(b[k]['ee_force'])
        row['ee_force'] = rel_l2(fa, fb) if np.linalg.norm(fb) > 0 else 0.0
        out.append(row)
    return seed, [(s[0], s[1]) for s in sa], [(s[0], s[1]) for s in sb], out, t1 - t0, t2 - t1


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else min(n, os.cpu_count() or 1)
    from oracle import oracle
    oracle.build()
    with mp.get_context('spawn').Pool(workers) as pool:
        res = pool.map(work, [(s, F) for s in range(n)], chunksize=1)
    names = ('sol_out_no_dynamics', 'sol_out_dynamics', 'sol_out_durations')
    print('| seed | shipped: (status, iterations) per stage | IPOPT-like: (status, iterations) per stage | CPU s shipped / IPOPT-like |')
    print('|---|---|---|---|')
    for seed, sa, sb, out, ta, tb in res:
        print('| %d | %s | %s | %.0f / %.0f |' % (seed, sa, sb, ta, tb))
    print()
    print('| seed | snapshot | base_lin | base_ang | ee_pos | ee_force |')
    print('|---|---|---|---|---|---|')
    agg = {k: {q: [] for q in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force')} for k in range(3)}
    for seed, sa, sb, out, ta, tb in res:
        for k in range(3):
            r = out[k]
            for q in agg[k]:
                agg[k][q].append(r[q])
            print('| %d | %s | %.1e | %.1e | %.1e | %.1e |' % (seed, names[k], r['base_lin'], r['base_ang_deg'], r['ee_pos'], r['ee_force']))
    print()
    print('| snapshot | median base_lin | median base_ang | median ee_pos | median ee_force | max over all quantities |')
    print('|---|---|---|---|---|---|')
    for k in range(3):
        print('| %s | %.1e | %.1e | %.1e | %.1e | %.1e |' % ((names[k],) + tuple(float(np.median(agg[k][q])) for q in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force')) +
                                                            (max(max(agg[k][q]) for q in agg[k]),)))
