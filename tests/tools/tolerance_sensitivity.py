"""How far apart can two CORRECT solvers be that both stop at IPOPT's tol = 1e-3 (phys_optim.cpp:578)?

The north-star tolerance is "trajectories and GRFs within 1e-3 relative L2 of the IPOPT reference".  IPOPT's iterates
cannot be reproduced without the binary, but the distance between two solutions of the SAME NLP that differ
only in where the iteration stops is a lower bound for what any pair of different, correct solvers can be expected to
agree to.  This script runs the oracle's staged solve twice per sequence -- every stage at tol = 1e-3 (the reference's
setting) and at tol = 1e-6 -- and reports the relative L2 distance of the three output snapshots per quantity, plus the
total ground reaction force (the sum over the four contact points, which the dynamics rows determine) next to the
per-contact forces (which they do not: four contact points share one wrench).

    python tests/tools/tolerance_sensitivity.py [n_sequences] [frames]      -> markdown on stdout
"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
CAPS = [7000, 7000, 7000, 2500, 2000, 7000]


def work(args):
    seed, F = args
    import chd_amd  # noqa: F401
    from chd_amd.synth import make_walk
    from common import oracle_run, rel_l2
    seq = make_walk(seed=seed, F=F, randomize=True)
    sa, a = oracle_run(seq, CAPS, tol=1e-3)
    sb, b = oracle_run(seq, CAPS, tol=1e-6)
    out = []
    for k in range(3):
        row = {q: rel_l2(a[k][q], b[k][q]) for q in ('base_lin', 'base_ang_deg', 'ee_pos')}
        fa, fb = np.asarray(a[k]['ee_force']), np.asarray(b[k]['ee_force'])
        row['ee_force'] = rel_l2(fa, fb) if np.linalg.norm(fb) > 0 else 0.0
        row['total_force'] = rel_l2(fa.sum(axis=0), fb.sum(axis=0)) if np.linalg.norm(fb) > 0 else 0.0
        out.append(row)
    return seed, [(s[0], s[1]) for s in sa], [(s[0], s[1]) for s in sb], out


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 90
    from oracle import oracle
    oracle.build()
    with mp.get_context('spawn').Pool(min(n, os.cpu_count() or 1)) as pool:
        res = pool.map(work, [(s, F) for s in range(n)])
    names = ('sol_out_no_dynamics', 'sol_out_dynamics', 'sol_out_durations')
    print('| seed | snapshot | base_lin | base_ang | ee_pos | ee_force (per contact) | total force | iterations 1e-3 | iterations 1e-6 |')
    print('|---|---|---|---|---|---|---|---|---|')
    agg = {k: {q: [] for q in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force', 'total_force')} for k in range(3)}
    for seed, sa, sb, out in res:
        for k in range(3):
            r = out[k]
            for q in agg[k]:
                agg[k][q].append(r[q])
            print('| %d | %s | %.1e | %.1e | %.1e | %.1e | %.1e | %s | %s |' % (seed, names[k], r['base_lin'], r['base_ang_deg'], r['ee_pos'], r['ee_force'], r['total_force'],
                                                                             sum(s[1] for s in sa), sum(s[1] for s in sb)))
    print()
    print('| snapshot | median base_lin | median base_ang | median ee_pos | median per-contact force | median total force |')
    print('|---|---|---|---|---|---|')
    for k in range(3):
        print('| %s | %.1e | %.1e | %.1e | %.1e | %.1e |' % ((names[k],) + tuple(float(np.median(agg[k][q])) for q in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force', 'total_force'))))
