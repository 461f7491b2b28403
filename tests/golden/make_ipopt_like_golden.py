"""Generates tests/golden/ipopt_like_golden.json: the shipped algorithm against the oracle's IPOPT-LIKE mode, per trajectory (VERDICT r04 "missing" 1, item 3).

The reference solves every stage with IPOPT (L-BFGS(6) Hessian, filter line search, MA57: phys_optim.cpp:567-580); its binary cannot be built here,
so "within 1e-3 of the IPOPT reference" cannot be measured.  The next best thing inside the repo: the oracle's own interior point in IPOPT-LIKE mode
(oracle/ipm_solver.hpp IpmOptions::lbfgs + ::filter -- limited-memory BFGS(6) of the Lagrangian folded into the KKT solve, IPOPT's mu_init 0.1 on every stage,
the filter line search with IPOPT's constants and second-order correction, acceptable_tol exit; NOT IPOPT: no restoration phase, banded L D L^T for MA57) run to the
reference's tol 1e-3 on the same sequences as the shipped algorithm, and the relative L2 distance of every output trajectory between the two.
This is an explicit PROXY, labelled as such wherever it is quoted (bench.py: parity.vs_ipopt_like).

    python tests/golden/make_ipopt_like_golden.py [--seeds 32] [--frames 40] [--workers 6] [--long-seeds 8 --long-frames 90]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
CAPS = [7000, 7000, 7000, 2500, 2000, 7000]
KEYS = ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force')


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
    lib().orc_set_ipopt_like(2)
    sb, b = oracle_run(seq, CAPS)
    lib().orc_set_ipopt_like(0)
    t2 = time.time()
    dist = [[(rel_l2(a[k][q], b[k][q]) if np.asarray(a[k][q]).shape == np.asarray(b[k][q]).shape and np.linalg.norm(b[k][q]) > 0 else 0.0) for q in KEYS] for k in range(3)]
    obj = lambda st: [st[1][2], st[3][2], st[-1][2]]      # noqa: E731
    return dict(seed=seed, frames=F, shipped=[[int(s[0]), int(s[1])] for s in sa], ipopt_like=[[int(s[0]), int(s[1])] for s in sb], rel_l2=dist,
                objective_shipped=obj(sa), objective_ipopt_like=obj(sb), cpu_seconds=[t1 - t0, t2 - t1])


def summary(rows):
    d = np.array([r['rel_l2'] for r in rows])          # seeds x snapshots x quantities
    return {'sequences': len(rows), 'frames': rows[0]['frames'],
            'median_rel_l2': {q: [float(v) for v in np.median(d[:, :, i], axis=0)] for i, q in enumerate(KEYS)},
            'max_rel_l2': {q: [float(v) for v in d[:, :, i].max(axis=0)] for i, q in enumerate(KEYS)},
            'sequences_within_1e-3_on_every_trajectory': int(np.sum(np.all(d.reshape(len(rows), -1) <= 1e-3, axis=1))),
            'median_iterations': {'shipped': float(np.median([sum(s[1] for s in r['shipped']) for r in rows])), 'ipopt_like': float(np.median([sum(s[1] for s in r['ipopt_like']) for r in rows]))},
            'stages_failed': {'shipped': int(sum(any(s[0] != 0 for s in r['shipped'][:5]) for r in rows)), 'ipopt_like': int(sum(any(s[0] != 0 for s in r['ipopt_like'][:5]) for r in rows))},
            'objective_ipopt_like_over_shipped_median': [float(v) for v in np.median(np.array([r['objective_ipopt_like'] for r in rows]) / np.array([r['objective_shipped'] for r in rows]), axis=0)]}


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=32); ap.add_argument('--frames', type=int, default=40); ap.add_argument('--workers', type=int, default=6)
    ap.add_argument('--long-seeds', type=int, default=8); ap.add_argument('--long-frames', type=int, default=90)
    ap.add_argument('--out', default=os.path.join(HERE, 'ipopt_like_golden.json'))
    a = ap.parse_args()
    from oracle import oracle
    oracle.build()
    jobs = [(s, a.long_frames) for s in range(a.long_seeds)] + [(s, a.frames) for s in range(a.seeds)]          # (long ones first: they set the wall time)
    rows = []
    with mp.get_context('spawn').Pool(a.workers) as pool:
        for r in pool.imap_unordered(work, jobs, chunksize=1):
            rows.append(r)
            print('seed %2d F %d: shipped %s  ipopt-like %s  forces %s  COM %s  (%.0f / %.0f s)' % (r['seed'], r['frames'], r['shipped'], r['ipopt_like'], np.round([d[3] for d in r['rel_l2']], 3),
                                                                                              np.round([d[0] for d in r['rel_l2']], 4), r['cpu_seconds'][0], r['cpu_seconds'][1]), flush=True)
    rows.sort(key=lambda r: (r['frames'], r['seed']))
    out = {'what': 'rel-L2 of every output trajectory (3 snapshots x base_lin, base_ang_deg, ee_pos, ee_force) between the shipped algorithm and the oracle\'s IPOPT-like mode '
                   '(L-BFGS(6), mu_init 0.1, filter line search + second-order correction; no restoration phase, no MA57), both at the reference\'s tol 1e-3 and iteration caps',
           'proxy_for': 'north_star "trajectories within 1e-3 rel-L2 of the IPOPT reference" -- UNMEASURED against IPOPT itself (binary not buildable: SURVEY 8c)',
           'snapshots': ['sol_out_no_dynamics', 'sol_out_dynamics', 'sol_out_durations'], 'quantities': list(KEYS), 'generator': 'tests/golden/make_ipopt_like_golden.py',
           'summary': {('%d_frames' % F): summary([r for r in rows if r['frames'] == F]) for F in sorted(set(r['frames'] for r in rows))},
           'per_sequence': rows}
    json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps(out['summary'], indent=1))
